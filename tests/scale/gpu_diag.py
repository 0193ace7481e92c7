"""GPU diagnostics: detailed parity metrics per case (prints, never asserts)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import ffi, bits
from raylib_amd import binding
lib = binding.load(); assert lib.Raylib_Initialize() == 1
lib.RaylibAMD_SetSeed(1)
tmp = os.environ.get("TMPDIR", "/tmp")
orc = ffi.load_oracle()
def metrics(a, b):
    d = a[..., :3].astype(np.float64) - b[..., :3]
    per = np.sqrt((d * d).sum(-1))
    return "L2 %.3e max %.3e biteq %.4f  px>1e-4: %d  px>1e-2: %d" % (np.sqrt((per ** 2).mean()), per.max(), (bits(a[..., :3]) == bits(b[..., :3])).all(-1).mean(), (per > 1e-4).sum(), (per > 1e-2).sum())
for name in helpers.CASES:
    g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
    ses = helpers.session_for_case(lib, name, tmp)
    for mode in (1, 2, 4, 5):
        print(name, "mode", mode, metrics(ses.render(64, 64, 1, mode=mode), g["mode%d" % mode]))
    for spp in (1, 4, 16):
        print(name, "spp", spp, metrics(ses.render(64, 64, spp), g["mode0_spp%d" % spp]))
    rays = np.ascontiguousarray(g["hit_rays"], np.float32)
    out = np.zeros(len(rays), ffi.HIT_DTYPE)
    lib.RaylibAMD_ClosestHit(ses.scene, rays.ctypes.data_as(C.POINTER(C.c_float)), len(rays), 1e-4, out.ctypes.data)
    want = g["hits"]
    print(name, "hits: flag eq", (out["hit"] == want["hit"]).mean(), "t biteq", (bits(out["t"]) == bits(want["t"])).mean(),
          "p", (bits(out["p"]) == bits(want["p"])).all(-1).mean(), "n", (bits(out["n"]) == bits(want["n"])).all(-1).mean(),
          "u", (bits(out["paramU"]) == bits(want["paramU"])).mean(), "v", (bits(out["paramV"]) == bits(want["paramV"])).mean(),
          "mat", (out["material"] == want["material"]).mean())
    bad = np.nonzero((bits(out["paramU"]) != bits(want["paramU"])) | (bits(out["t"]) != bits(want["t"])))[0][:5]
    for i in bad:
        print("   ray", i, rays[i], "gpu", out[i], "ref", want[i])
    ses.close()
