import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import ffi, bits
from raylib_amd import binding
lib = binding.load(); assert lib.Raylib_Initialize() == 1
lib.RaylibAMD_SetSeed(1)
orc = ffi.load_oracle()
g = np.load(os.path.join(helpers.GOLDEN, "procedural.npz"))
mats, sph, cub, c = helpers.procedural_case()
ses = binding.ProceduralSession(lib, mats, sph, cub, c["origin"], c["look_at"], c["fov"], c["aspect"], sun=c["sun"], sun_dir=c["sun_dir"],
                                aperture=c["aperture"], focal=c["focal"], shutter=c["shutter"])
flat, _ = helpers.procedural_flat()
scene = orc.scene_create(flat, 1)
cam = ffi.make_camera(c["origin"], c["look_at"], c["fov"], c["aspect"], c["aperture"], c["focal"], *c["shutter"])
for mode in (2, 1, 5, 0):
    img = ses.render(96, 64, 1, mode=mode)
    want = g["mode0_spp1"] if mode == 0 else g["mode%d" % mode]
    bad = np.argwhere((bits(img) != bits(want)).any(-1))
    print("mode", mode, "differing", len(bad))
    for (y, x) in bad[:8]:
        print("   px", x, y, "gpu", img[y, x, :3], "ref", want[y, x, :3])
rays = np.ascontiguousarray(g["hit_rays"], np.float32)
out = np.zeros(len(rays), ffi.HIT_DTYPE)
lib.RaylibAMD_ClosestHit(ses.scene, rays.ctypes.data_as(C.POINTER(C.c_float)), len(rays), 1e-4, out.ctypes.data)
want = g["hits"]
for f in ("hit", "t", "p", "n", "paramU", "paramV", "material"):
    a, b = out[f], want[f]
    eq = (a == b) if a.dtype != np.float32 else (bits(a) == bits(b))
    if eq.ndim > 1: eq = eq.all(-1)
    print(f, eq.mean())
bad = np.nonzero(bits(out["t"]) != bits(want["t"]))[0][:6]
for i in bad: print(i, rays[i], "gpu", out[i], "ref", want[i])
