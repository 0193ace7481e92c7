"""C3-sized scene (298k triangles, sun): parity windows vs the oracle + timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import ffi, bits, scenes, objflat
from raylib_amd import binding
lib = binding.load(); assert lib.Raylib_Initialize() == 1
lib.RaylibAMD_SetSeed(1)
tmp = os.environ.get("TMPDIR", "/tmp")
orc = ffi.load_oracle()
cam = scenes.CONFIG_CAMERAS["breakfast"]
t = time.time(); obj, n = scenes.cornell(os.path.join(tmp, "big.obj"), tess=91, displace_fraction=0.2); print("gen", n, time.time() - t)
t = time.time(); ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], 1920 / 1080, sun=cam["sun"], sun_dir=cam["sun_dir"]); print("load+bvh", time.time() - t)
W, H = 1920, 1080
for spp in (1, 8):
    t = time.time(); img = ses.render(W, H, spp); dt = time.time() - t
    s = ses.stats()
    print("spp", spp, "wall %.3f trace %.1f ms Mrays/s %.1f rays/sample %.2f nodes/ray %.1f tris/ray %.2f  alg GB/s %.0f" % (
        dt, s.traceKernelMs, s.rays / s.traceKernelMs / 1e3, s.rays / s.cameraSamples, s.nodesVisited / s.rays, s.trisTested / s.rays,
        binding.algorithmic_bytes(s) / s.traceKernelMs / 1e6), s.as_dict())
t = time.time(); flat = objflat.load_obj(obj, orc, sun_illuminance=cam["sun"], sun_direction=cam["sun_dir"]); scene = orc.scene_create(flat, 1); print("oracle scene", time.time() - t)
print("ref-style bvh", orc.bvh_stats(scene))
ocam = ffi.make_camera(cam["origin"], cam["look_at"], cam["fov"], W / H)
st = ffi.make_settings(W, H, 8)
tot = eq = 0
for (x0, y0) in ((952, 536), (300, 300), (1500, 800), (100, 900), (1800, 100), (960, 200)):
    t = time.time(); want = orc.render_region(scene, ocam, st, x0, y0, 16, 16, seed=1); dt = time.time() - t
    got = img[y0:y0 + 16, x0:x0 + 16]
    e = (bits(got[..., :3]) == bits(want[..., :3])).all(-1)
    d = np.abs(got[..., :3] - want[..., :3]).max()
    print("window", x0, y0, "bit-equal %d/256 maxdiff %.3e (oracle %.2fs)" % (e.sum(), d, dt))
    tot += 256; eq += e.sum()
print("TOTAL bit-equal", eq, "/", tot, orc.counters(scene))
