"""Large scene (2.36 M triangles by default): build time, depth, render rate, parity windows vs the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import ffi, bits, scenes, objflat
from raylib_amd import binding
tess = int(sys.argv[1]) if len(sys.argv) > 1 else 256
check = (sys.argv[2] != "nocheck") if len(sys.argv) > 2 else True
lib = binding.load(); assert lib.Raylib_Initialize() == 1
lib.RaylibAMD_SetSeed(1)
tmp = os.environ.get("TMPDIR", "/tmp")
orc = ffi.load_oracle()
cam = scenes.CONFIG_CAMERAS["breakfast"]
t = time.time(); obj, n = scenes.cornell(os.path.join(tmp, "huge.obj"), tess=tess, displace_fraction=0.2); print("gen %d tris %.1fs, %.0f MB" % (n, time.time() - t, os.path.getsize(obj) / 1e6), flush=True)
t = time.time(); ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], 3840 / 2160, sun=cam["sun"], sun_dir=cam["sun_dir"]); print("load+bvh %.1fs" % (time.time() - t), flush=True)
W, H = 3840, 2160
for spp in (1, 4):
    t = time.time(); img = ses.render(W, H, spp); dt = time.time() - t
    s = ses.stats()
    print("4K spp", spp, "wall %.3f trace %.1f ms Mrays/s %.1f nodes/ray %.1f tris/ray %.2f depth %d nodes %d alg GB/s %.0f" % (
        dt, s.traceKernelMs, s.rays / s.traceKernelMs / 1e3, s.nodesVisited / s.rays, s.trisTested / s.rays, s.bvhDepth, s.numNodes,
        binding.algorithmic_bytes(s) / s.traceKernelMs / 1e6), flush=True)
if check:
    t = time.time(); flat = objflat.load_obj(obj, orc, sun_illuminance=cam["sun"], sun_direction=cam["sun_dir"]); print("py parse %.1fs" % (time.time() - t), flush=True)
    t = time.time(); scene = orc.scene_create(flat, 1); print("oracle scene %.1fs" % (time.time() - t), flush=True)
    ocam = ffi.make_camera(cam["origin"], cam["look_at"], cam["fov"], W / H)
    st = ffi.make_settings(W, H, 4)
    tot = eq = 0
    for (x0, y0) in ((1900, 1072), (1500, 900), (2300, 1500), (1700, 1300), (2100, 700)):
        want = orc.render_region(scene, ocam, st, x0, y0, 16, 16, seed=1)
        got = img[y0:y0 + 16, x0:x0 + 16]
        e = (bits(got[..., :3]) == bits(want[..., :3])).all(-1)
        print("window", x0, y0, "bit-equal %d/256 mean %.4f" % (e.sum(), want[..., :3].mean()), flush=True)
        tot += 256; eq += e.sum()
    print("TOTAL bit-equal", eq, "/", tot)
