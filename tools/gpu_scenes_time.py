"""Frame times of the BASELINE-size stand-ins (configs[2], [3], [4]) on one GPU: python tools/gpu_scenes_time.py [all | c2 c3 c4a c4b ...]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import ffi, scenes
from raylib_amd import binding
which = sys.argv[1:] or ["all"]
lib = binding.load(); assert lib.Raylib_Initialize() == 1
lib.RaylibAMD_SetSeed(1)
orc = ffi.load_oracle()
tmp = tempfile.mkdtemp()
CASES = {"c2": ("breakfast", scenes.cornell_objects, 91, 0.2, 1920, 1080, 128), "c3": ("sponza", scenes.colonnade_objects, 12, 0.0, 1920, 1080, 256),
         "c4a": ("breakfast", scenes.cornell_objects, 256, 0.2, 3840, 2160, 64), "c4b": ("breakfast", scenes.cornell_objects, 530, 0.2, 3840, 2160, 128)}
for name, (camname, objs, tess, disp, w, h, spp) in CASES.items():
    if "all" not in which and name not in which: continue
    cam = scenes.CONFIG_CAMERAS[camname]
    obj, flat = helpers.big_scene(os.path.join(tmp, name + ".obj"), objs(), scenes.CORNELL_MTL, orc, tess, disp, sun=cam["sun"], sun_dir=cam["sun_dir"])
    ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], w / h, sun=cam["sun"], sun_dir=cam["sun_dir"])
    os.remove(obj)
    ses.render(w, h, 1)
    best = None
    for _ in range(3):
        ses.render(w, h, spp); s = ses.stats()
        if best is None or s.traceKernelMs < best.traceKernelMs: best = s
    s = best
    print("%s: %d triangles, %dx%d x %d spp: megakernel %.1f ms (%d launches), %.0f Mrays/s, %.2f rays/sample, %.1f node records/ray, %.2f tris/ray, depth %d, paths/wave %d" % (
        name, s.numTriangles, w, h, spp, s.traceKernelMs, s.traceLaunches, s.rays / s.traceKernelMs / 1e3, s.rays / s.cameraSamples, s.nodesVisited / s.rays, s.trisTested / s.rays, s.bvhDepth, s.pathsPerWave), flush=True)
    ses.close()
