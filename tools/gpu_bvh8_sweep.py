"""Where does the 8-wide walk pay?  Scenes of growing size (the Cornell room and the colonnade at several tessellations, from outside and from inside) timed
under RAYLIB_BVH8=0 and =1 in one process, next to the tree's expected node steps per ray (the scene log's figure the runtime's choice is made on).
usage: python tools/gpu_bvh8_sweep.py"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import ffi, scenes
from raylib_amd import binding
os.environ.pop("RAYLIB_QUIET", None)
lib = binding.load(); assert lib.Raylib_Initialize() == 1
lib.RaylibAMD_SetSeed(1)
orc = ffi.load_oracle()
tmp = tempfile.mkdtemp()
CASES = [("room", scenes.cornell_objects, t, 0.2, c) for t in (6, 12, 24, 48) for c in ("breakfast", "breakfast_interior")] + \
        [("colonnade", scenes.colonnade_objects, t, 0.0, "sponza") for t in (3, 6, 12, 24)]
for name, objs, tess, disp, camname in CASES:
    cam = scenes.CONFIG_CAMERAS[camname]
    obj, flat = helpers.big_scene(os.path.join(tmp, "s.obj"), objs(), scenes.CORNELL_MTL, orc, tess, disp, sun=cam["sun"], sun_dir=cam["sun_dir"])
    ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], 1920 / 1080, sun=cam["sun"], sun_dir=cam["sun_dir"])
    lib.Raylib_FlushLogThread()
    row = []
    for mode in ("0", "1"):
        os.environ["RAYLIB_BVH8"] = mode
        ses.render(1920, 1080, 2)
        best = None
        for _ in range(3):
            ses.render(1920, 1080, 32); s = ses.stats()
            if best is None or s.traceKernelMs < best.traceKernelMs: best = s
        row.append(best)
    a, b = row
    print("%s tess %d cam %s: %d triangles | 4-wide %.2f ms %.1f rec/ray | 8-wide %.2f ms %.1f rec/ray | ratio %.3f" % (
        name, tess, camname, a.numTriangles, a.traceKernelMs, a.nodesVisited / a.rays, b.traceKernelMs, b.nodesVisited / b.rays, b.traceKernelMs / a.traceKernelMs), flush=True)
    ses.close()
