"""A/B of the 8-wide collapse: planned (dynamic programme) against opening the largest child first (RAYLIB_WIDE_GREEDY=1 at scene build), RAYLIB_BVH8=1."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import ffi, scenes
from raylib_amd import binding
lib = binding.load(); assert lib.Raylib_Initialize() == 1
lib.RaylibAMD_SetSeed(1)
orc = ffi.load_oracle()
tmp = tempfile.mkdtemp()
CASES = {"c2": ("breakfast", scenes.cornell_objects, 91, 0.2, 1920, 1080, 64), "c2i": ("breakfast_interior", scenes.cornell_objects, 91, 0.2, 1920, 1080, 32),
         "c3": ("sponza", scenes.colonnade_objects, 12, 0.0, 1920, 1080, 64), "c4a": ("breakfast", scenes.cornell_objects, 256, 0.2, 3840, 2160, 32),
         "c4b": ("breakfast", scenes.cornell_objects, 530, 0.2, 3840, 2160, 32)}
which = (sys.argv[1] if len(sys.argv) > 1 else "c2,c2i,c3,c4a").split(",")
os.environ["RAYLIB_BVH8"] = "1"
for name in which:
    camname, objs, tess, disp, w, h, spp = CASES[name]
    cam = scenes.CONFIG_CAMERAS[camname]
    obj, flat = helpers.big_scene(os.path.join(tmp, name + ".obj"), objs(), scenes.CORNELL_MTL, orc, tess, disp, sun=cam["sun"], sun_dir=cam["sun_dir"])
    ref = None
    for g in ("1", "0"):
        os.environ["RAYLIB_WIDE_GREEDY"] = g
        ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], w / h, sun=cam["sun"], sun_dir=cam["sun_dir"])
        img = ses.render(w, h, 1)
        best = None
        for _ in range(3):
            img = ses.render(w, h, spp); s = ses.stats()
            if best is None or s.traceKernelMs < best.traceKernelMs: best = s
        s = best
        same = True if ref is None else bool((img.view("uint32") == ref.view("uint32")).all())
        ref = img.copy()
        print("%s %s: %.2f ms, %.1f node records/ray, %.2f tris/ray, same bits %s" % (name, "greedy" if g == "1" else "planned", s.traceKernelMs, s.nodesVisited / s.rays, s.trisTested / s.rays, same), flush=True)
        ses.close()
    os.remove(obj)
