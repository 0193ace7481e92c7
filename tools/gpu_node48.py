"""48-byte grid nodes (RAYLIB_NODE48=1, three loads per traversal step) against the 64-byte ones (=0) on the 298 k-triangle room seen from inside, under the
reference's sun direction (-1, -1, 0) -- an exact zero: rays that leave the back wall lie IN its plane -- and under a sun without a zero component."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from raylib_amd import binding, scenes
lib = binding.load(); assert lib.Raylib_Initialize() == 1
lib.RaylibAMD_SetSeed(1)
d = tempfile.mkdtemp()
obj, _ = scenes.cornell(os.path.join(d, "b.obj"), tess=91, displace_fraction=0.2)
cam = scenes.CONFIG_CAMERAS["breakfast_interior"]
for sun_dir in ((-1.0, -1.0, 0.0), (-1.0, -1.0, -0.13), None):
    ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], 1920 / 1080, sun=cam["sun"] if sun_dir else (0, 0, 0), sun_dir=sun_dir or (0.0, -1.0, -0.5))
    for mode in ("0", "1"):
        os.environ["RAYLIB_NODE48"] = mode
        ses.render(1920, 1080, 2)
        best = None
        for _ in range(2):
            ses.render(1920, 1080, 64); s = ses.stats()
            if best is None or s.traceKernelMs < best.traceKernelMs: best = s
        s = best
        print("sun direction %s, NODE48=%s: %.1f ms, %.0f Mrays/s, %.1f node records/ray, %.2f tris/ray" % (sun_dir, mode, s.traceKernelMs, s.rays / s.traceKernelMs / 1e3, s.nodesVisited / s.rays, s.trisTested / s.rays), flush=True)
    ses.close()
