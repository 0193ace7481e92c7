"""Phase shares of the Cornell frame's megakernel from the diagnostic builds (make variant VARIANT=stamps EXTRA=-DRL_DIAG_STAMPS=1, =2 for
wave-step against lane-step counts): RAYLIB_LIB=.../libraylib_stamps.so python tools/gpu_stamps.py [workload]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ["RAYLIB_PRINT_STAMPS"] = "1"
os.environ.pop("RAYLIB_QUIET", None)
from raylib_amd import binding, scenes
lib = binding.load(); assert lib.Raylib_Initialize() == 1
d = tempfile.mkdtemp()
which = sys.argv[1] if len(sys.argv) > 1 else "cornell"
if which == "cornell":
    cam = scenes.CONFIG_CAMERAS["cornell"]; obj, _ = scenes.cornell(os.path.join(d, "c.obj")); spp = 64
elif which == "textured":   # the 298 k room with albedo maps and alpha-cut-out foliage cards, camera inside (bench.py's extra_textured)
    cam = scenes.CONFIG_CAMERAS["breakfast_interior"]; obj, _ = scenes.textured(os.path.join(d, "t.obj")); spp = int(os.environ.get("SPP", "128"))
else:   # "breakfast" (the configs[2] stand-in seen from outside) or "interior" (the same scene, camera inside)
    cam = scenes.CONFIG_CAMERAS["breakfast_interior" if which == "interior" else "breakfast"]; obj, _ = scenes.cornell(os.path.join(d, "b.obj"), tess=91, displace_fraction=0.2); spp = int(os.environ.get("SPP", "128"))
ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], 1920 / 1080, sun=cam["sun"], sun_dir=cam["sun_dir"])
ses.render(1920, 1080, 1)
ses.render(1920, 1080, spp)
s = ses.stats()
print("trace %.3f ms; rays %d, samples %d, wave trips %d, nodes %d, tris %d, shaded %d" % (s.traceKernelMs, s.rays, s.cameraSamples, s.waveTrips, s.nodesVisited, s.trisTested, s.shadedHits), flush=True)
lib.Raylib_FlushLogThread()
