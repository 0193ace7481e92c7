import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RAYLIB_LIB"] = os.path.join(ROOT, "software-raytracing_amd", "libraylib_diag.so")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import scenes
from raylib_amd import binding
lib = binding.load(); assert lib.Raylib_Initialize() == 1
tmp = os.environ.get("TMPDIR", "/tmp")
for name, kw, camname, spp in (("cornell", {}, "cornell", 16), ("breakfast", dict(tess=91, displace_fraction=0.2), "breakfast", 8)):
    cam = scenes.CONFIG_CAMERAS[camname]
    obj, n = scenes.cornell(os.path.join(tmp, name + "_d.obj"), **kw)
    ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], 1920 / 1080, sun=cam["sun"], sun_dir=cam["sun_dir"])
    ses.render(1920, 1080, spp)
    s = ses.stats()
    print(name, "trace ms %.1f rays %d nodes/ray %.1f | wave trips %d, wave-level node steps %d -> %.1f steps per trip; lane nodes per trip-lane %.1f; step efficiency %.3f" % (
        s.traceKernelMs, s.rays, s.nodesVisited / s.rays, s.waveTrips, s.texFetches, s.texFetches / s.waveTrips,
        s.nodesVisited / (64.0 * s.waveTrips), s.nodesVisited / (64.0 * s.texFetches)))
