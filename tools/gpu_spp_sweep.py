"""Launch time against samples per pixel on the configs[2]-size scene: the fixed cost of a launch (ramp-up + tail)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
from raylib_amd import binding, scenes
import tempfile
lib = binding.load()
d = tempfile.mkdtemp()
tess = int(sys.argv[1]) if len(sys.argv) > 1 else 137
obj, n = scenes.cornell(os.path.join(d, "s.obj"), tess=tess, displace_fraction=0.2)
cam = scenes.CONFIG_CAMERAS["breakfast"]
ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], 1920 / 1080, sun=cam["sun"], sun_dir=cam["sun_dir"])
prev = None
for spp in (1, 2, 4, 8, 16, 32, 64, 128):
    ses.render(1920, 1080, spp); ses.render(1920, 1080, spp)
    s = ses.stats()
    print("spp %3d: trace %.2f ms in %d launch(es), %.3f ms/spp, trips %d%s" % (spp, s.traceKernelMs, s.traceLaunches, s.traceKernelMs / spp, s.waveTrips,
          "" if prev is None else ", marginal %.3f ms/spp" % ((s.traceKernelMs - prev[1]) / (spp - prev[0]))), flush=True)
    prev = (spp, s.traceKernelMs)
