"""Strong-scaling projection on one GPU: kernel and wall time of the slice each of N ranks renders (cells r::N)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
from raylib_amd import binding, scenes
import tempfile
lib = binding.load()
d = tempfile.mkdtemp()
obj, _ = scenes.cornell(os.path.join(d, "cornell.obj"))
ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1920 / 1080)
ses.render(1920, 1080, 64)
for rep in range(2):
    t = time.time(); ses.render(1920, 1080, 64); full_wall = (time.time() - t) * 1e3
    s = ses.stats(); full = (s.traceKernelMs, s.kernelMs, s.wallMs)
print("full frame: trace %.2f ms, all kernels %.2f ms, library wall %.2f ms, python wall %.2f ms" % (full + (full_wall,)))
for world in (2, 4, 8):
    rows = []
    for r in range(world):
        ses.render_cells(1920, 1080, 64, r, world)
        t = time.time(); ses.render_cells(1920, 1080, 64, r, world); pw = (time.time() - t) * 1e3
        s = ses.stats(); rows.append((s.traceKernelMs, s.kernelMs, s.wallMs, pw))
    worst = max(rows)
    print("world %d: slice trace %.2f ms, kernels %.2f, lib wall %.2f (incl. D2H), python %.2f -> x%.2f of the full trace time" %
          (world, worst[0], worst[1], worst[2], worst[3], full[0] / worst[0]))
print("spp sweep (full frame): trace ms")
for spp in (1, 2, 4, 8, 16, 32, 64):
    ses.render(1920, 1080, spp); ses.render(1920, 1080, spp)
    s = ses.stats(); print("  spp %2d: trace %.3f ms  (%.3f ms/spp) trips %d rays %d" % (spp, s.traceKernelMs, s.traceKernelMs / spp, s.waveTrips, s.rays))
import ctypes as C
print("tight loop, no D2H (RaylibAMD_RenderDevice, library-owned output buffer):")
for world, spp in ((1, 64), (8, 64), (1, 8), (1, 1)):
    st = binding.RendererSettings(1920, 1080, spp, 5, 1e-4, 0)
    ts = []
    for it in range(12):
        t = time.time()
        assert lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, 0, world, None) == 1
        ts.append((time.time() - t) * 1e3)
    s = ses.stats()
    print("  world %d spp %2d: trace %.3f ms, kernels %.3f, lib wall %.3f, python per call (last 6 mean) %.3f" % (world, spp, s.traceKernelMs, s.kernelMs, s.wallMs, sum(ts[-6:]) / 6))
