"""1/8 slice of the Cornell 1080p x 64 spp frame (what one of 8 ranks renders): kernel time against the job-chunk size."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
from raylib_amd import binding, scenes
import tempfile
lib = binding.load()
d = tempfile.mkdtemp()
obj, _ = scenes.cornell(os.path.join(d, "cornell.obj"))
ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1920 / 1080)
st = binding.RendererSettings(1920, 1080, 64, 5, 1e-4, 0)
for world in (8, 1):
    for chunk in ("", "64", "128", "256", "512", "1024"):
        if chunk: os.environ["RAYLIB_JOB_CHUNK"] = chunk
        elif "RAYLIB_JOB_CHUNK" in os.environ: del os.environ["RAYLIB_JOB_CHUNK"]
        ts = []
        for it in range(8):
            assert lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, 0, world, None) == 1
            ts.append(ses.stats().traceKernelMs)
        print("world %d chunk %-7s trace %.3f ms (min %.3f)" % (world, chunk or "default", sum(ts[-4:]) / 4, min(ts)), flush=True)
