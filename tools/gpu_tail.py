"""Small launches of k_trace: the slice one of 8 ranks renders of the Cornell frame, the whole frame at 64 / 8 / 1 spp (no read-back)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
from raylib_amd import binding, scenes
import tempfile
lib = binding.load()
d = tempfile.mkdtemp()
obj, _ = scenes.cornell(os.path.join(d, "cornell.obj"))
ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1920 / 1080)
out = []
for world, spp in ((1, 64), (8, 64), (4, 64), (1, 8), (1, 1)):
    st = binding.RendererSettings(1920, 1080, spp, 5, 1e-4, 0)
    tr = []
    for it in range(10):
        assert lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, 0, world, None) == 1
        tr.append(ses.stats().traceKernelMs)
    out.append("1/%d x %2d spp %.3f ms" % (world, spp, sorted(tr[4:])[len(tr[4:]) // 2]))
print(os.path.basename(os.environ.get("RAYLIB_LIB", "libraylib.so")), " | ".join(out))
