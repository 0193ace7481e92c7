"""Launch time at low sample counts on the configs[2]-size scene in a tight loop (the end-game of a launch dominates there)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
from raylib_amd import binding, scenes
import tempfile
lib = binding.load()
d = tempfile.mkdtemp()
obj, n = scenes.cornell(os.path.join(d, "s.obj"), tess=137, displace_fraction=0.2)
cam = scenes.CONFIG_CAMERAS["breakfast"]
ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], 1920 / 1080, sun=cam["sun"], sun_dir=cam["sun_dir"])
out = []
for spp in (1, 4, 16, 64):
    st = binding.RendererSettings(1920, 1080, spp, 5, 1e-4, 0)
    ts = []
    for it in range(10):
        assert lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, 0, 1, None) == 1
        ts.append(ses.stats().traceKernelMs)
    out.append("spp %d: %.2f ms" % (spp, sum(ts[-6:]) / 6))
print(os.environ.get("RAYLIB_LIB", "base").split("libraylib")[-1], " | ".join(out), flush=True)
