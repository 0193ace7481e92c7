"""Raylib_Render wall time: kernel + 33 MB read-back into the host Image2D (the drop-in path of the CUI / GUI front-ends)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
from raylib_amd import binding, scenes
import tempfile
lib = binding.load()
d = tempfile.mkdtemp()
obj, _ = scenes.cornell(os.path.join(d, "cornell.obj"))
ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1920 / 1080)
st = binding.RendererSettings(1920, 1080, 64, 5, 1e-4, 0)
def run(fresh):
    img = lib.Raylib_CreateImage(1920, 1080)
    ts = []
    for it in range(8):
        if fresh and it:
            lib.Raylib_DestroyImage(img); img = lib.Raylib_CreateImage(1920, 1080)
        t = time.perf_counter(); lib.Raylib_Render(C.byref(st), ses.scene, ses.camera, img); ts.append((time.perf_counter() - t) * 1e3)
    lib.Raylib_DestroyImage(img)
    return sum(ts[-5:]) / 5
print("Raylib_Render wall %.2f ms (image reused), %.2f ms (fresh image per frame)" % (run(False), run(True)), flush=True)
