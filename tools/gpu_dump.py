"""Raylib_Render + Raylib_DumpImageData at 1080p: what a front-end sees of a frame (the reference's Raylib_Render returns with the pixels in host memory,
render/renderer.cc:292-296).  The fast path (device packing + pinned, chunked staging: csrc/rl_runtime.inl DeviceDumpRGB) against the plain one."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
from raylib_amd import binding, scenes
import tempfile
lib = binding.load()
assert lib.Raylib_Initialize() == 1
d = tempfile.mkdtemp()
obj, _ = scenes.cornell(os.path.join(d, "cornell.obj"))
ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1920 / 1080)
st = binding.RendererSettings(1920, 1080, 64, 5, 1e-4, 0)
img = lib.Raylib_CreateImage(1920, 1080)
host = np.zeros(1920 * 1080 * 3, np.float32)
hp = host.ctypes.data_as(C.POINTER(C.c_float))
def run(n, dump):
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        lib.Raylib_Render(C.byref(st), ses.scene, ses.camera, img)
        if dump:
            lib.Raylib_DumpImageData(img, hp)
        ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
run(3, True)
r = run(20, False)
f = run(20, True)
os.environ["RAYLIB_FAST_DUMP"] = "0"
p = run(20, True)
print("median of 20: render %.2f ms | render + dump %.2f ms (fast path, +%.2f) | %.2f ms (RGBA read-back to pageable memory + host packing, +%.2f)" % (r, f, f - r, p, p - r), flush=True)
