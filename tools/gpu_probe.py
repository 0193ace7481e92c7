"""Ad-hoc GPU probe: parity of the HIP path against the oracle on small images, then timing at 1080p."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
from oracle import ffi, objflat
from raylib_amd import scenes, binding

lib = binding.load()
print("init", lib.Raylib_Initialize())
orc = ffi.load_oracle()
ref = ffi.load_ref(True)
tmp = os.environ.get("TMPDIR", "/tmp")
obj, n = scenes.cornell(os.path.join(tmp, "cornell_probe.obj"))
flat = objflat.load_obj(obj, orc)
ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1.0)
so = orc.scene_create(flat, 1)
cam = ffi.make_camera((0, 1, 4), (0, 1, -1), 45.0, 1.0)
lib.RaylibAMD_SetSeed(1)
for mode in (2, 1, 4, 5, 3, 0):
    spp = 4 if mode == 0 else 1
    W = H = 64
    g = ses.render(W, H, spp, mode=mode)
    o = orc.render(so, cam, ffi.make_settings(W, H, spp, mode=mode), seed=1)
    d = np.abs(g[..., :3] - o[..., :3])
    print("mode", mode, "gpu mean %.6f oracle mean %.6f maxdiff %.3e  px differing %d / %d  bit-equal px %d  rmse %.3e" % (
        g[..., :3].mean(), o[..., :3].mean(), d.max(), (d.max(-1) > 0).sum(), W * H,
        (g.view(np.uint32) == o.view(np.uint32)).all(-1).sum(), np.sqrt((d ** 2).mean())))
    print("   stats", ses.stats().as_dict())
if ref is not None:
    sr = ref.scene_create(flat, 1)
    r = ref.render(sr, cam, ffi.make_settings(64, 64, 4), seed=1)
    print("ref vs oracle bit equal:", np.array_equal(r.view(np.uint32), o.view(np.uint32)))
# timing
W, H = 1920, 1080
ses2 = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, W / H)
for spp in (4, 64):
    t = time.time(); img = ses2.render(W, H, spp); dt = time.time() - t
    s = ses2.stats()
    print("   lane slots tracing: %.3f  shading: %.3f" % (s.rays / (64.0 * s.waveTrips), s.shadedHits / (64.0 * s.waveTrips)))
    print("1080p spp", spp, "wall %.3f s kernel %.1f ms trace %.1f ms rays %d  Mrays/s %.1f mean %.5f" % (
        dt, s.kernelMs, s.traceKernelMs, s.rays, s.rays / s.traceKernelMs / 1e3, img[..., :3].mean()))
    print("   ", s.as_dict())
