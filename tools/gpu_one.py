"""One small render through the C-ABI (debug aid): python tools/gpu_one.py [w h spp]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
from raylib_amd import binding, scenes
import tempfile
w, h, spp = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (64, 64, 1)))
lib = binding.load()
d = tempfile.mkdtemp()
obj, _ = scenes.cornell(os.path.join(d, 'cornell.obj'))
ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, w / h)
t = time.time()
img = ses.render(w, h, spp)
print("rendered", img.shape, float(img[..., :3].mean()), "in %.3f s" % (time.time() - t), flush=True)
