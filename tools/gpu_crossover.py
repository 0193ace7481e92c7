"""Where does the pool schedule start to pay?  Tessellated, displaced Cornell rooms of growing triangle count, 1080p x 16 spp."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
from raylib_amd import binding, scenes
import tempfile
lib = binding.load()
d = tempfile.mkdtemp()
for tess in (1, 2, 3, 4, 6, 8, 12, 16, 24):
    obj, n = scenes.cornell(os.path.join(d, "c%d.obj" % tess), tess=tess, displace_fraction=0.2 if tess > 1 else 0.0)
    ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1920 / 1080)
    row = []
    for mode in ("0", "2"):
        os.environ["RAYLIB_POOL"] = mode
        ses.render(1920, 1080, 16); ses.render(1920, 1080, 16)
        s = ses.stats(); row.append((s.traceKernelMs, s.pathsPerWave, s.bvhDepth))
    print("tess %2d tris %6d depth %2d: k_trace %.2f ms, pool %.2f ms (paths/wave %d) -> pool/k_trace %.2f" % (tess, n, row[0][2], row[0][0], row[1][0], row[1][1], row[1][0] / row[0][0]), flush=True)
    ses.close()
