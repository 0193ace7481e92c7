"""Leaf list against the BVH4 walk (k_trace, LDS-resident scenes): Cornell rooms with 0..7 extra boxes (36..120 triangles), 1080p x 16 spp."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
from raylib_amd import binding, scenes
import tempfile
lib = binding.load()
d = tempfile.mkdtemp()
spots = [(-0.6, 0.15, 0.6), (0.7, 0.9, -0.5), (0.0, 0.12, 0.75), (-0.7, 1.5, -0.6), (0.6, 1.6, 0.5), (0.1, 1.2, -0.7), (-0.2, 0.2, 0.1)]
for extra in range(0, 8):
    objs = scenes.cornell_objects()
    for k in range(extra):
        x, y, z = spots[k]
        objs.append(("extra%d" % k, scenes.WHITE if k % 2 else scenes.RED, scenes._box((x, y, z), (0.22, 0.22, 0.22), 10.0 + 23.0 * k)))
    obj, n = scenes.write_obj(os.path.join(d, "c%d.obj" % extra), objs, scenes.CORNELL_MTL)
    ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1920 / 1080)
    row = []
    for mode in ("1", "0"):
        os.environ["RAYLIB_LEAF_LIST"] = mode
        ses.render(1920, 1080, 16); ses.render(1920, 1080, 16)
        s = ses.stats(); row.append((s.traceKernelMs, s.trisTested / max(1, s.rays)))
    print("%3d triangles: leaf list %.2f ms (%.2f triangle tests per ray), BVH4 %.2f ms (%.2f) -> %.2f" % (n, row[0][0], row[0][1], row[1][0], row[1][1], row[0][0] / row[1][0]), flush=True)
    ses.close()
