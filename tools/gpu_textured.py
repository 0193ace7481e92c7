"""Textured, alpha-tested scene (tessellated cut-out cards + sky + sun), 1080p x 16 spp: trace time and texel fetches."""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
from raylib_amd import binding, scenes
lib = binding.load()
d = tempfile.mkdtemp()
obj, n = scenes.cutout(os.path.join(d, "c.obj"), tess=int(sys.argv[1]) if len(sys.argv) > 1 else 24)
ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1920 / 1080, sun=(3, 3, 3), sun_dir=(0.1, -0.2, -1.0), sky_image=scenes.sky_panorama())
ses.render(1920, 1080, 16); ses.render(1920, 1080, 16); s = ses.stats()
print("%s: cut-out scene, %d triangles: trace %.2f ms, %.0f Mrays/s, %.1f texel fetches per ray, paths/wave %d" % (
    os.environ.get("RAYLIB_LIB", "libraylib.so").split("/")[-1], n, s.traceKernelMs, s.rays / s.traceKernelMs / 1e3, s.texFetches / s.rays, s.pathsPerWave), flush=True)
