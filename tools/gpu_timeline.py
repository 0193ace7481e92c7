"""Wave timeline of k_trace (diagnostic build: make variant VARIANT=tl EXTRA=-DRL_DIAG_TIMELINE=1; RAYLIB_LIB=.../libraylib_tl.so
RAYLIB_PRINT_STAMPS=1): when the waves start, first see the job queue empty and end, for the full Cornell frame, the slice one of
8 ranks renders, and a 1-spp frame."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
os.environ["RAYLIB_PRINT_STAMPS"] = "1"
from raylib_amd import binding, scenes
import tempfile
lib = binding.load()
d = tempfile.mkdtemp()
obj, _ = scenes.cornell(os.path.join(d, "cornell.obj"))
ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1920 / 1080)
for label, fn in (("full frame 64 spp", lambda: ses.render(1920, 1080, 64)), ("slice 0 of 8, 64 spp", lambda: ses.render_cells(1920, 1080, 64, 0, 8)),
                  ("slice 3 of 8, 64 spp", lambda: ses.render_cells(1920, 1080, 64, 3, 8)), ("full frame 1 spp", lambda: ses.render(1920, 1080, 1))):
    fn()
    print("==", label, flush=True)
    fn()
    s = ses.stats()
    print("   trace %.3f ms" % s.traceKernelMs, flush=True)
