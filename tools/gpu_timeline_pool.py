"""Wave timeline of the pool megakernel on the configs[2]-sized scene (diagnostic build: make variant VARIANT=tl EXTRA=-DRL_DIAG_TIMELINE=1;
RAYLIB_LIB=.../libraylib_tl.so): start, first "job list empty", end -- per XCD -- with one head (RAYLIB_JOB_HEADS=1) and one head per XCD."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ["RAYLIB_PRINT_STAMPS"] = "1"
os.environ.pop("RAYLIB_QUIET", None)
from raylib_amd import binding, scenes
lib = binding.load(); assert lib.Raylib_Initialize() == 1
d = tempfile.mkdtemp()
cam = scenes.CONFIG_CAMERAS["breakfast"]
obj, _ = scenes.cornell(os.path.join(d, "b.obj"), tess=91, displace_fraction=0.2)
ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], 1920 / 1080, sun=cam["sun"], sun_dir=cam["sun_dir"])
ses.render(1920, 1080, 1)
for heads, guided in (("1", "0"), ("8", "0"), ("8", "1")):
    os.environ["RAYLIB_JOB_HEADS"] = heads; os.environ["RAYLIB_GUIDED"] = guided
    ses.render(1920, 1080, 128)
    print("== heads", heads, "guided", guided, flush=True)
    ses.render(1920, 1080, 128)
    s = ses.stats()
    print("   trace %.3f ms" % s.traceKernelMs, flush=True)
    lib.Raylib_FlushLogThread()
