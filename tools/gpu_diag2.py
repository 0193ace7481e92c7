import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import ffi, bits
from raylib_amd import binding
lib = binding.load(); assert lib.Raylib_Initialize() == 1
tmp = os.environ.get("TMPDIR", "/tmp")
orc = ffi.load_oracle()
name = "cutout_sky"
obj, c, flat = helpers.flat_for_case(name, tmp, orc)
scene = orc.scene_create(flat, 1)
ses = binding.SceneSession(lib, obj, c["origin"], c["look_at"], c["fov"], 40 / 28, sun=c["sun"], sun_dir=c["sun_dir"], sky_image=helpers.scenes.sky_panorama())
cam = ffi.make_camera(c["origin"], c["look_at"], c["fov"], 40 / 28)
lib.RaylibAMD_SetSeed(7)
for maxp in (5, 6, 8):
  for spp in (1, 2, 3):
    g = ses.render(40, 28, spp, max_path=maxp)
    o, smp = orc.render(scene, cam, ffi.make_settings(40, 28, spp, max_path=maxp), seed=7, want_samples=True)
    bad = np.argwhere((bits(g) != bits(o)).any(-1))
    print("maxp", maxp, "spp", spp, "differing px", len(bad))
    for (y, x) in bad[:6]:
        print("   px", x, y, "gpu", g[y, x, :3], "oracle", o[y, x, :3], "samples", smp[y, x].tolist())
