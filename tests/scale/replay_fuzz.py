"""Replay one case of tools/gpu_fuzz.py (same RNG sequence) and ask the CPU oracle which schedule is right.
usage: python tests/scale/replay_fuzz.py <seed> <case>"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import ffi, bits, scenes, objflat
from raylib_amd import binding
lib = binding.load()
orc = ffi.load_oracle()
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(seed)
d = tempfile.mkdtemp()
for case in range(target + 1):
    kind = rng.randint(4)
    if kind == 0:
        gen = (scenes.cornell, dict(tess=int(rng.randint(1, 20)), displace_fraction=float(rng.choice([0.0, 0.1, 0.3])),
                                    tall_material=str(rng.choice([scenes.MIRROR, scenes.GLASS, scenes.WHITE])), short_material=str(rng.choice([scenes.WHITE, scenes.GLASS]))))
    elif kind == 1:
        gen = (scenes.soup, dict(n_tris=int(rng.randint(10, 30000)), seed=int(rng.randint(1 << 30)), extent=float(rng.uniform(1, 5)), size=float(rng.uniform(0.05, 1.0))))
    elif kind == 2:
        gen = (scenes.cutout, dict(tess=int(rng.randint(1, 12))))
    else:
        gen = (scenes.colonnade, dict(tess=int(rng.randint(1, 4))))
    sun = (0, 0, 0) if rng.rand() < 0.4 else tuple(float(x) for x in rng.uniform(1, 20, 3))
    sun_dir = tuple(float(x) for x in rng.uniform(-1, 1, 3) * np.array([1, 1, 1]) + np.array([0, -1.2, 0]))
    sky = scenes.sky_panorama() if rng.rand() < 0.4 else None
    origin = tuple(float(x) for x in np.array([0, 1, 4]) + rng.uniform(-1.5, 1.5, 3))
    w, h = int(rng.randint(9, 200)), int(rng.randint(9, 120))
    spp, max_path = int(rng.choice([1, 2, 5, 8])), int(rng.choice([1, 2, 5, 9]))
    aperture = 0.0 if rng.rand() < 0.6 else float(rng.uniform(0.01, 0.2))
    seed_val = int(rng.randint(1, 1 << 30))
    fov = float(rng.uniform(30, 80)); shutter = (0.0, float(rng.choice([0.0, 1.0])))   # drawn in this order by tools/gpu_fuzz.py
obj, n = gen[0](os.path.join(d, "f.obj"), **gen[1])
print("case", target, gen, "tris", n, w, h, "spp", spp, "len", max_path, "sun", sun, "aperture", aperture, "sky", sky is not None, flush=True)
ses = binding.SceneSession(lib, obj, origin, (0, 1, -1), fov, w / h, sun=sun, sun_dir=sun_dir, sky_image=sky, aperture=aperture, focal=4.0, shutter=shutter)
lib.RaylibAMD_SetSeed(seed_val)
imgs = {}
for name, env in (("k_trace bvh4", dict(RAYLIB_POOL="0")), ("k_trace bvh2", dict(RAYLIB_POOL="0", RAYLIB_BVH4="0")), ("pool bvh4", dict(RAYLIB_POOL="2")), ("pool bvh2", dict(RAYLIB_POOL="2", RAYLIB_BVH4="0"))):
    for k, v in env.items(): os.environ[k] = v
    imgs[name] = ses.render(w, h, spp, max_path=max_path)
    for k in env: del os.environ[k]
flat = objflat.load_obj(obj, orc, texture_loader=helpers.texture_loader, sun_illuminance=sun, sun_direction=sun_dir)
if sky is not None:
    flat.textures.append(np.ascontiguousarray(sky, np.float32)); flat.sky_texture = len(flat.textures) - 1
scene = orc.scene_create(flat, 1)
cam = ffi.make_camera(origin, (0, 1, -1), fov, w / h, aperture, 4.0, *shutter)
st = ffi.make_settings(w, h, spp, max_path=max_path)
want = orc.render_region(scene, cam, st, 0, 0, w, h, seed=seed_val)
for name, img in imgs.items():
    diff = np.argwhere((bits(img[..., :3]) != bits(want[..., :3])).any(-1))
    print("%-13s vs oracle: %d pixels differ %s" % (name, len(diff), diff[:4].tolist()), flush=True)
    for (py, px) in diff[:4]:
        orc.render_region(scene, cam, st, int(px), int(py), 1, 1, seed=seed_val)
        print("     pixel", px, py, "gpu", img[py, px, :3], "oracle", want[py, px, :3], "oracle ties / hits outside own box:", orc.counters(scene)["closest_hit_ties"], orc.counters(scene)["hits_outside_own_box"], flush=True)
