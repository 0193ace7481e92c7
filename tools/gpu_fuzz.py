"""Schedule fuzz: random scenes / cameras / settings rendered under every schedule of the megakernel; all must give the same bits
and the same ray and shading counts.  usage: python tools/gpu_fuzz.py [cases=60] [seed=1]   (run it under `timeout`)"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "software-raytracing_amd"))
from raylib_amd import binding, scenes
lib = binding.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
d = tempfile.mkdtemp()
MODES = [dict(RAYLIB_POOL="0"), dict(), dict(RAYLIB_BVH4="0"), dict(RAYLIB_POOL_SHORT_STACK="0"), dict(RAYLIB_BVH4="0", RAYLIB_POOL_SHORT_STACK="4"),
         dict(RAYLIB_POOL="3", RAYLIB_BVH4="0"), dict(RAYLIB_POOL="2", RAYLIB_SAMPLE_BATCH="1"),
         dict(RAYLIB_POOL="0", RAYLIB_LEAF_LIST="0"), dict(RAYLIB_POOL="0", RAYLIB_LDS_SCENE="0"),
         dict(RAYLIB_JOB_HEADS="1"), dict(RAYLIB_POOL="0", RAYLIB_JOB_HEADS="3", RAYLIB_JOB_CHUNK="64"),
         dict(RAYLIB_CULL_CELLS="0"),   # round 3: one head; three heads with the smallest chunks (every wave steals)
         dict(RAYLIB_POOL="2", RAYLIB_BVH8="1")]   # round 4: the 8-wide tree (scenes above 108 triangles carry one; these are too shallow to get it by default)
if os.environ.get("FUZZ_MODES_JSON"): import json; MODES = json.loads(os.environ["FUZZ_MODES_JSON"])   # debugging: the schedules to run, e.g. '[{"RAYLIB_POOL": "0"}]'
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
bad = 0
only = int(os.environ.get("FUZZ_ONLY", "-1"))
t0 = time.time()
for case in range(cases):
    kind = rng.randint(int(os.environ.get("FUZZ_KINDS", "6")))   # FUZZ_KINDS=5 reproduces the case list of the runs recorded before the small-soup kind existed
    if os.environ.get("FUZZ_FORCE_KIND"): kind = int(os.environ["FUZZ_FORCE_KIND"])   # e.g. 5: only small soups (the draw above still happens: same stream)
    if kind == 0:
        gen = (scenes.cornell, dict(tess=int(rng.randint(1, 20)), displace_fraction=float(rng.choice([0.0, 0.1, 0.3])),
                                    tall_material=str(rng.choice([scenes.MIRROR, scenes.GLASS, scenes.WHITE])), short_material=str(rng.choice([scenes.WHITE, scenes.GLASS]))))
    elif kind == 1:
        gen = (scenes.soup, dict(n_tris=int(rng.randint(10, 30000)), seed=int(rng.randint(1 << 30)), extent=float(rng.uniform(1, 5)), size=float(rng.uniform(0.05, 1.0))))
    elif kind == 2:
        gen = (scenes.cutout, dict(tess=int(rng.randint(1, 12))))
    elif kind == 3:
        gen = (scenes.colonnade, dict(tess=int(rng.randint(1, 4))))
    elif kind == 4:
        gen = (scenes.pbr_maps, dict(tess=int(rng.randint(1, 10)), mtl=scenes.random_pbr_mtl(rng)))
    else:   # small soups: the LDS-resident class (leaf list up to 108 triangles, BVH4 in LDS up to 128)
        gen = (scenes.soup, dict(n_tris=int(rng.randint(8, 140)), seed=int(rng.randint(1 << 30)), extent=float(rng.uniform(0.5, 3)), size=float(rng.uniform(0.05, 2.0))))
    sun = (0, 0, 0) if rng.rand() < 0.4 else tuple(float(x) for x in rng.uniform(1, 20, 3))
    sun_dir = tuple(float(x) for x in rng.uniform(-1, 1, 3) * np.array([1, 1, 1]) + np.array([0, -1.2, 0]))
    sky = scenes.sky_panorama() if rng.rand() < 0.4 else None
    origin = tuple(float(x) for x in np.array([0, 1, 4]) + rng.uniform(-1.5, 1.5, 3))
    w, h = int(rng.randint(9, 200)), int(rng.randint(9, 120))
    spp, max_path = int(rng.choice([1, 2, 5, 8])), int(rng.choice([1, 2, 5, 9]))
    aperture = 0.0 if rng.rand() < 0.6 else float(rng.uniform(0.01, 0.2))
    seed_val = int(rng.randint(1, 1 << 30))
    tmin = float(rng.choice([1e-4, 1e-4, 1e-4, 0.0, 1e-2, -0.05])) if os.environ.get("FUZZ_TMIN") else 1e-4   # FUZZ_TMIN=1: also rayTMin 0, 1e-2 and a negative one
    fov, shutter_end = float(rng.uniform(30, 80)), float(rng.choice([0.0, 1.0]))   # drawn before the skip: FUZZ_ONLY must see the full run's stream
    if only >= 0 and case != only: continue
    if os.environ.get("FUZZ_TMIN_OVERRIDE"): tmin = float(os.environ["FUZZ_TMIN_OVERRIDE"])
    obj, n = gen[0](os.path.join(d, "f%d.obj" % case), **gen[1])
    ses = binding.SceneSession(lib, obj, origin, (0, 1, -1), fov, w / h, sun=sun, sun_dir=sun_dir, sky_image=sky,
                               aperture=aperture, focal=4.0, shutter=(0.0, shutter_end))
    lib.RaylibAMD_SetSeed(seed_val)
    ref = None
    for env in MODES:
        for k, v in env.items(): os.environ[k] = v
        if os.environ.get("FUZZ_VERBOSE"): print("case %d kind %d tris %d %dx%d spp %d len %d tmin %g env %s" % (case, kind, n, w, h, spp, max_path, tmin, env), flush=True)
        img = ses.render(w, h, spp, max_path=max_path, tmin=tmin)
        st = ses.stats().as_dict()
        for k in env: del os.environ[k]
        # cut-out scenes run the alpha test on traversal candidates, whose number depends on the order candidates are met in:
        # their shading / texel counts are schedule-dependent, rays and samples are not
        key = (st["frameRays"], st["frameSamples"]) if kind in (2, 4) else (st["frameRays"], st["shadedHits"], st["frameSamples"], st["texFetches"])   # (frame totals: executed + culled)
        if ref is None: ref = (img, key)
        elif not (np.array_equal(bits(img), bits(ref[0])) and key == ref[1]):
            bad += 1
            dpx = np.argwhere((bits(img) != bits(ref[0])).any(-1))
            print("MISMATCH case %d kind %d tris %d %dx%d spp %d len %d env %s: %d pixels differ %s, counts %s vs %s" % (
                case, kind, n, w, h, spp, max_path, env, len(dpx), dpx[:3].tolist(), key, ref[1]), flush=True)
            if True:
                for (py, px) in dpx[:1]: print("   pixel", px, py, img[py, px, :3], "vs", ref[0][py, px, :3], "case params", gen[1], origin, sun, sun_dir, aperture, flush=True)
    ses.close()
    if case % 10 == 9: print("case %d done (%.0f s), mismatches so far %d" % (case + 1, time.time() - t0, bad), flush=True)
lib.RaylibAMD_SetSeed(1)
print("fuzz: %d cases x %d schedules, mismatches: %d" % (cases, len(MODES), bad))
sys.exit(1 if bad else 0)
