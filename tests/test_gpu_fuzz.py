"""Randomised parity: scenes, cameras and settings that are in no fixture, HIP path (default schedule per scene) against the CPU
oracle on the same pixel keys.  A pixel may differ only if one of its samples met two surfaces at exactly the same t (reference-
undefined): the oracle's tie counter must fire for it."""
import os
import numpy as np
import pytest

import helpers
from helpers import ffi, bits, scenes, objflat

pytestmark = pytest.mark.gpu


TIE_BUDGET = {11: 4, 12: 4, 13: 4,   # measured on MI355X in round 2: 0 / 0 / 0 excused pixels of 4 936 / 6 003 / 7 410
              21: 4, 22: 4}          # seeds from 20 on: small soups only (8..127 triangles: the LDS-resident class, walked through the leaf list up to 108)


def _l2(a, b):
    return float(np.sqrt(((a[..., :3].astype(np.float64) - b[..., :3]) ** 2).sum(-1)).max())


@pytest.mark.parametrize("seed", [11, 12, 13, 21, 22])
def test_random_scenes_against_oracle(seed, gpu_lib, oracle, workdir, monkeypatch):
    from raylib_amd import binding
    rng = np.random.RandomState(seed)
    d = os.path.join(str(workdir), "fuzz%d" % seed); os.makedirs(d, exist_ok=True)
    total = tied = 0
    for case in range(8):
        kind = rng.randint(4)
        if seed >= 20:
            kind = 1
            obj, n = scenes.soup(os.path.join(d, "f%d.obj" % case), n_tris=int(rng.randint(8, 128)), seed=int(rng.randint(1 << 30)), extent=float(rng.uniform(0.5, 3)), size=float(rng.uniform(0.05, 2.0)))
        elif kind == 0:
            obj, n = scenes.cornell(os.path.join(d, "f%d.obj" % case), tess=int(rng.randint(1, 9)), displace_fraction=float(rng.choice([0.0, 0.2])),
                                    tall_material=str(rng.choice([scenes.MIRROR, scenes.GLASS, scenes.WHITE])), short_material=str(rng.choice([scenes.WHITE, scenes.GLASS])))
        elif kind == 1:
            obj, n = scenes.soup(os.path.join(d, "f%d.obj" % case), n_tris=int(rng.randint(10, 3000)), seed=int(rng.randint(1 << 30)), extent=float(rng.uniform(1, 3)), size=float(rng.uniform(0.1, 1.0)))
        elif kind == 2:
            obj, n = scenes.cutout(os.path.join(d, "f%d.obj" % case), tess=int(rng.randint(1, 6)))
        else:
            obj, n = scenes.pbr_maps(os.path.join(d, "f%d.obj" % case), tess=int(rng.randint(1, 5)), mtl=scenes.random_pbr_mtl(rng))
        sun = (0, 0, 0) if rng.rand() < 0.4 else tuple(float(x) for x in rng.uniform(1, 20, 3))
        sun_dir = tuple(float(x) for x in rng.uniform(-1, 1, 3) + np.array([0, -1.2, 0]))
        sky = rng.rand() < 0.4
        origin = tuple(float(x) for x in np.array([0, 1, 4]) + rng.uniform(-1.0, 1.0, 3))
        fov = float(rng.uniform(30, 80))
        w, h = int(rng.randint(9, 48)), int(rng.randint(9, 40))
        spp, max_path = int(rng.choice([1, 3, 6])), int(rng.choice([1, 3, 7]))
        aperture = 0.0 if rng.rand() < 0.6 else float(rng.uniform(0.01, 0.2))
        shutter = (0.0, float(rng.choice([0.0, 1.0])))
        seed_val = int(rng.randint(1, 1 << 30))
        ses = binding.SceneSession(gpu_lib, obj, origin, (0, 1, -1), fov, w / h, sun=sun, sun_dir=sun_dir, aperture=aperture, focal=4.0, shutter=shutter,
                                   sky_image=scenes.sky_panorama() if sky else None)
        gpu_lib.RaylibAMD_SetSeed(seed_val)
        img = ses.render(w, h, spp, max_path=max_path)
        if n > 108:
            # these scenes are too shallow for the 8-wide tree to be the default (RaylibAMD_SceneBVH8Info): force it, same bits required
            monkeypatch.setenv("RAYLIB_BVH8", "1"); monkeypatch.setenv("RAYLIB_POOL", "2")
            img8 = ses.render(w, h, spp, max_path=max_path)
            assert ses.stats().treeWidth == 8 and ses.stats().nodeBytes == 80
            monkeypatch.delenv("RAYLIB_BVH8"); monkeypatch.delenv("RAYLIB_POOL")
            assert ((bits(img8) == bits(img)) | (np.isnan(img8) & np.isnan(img))).all(), "case %d: the 8-wide walk differs from the default schedule" % case
        gpu_lib.RaylibAMD_SetSeed(1)
        ses.close()
        flat = objflat.load_obj(obj, oracle, texture_loader=helpers.texture_loader, sun_illuminance=sun, sun_direction=sun_dir)
        if sky:
            flat.textures.append(np.ascontiguousarray(scenes.sky_panorama(), np.float32)); flat.sky_texture = len(flat.textures) - 1
        scene = oracle.scene_create(flat, 1)
        cam = ffi.make_camera(origin, (0, 1, -1), fov, w / h, aperture, 4.0, *shutter)
        st = ffi.make_settings(w, h, spp, max_path=max_path)
        want = oracle.render_region(scene, cam, st, 0, 0, w, h, seed=seed_val)
        # random MTL constants can drive the reference to NaN (e.g. roughness -> 0 with a grazing normal map): NaN must meet NaN
        differ = ~((bits(img[..., :3]) == bits(want[..., :3])) | (np.isnan(img[..., :3]) & np.isnan(want[..., :3]))).all(-1)
        for (py, px) in zip(*np.nonzero(differ)):
            oracle.render_region(scene, cam, st, int(px), int(py), 1, 1, seed=seed_val)
            cn = oracle.counters(scene)
            assert cn["closest_hit_ties"] > 0 or cn["hits_outside_own_box"] > 0, \
                "case %d (%s, %d triangles, %dx%d, spp %d, len %d): pixel %d,%d differs without a closest-hit tie: %s vs %s" % (
                    case, ("room", "soup", "cutout", "pbr")[kind], n, w, h, spp, max_path, px, py, img[py, px, :3], want[py, px, :3])
            tied += 1
        ok = np.isfinite(want[..., :3]).all(-1) & np.isfinite(img[..., :3]).all(-1)
        total += w * h
    # every differing pixel was shown above to have a closest-hit tie (or an own-box event) among its samples; how many there were is
    # reported, and bounded by what these seeds measure (TIE_BUDGET) plus a margin of four pixels -- not by a percentage of the image
    print("seed %d: %d excused tie pixels of %d" % (seed, tied, total))
    assert tied <= TIE_BUDGET[seed], "%d tie pixels of %d" % (tied, total)
