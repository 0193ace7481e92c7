"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libref_seeded.so).

Run in the build container only (needs /root/reference to have been compiled into
oracle/_ref by `make -C oracle ref`):   python tests/golden/gen_golden.py
The fixtures are data -- inputs and the reference's outputs -- and travel to the
GPU box, where /root/reference does not exist.
"""
import os
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402
from helpers import ffi, scenes, objflat  # noqa: E402

SEED, BUILD_SEED = 1, 1


def main(only=None):
    """only: names of fixture files to (re)generate (default all) -- np.savez output is not byte-reproducible (zip time stamps),
    so unchanged fixtures are better left alone."""
    def want(name):
        return only is None or name in only

    ref = ffi.load_ref(True)
    assert ref is not None and ref.lib.ref_is_seeded() == 1, "build oracle/_ref first (make -C oracle ref)"
    orc = ffi.load_oracle()   # only for the MTL rule inside objflat (not reference-pinned; see objflat.py)
    tmp = tempfile.mkdtemp()

    # ---- image-level goldens ------------------------------------------------------
    for name in helpers.CASES:
        if not want(name):
            continue
        obj, c, flat = helpers.flat_for_case(name, tmp, orc)
        scene = ref.scene_create(flat, BUILD_SEED)
        cam = helpers.camera_for_case(c)
        out = {"triangles": flat.triangles, "materials": flat.materials}
        for spp in (1, 4, 16):
            img, smp = ref.render(scene, cam, ffi.make_settings(64, 64, spp), seed=SEED, want_samples=True)
            out["mode0_spp%d" % spp] = img
            if spp == 4:
                out["mode0_spp4_samples"] = smp
        for mode in (1, 2, 4, 5):
            out["mode%d" % mode] = ref.render(scene, cam, ffi.make_settings(64, 64, 1, mode=mode), seed=SEED)
        # non-square, not a multiple of 8, deeper paths
        out["mode0_40x28_spp3_len8"] = ref.render(scene, ffi.make_camera(c["origin"], c["look_at"], c["fov"], 40 / 28, c["aperture"], c["focal"], *c["shutter"]),
                                                  ffi.make_settings(40, 28, 3, max_path=8), seed=7)
        rays = helpers.random_rays(2048, 11, extent=1.5)
        rays[:, 1] += 1.0
        out["hit_rays"] = rays
        out["hits"] = ref.closest_hit(scene, rays, 1e-4)
        # per-material scatter records on this scene's materials
        rng = np.random.RandomState(5)
        rec = np.zeros((128, 16), np.float32)
        d = rng.normal(size=(128, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        n = rng.normal(size=(128, 3)); n /= np.linalg.norm(n, axis=1, keepdims=True)
        n[np.sum(n * d, axis=1) > 0] *= -1     # face the ray
        rec[:, 0:3] = rng.uniform(-1, 1, (128, 3)); rec[:, 3:6] = d; rec[:, 6] = 0.0
        rec[:, 7] = 1.0; rec[:, 8:11] = rng.uniform(-1, 1, (128, 3)); rec[:, 11:14] = n
        rec[:, 14:16] = rng.uniform(-0.5, 1.5, (128, 2))
        out["scatter_in"] = rec
        for mi in range(len(flat.materials)):
            out["scatter_mat%d" % mi] = ref.scatter(scene, mi, rec, seed=SEED)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "done", {k: v.shape for k, v in out.items() if k.startswith("mode0_spp")})

    # ---- BASELINE configs[0]: Cornell box 256x256, 4 spp, the reference's CPU path ---------------
    if want("config0"):
        obj, c, flat = helpers.flat_for_case("cornell", tmp, orc)
        scene = ref.scene_create(flat, BUILD_SEED)
        img = ref.render(scene, helpers.camera_for_case(c), ffi.make_settings(256, 256, 4), seed=SEED)
        np.savez_compressed(os.path.join(HERE, "config0.npz"), mode0_256x256_spp4=img)
        print("config0 done", img.shape, float(img[..., :3].mean()))

    # ---- windows of bench.py's timed frames (its N = 1 frame check reads this file; it is data, the oracle is not involved there) ----
    if want("bench_windows"):
        out = {}
        for wl, gen, kw, camname, w, h, spp, wins in (
                ("cornell_1080p_64spp", scenes.cornell, {}, "cornell", 1920, 1080, 64, ((952, 536), (700, 300), (1100, 800), (0, 0), (860, 200), (600, 500), (1300, 600), (1000, 900), (800, 750), (1200, 250))),
                ("breakfast_300k_1080p_128spp", scenes.cornell, dict(tess=91, displace_fraction=0.2), "breakfast", 1920, 1080, 128,
                 ((952, 536), (760, 340), (1150, 700), (800, 650), (1100, 380), (300, 300), (1000, 560), (900, 420), (1050, 640))),
                # the same scene seen from inside (round 4): every window is geometry
                ("breakfast_interior_300k_1080p_128spp", scenes.cornell, dict(tess=91, displace_fraction=0.2), "breakfast_interior", 1920, 1080, 128,
                 ((952, 536), (100, 100), (1800, 60), (40, 1000), (1850, 1040), (600, 300), (1300, 760), (480, 880), (1500, 200)))):
            c = scenes.CONFIG_CAMERAS[camname]
            obj, _ = gen(os.path.join(tmp, wl + ".obj"), **kw)
            flat = objflat.load_obj(obj, orc, sun_illuminance=c["sun"], sun_direction=c["sun_dir"])
            sr, so = ref.scene_create(flat, BUILD_SEED), orc.scene_create(flat, BUILD_SEED)
            cam = ffi.make_camera(c["origin"], c["look_at"], c["fov"], w / h)
            st = ffi.make_settings(w, h, spp)
            pos, px = [], []
            for (x0, y0) in wins:
                a = ref.render_region(sr, cam, st, x0, y0, 16, 16, seed=SEED)
                b = orc.render_region(so, cam, st, x0, y0, 16, 16, seed=SEED)
                cn = orc.counters(so)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
                # only windows in which no sample met two surfaces at exactly the same t: there the reference has ONE answer
                if cn["closest_hit_ties"] == 0 and cn["hits_outside_own_box"] == 0:
                    pos.append((x0, y0)); px.append(a)
                else:
                    print("   window", x0, y0, "of", wl, "left out:", cn["closest_hit_ties"], "ties")
            out[wl + "_pos"] = np.asarray(pos, np.int32); out[wl + "_px"] = np.asarray(px, np.float32)
            print("bench windows", wl, len(pos), "of", len(wins))
        np.savez_compressed(os.path.join(HERE, "bench_windows.npz"), **out)

    # ---- ... and of the textured / alpha-cut-out workload (round 5), a file of its own so that the fixtures above stay untouched ----
    if want("bench_windows_textured"):
        out = {}
        for wl, camname, w, h, spp, wins in (
                ("breakfast_textured_interior_300k_1080p_128spp", "breakfast_interior", 1920, 1080, 128,
                 ((952, 536), (100, 100), (1800, 60), (40, 1000), (1850, 1040), (600, 300), (1300, 760), (480, 880), (1500, 200))),):
            c = scenes.CONFIG_CAMERAS[camname]
            obj, _ = scenes.textured(os.path.join(tmp, wl + ".obj"), tess=91, displace_fraction=0.2)
            flat = objflat.load_obj(obj, orc, texture_loader=helpers.texture_loader, sun_illuminance=c["sun"], sun_direction=c["sun_dir"])
            assert len(flat.textures) == 4
            sr, so = ref.scene_create(flat, BUILD_SEED), orc.scene_create(flat, BUILD_SEED)
            cam = ffi.make_camera(c["origin"], c["look_at"], c["fov"], w / h)
            st = ffi.make_settings(w, h, spp)
            pos, px = [], []
            for (x0, y0) in wins:
                a = ref.render_region(sr, cam, st, x0, y0, 16, 16, seed=SEED)
                b = orc.render_region(so, cam, st, x0, y0, 16, 16, seed=SEED)
                cn = orc.counters(so)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
                if cn["closest_hit_ties"] == 0 and cn["hits_outside_own_box"] == 0:
                    pos.append((x0, y0)); px.append(a)
                else:
                    print("   window", x0, y0, "of", wl, "left out:", cn["closest_hit_ties"], "ties")
            out[wl + "_pos"] = np.asarray(pos, np.int32); out[wl + "_px"] = np.asarray(px, np.float32)
            print("bench windows", wl, len(pos), "of", len(wins))
        np.savez_compressed(os.path.join(HERE, "bench_windows_textured.npz"), **out)

    # ---- statistics of the UNTOUCHED reference (oracle/_ref/libref_native.so: its own std::random_device RNG, core/random.h:17-29) ----
    # Every other fixture goes through the determinism overlay (oracle/ref_shim/core/random.h).  This one pins the link the overlay
    # cannot: that the seeded stream contract samples the same distribution as the reference's own generator in the reference's own
    # draw order (core/random.cc:3-50, renderer.cc:210-248).  RUNS independent renders of SPP samples each; per pixel and channel the
    # mean of the run means and their standard deviation (ddof = 1) -> standard error of the mean = std / sqrt(RUNS).
    if want("native_stats"):
        native = ffi.load_ref(False)
        assert native is not None, "build oracle/_ref first (make -C oracle ref)"
        RUNS, SPP, W = 16, 512, 64
        out = {"runs": np.array(RUNS), "spp_per_run": np.array(SPP)}
        for name in ("cornell", "cornell_glass_sun", "pbr_maps", "cutout_sky"):
            obj, c, flat = helpers.flat_for_case(name, tmp, orc)
            scene = native.scene_create(flat, BUILD_SEED)
            cam = helpers.camera_for_case(c)
            st = ffi.make_settings(W, W, SPP)
            runs = np.stack([native.render_native(scene, cam, st)[..., :3].astype(np.float64) for _ in range(RUNS)])
            out[name + "_mean"] = runs.mean(0).astype(np.float32)
            out[name + "_std"] = runs.std(0, ddof=1).astype(np.float32)
            # what the seeded contract gives on the same statistic (the GPU must produce exactly these bits): printed, not stored
            seeded_scene = ref.scene_create(flat, BUILD_SEED)
            seeded = np.stack([ref.render(seeded_scene, cam, st, seed=SEED + k) for k in range(RUNS)])
            print("native_stats", name, helpers.z_statistics(seeded, out[name + "_mean"], out[name + "_std"], RUNS))
        np.savez_compressed(os.path.join(HERE, "native_stats.npz"), **out)

    # ---- long paths (round 5): maxPathLength 32 and 200 -- the GUI allows up to 1024 (gui-app/gui-app/MainForm.Designer.cs:140), the cut is
    # renderer.cc:120-123.  A closed-ish Cornell box keeps every path alive to the cut (MicrofacetMaterial::Scatter always returns true, material.cc:339), the
    # glass / mirror variant adds Dielectric and Mirror vertices, the procedural scene Metal / DiffuseLight / spheres / a moving cube.  64 x 64 x 4 spp each.
    if want("long_paths"):
        out = {}
        for name in ("cornell", "cornell_glass_sun", "pbr_maps"):
            obj, c, flat = helpers.flat_for_case(name, tmp, orc)
            scene = ref.scene_create(flat, BUILD_SEED)
            cam = helpers.camera_for_case(c)
            for depth in (32, 200):
                out["%s_len%d" % (name, depth)] = ref.render(scene, cam, ffi.make_settings(64, 64, 4, max_path=depth), seed=SEED)
                print("long_paths", name, depth, float(np.nanmean(out["%s_len%d" % (name, depth)][..., :3])))
        flat, c = helpers.procedural_flat()
        scene = ref.scene_create(flat, BUILD_SEED)
        cam = ffi.make_camera(c["origin"], c["look_at"], c["fov"], c["aspect"], c["aperture"], c["focal"], *c["shutter"])
        for depth in (32, 200):
            out["procedural_len%d" % depth] = ref.render(scene, cam, ffi.make_settings(96, 64, 4, max_path=depth), seed=SEED)
            print("long_paths procedural", depth, float(np.nanmean(out["procedural_len%d" % depth][..., :3])))
        # the hall of mirrors (scenes.mirror_hall): nearly every path reaches the cut -- also at the GUI's maximum, 1024
        obj, _ = scenes.mirror_hall(os.path.join(tmp, "mirror_hall.obj"))
        c = scenes.MIRROR_HALL_CAMERA
        flat = objflat.load_obj(obj, orc, sun_illuminance=c["sun"], sun_direction=c["sun_dir"])
        scene = ref.scene_create(flat, BUILD_SEED)
        cam = ffi.make_camera(c["origin"], c["look_at"], c["fov"], 1.0)
        for depth in (5, 32, 200, 1024):
            out["mirror_hall_len%d" % depth] = ref.render(scene, cam, ffi.make_settings(64, 64, 4, max_path=depth), seed=SEED)
            print("long_paths mirror_hall", depth, float(np.nanmean(out["mirror_hall_len%d" % depth][..., :3])))
        np.savez_compressed(os.path.join(HERE, "long_paths.npz"), **out)

    if only is not None and not (want("procedural") or want("kat") or want("soup")):
        return
    # ---- procedural scene: spheres, a moving cube, Metal / DiffuseLight / Dielectric / Mirror -----------------------
    flat, c = helpers.procedural_flat()
    scene = ref.scene_create(flat, BUILD_SEED)
    cam = ffi.make_camera(c["origin"], c["look_at"], c["fov"], c["aspect"], c["aperture"], c["focal"], *c["shutter"])
    out = {}
    for spp in (1, 4, 16):
        out["mode0_spp%d" % spp] = ref.render(scene, cam, ffi.make_settings(96, 64, spp), seed=SEED)
    for mode in (1, 2, 5):
        out["mode%d" % mode] = ref.render(scene, cam, ffi.make_settings(96, 64, 1, mode=mode), seed=SEED)
    rays = helpers.random_rays(2048, 12, extent=2.0)
    out["hit_rays"] = rays
    out["hits"] = ref.closest_hit(scene, rays, 1e-4)
    np.savez_compressed(os.path.join(HERE, "procedural.npz"), **out)
    print("procedural done")

    # ---- function-level known answers ------------------------------------------------
    rng = np.random.RandomState(17)
    kat = {}
    n = 2000
    boxes = np.sort(rng.uniform(-2, 2, (n, 2, 3)).astype(np.float32), axis=1).reshape(n, 6)
    boxes[::7, 4] = boxes[::7, 1]                        # flat boxes (tMax == tMin must pass)
    rays = helpers.random_rays(n, 18, extent=3.0)
    rays[::5, 3] = 0.0; rays[::11, 4] = 0.0              # axis-parallel directions (inf / NaN slabs)
    rays[::13, 1] = boxes[::13, 1]                       # origin on a slab plane
    kat["aabb_boxes"], kat["aabb_rays"] = boxes, rays
    kat["aabb_out"] = ref.aabb_hit(boxes, rays, 1e-4, 3.4028234663852886e38)
    tris = np.zeros(n, ffi.TRI_DTYPE)
    P = rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32)
    tris["v0"], tris["v1"], tris["v2"] = P[:, 0], P[:, 1], P[:, 2]
    N = rng.normal(size=(n, 3, 3)).astype(np.float32)
    tris["n0"], tris["n1"], tris["n2"] = N[:, 0], N[:, 1], N[:, 2]
    tris["st"] = rng.uniform(-2, 2, (n, 6)).astype(np.float32)
    trays = helpers.random_rays(n, 19, extent=1.0)
    target = (P.mean(axis=1) + rng.normal(scale=0.3, size=(n, 3))).astype(np.float32)
    dd = target - trays[:, :3]; dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    trays[:, 3:] = dd.astype(np.float32)
    kat["tri_tris"], kat["tri_rays"] = tris, trays
    kat["tri_out"] = ref.triangle_hit(tris, trays, 1e-4, 3.4028234663852886e38)
    nn = rng.normal(size=(512, 3)).astype(np.float32); nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    nn[:8] = np.eye(3, dtype=np.float32)[[0, 1, 2, 0, 1, 2, 0, 1]] * np.array([1, 1, 1, -1, -1, -1, 1, 1], np.float32)[:, None]
    vv = rng.normal(size=(512, 3)).astype(np.float32)
    kat["onb_n"], kat["onb_v"] = nn, vv
    kat["onb_local"], kat["onb_world"] = ref.onb(nn, vv)
    cam = ffi.make_camera((0.3, 1.2, 4), (0, 0.9, -1), 50.0, 1.5, 0.1, 3.0, 0.0, 2.0)
    uv = rng.uniform(0, 1, (512, 2)).astype(np.float32)
    kat["cam_uv"] = uv
    kat["cam_rays"] = ref.camera_rays(cam, uv, seed=3)
    cam2 = ffi.make_camera((0, 5, 0), (0, 0, 0), 60.0, 1.0)     # looking straight down: the up-vector switch (camera.h:62-66)
    kat["cam2_rays"] = ref.camera_rays(cam2, uv, seed=3)
    tex = scenes.texture_as_float(scenes.leaf_texture())
    tuv = rng.uniform(-3, 3, (1024, 2)).astype(np.float32)
    tuv[:16] = [[0, 0], [1, 1], [0.999999, 0.5], [-0.0, 0.25], [1.0, 0.0], [0.5, 1.0], [2.0, -1.0], [-1e-8, 1e-8]] * 2
    kat["tex_uv"] = tuv
    kat["tex_linear"] = ref.texture_sample(tex, 0, tuv)
    kat["tex_srgb"] = ref.texture_sample(tex, 1, tuv)
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **kat)

    # ---- closest hit on a 10k-triangle soup ------------------------------------------
    obj, _ = scenes.soup(os.path.join(tmp, "soup.obj"), 10000)
    flat = objflat.load_obj(obj, orc)
    scene = ref.scene_create(flat, BUILD_SEED)
    rays = helpers.random_rays(4096, 23, extent=4.5)
    hits = ref.closest_hit(scene, rays, 1e-4)
    nodes, depth = ref.bvh_stats(scene)
    np.savez_compressed(os.path.join(HERE, "soup.npz"), rays=rays, hits=hits, bvh=np.array([nodes, depth]))
    print("soup: %d / %d rays hit; reference BVH nodes %d depth %d" % (hits["hit"].sum(), len(hits), nodes, depth))


if __name__ == "__main__":
    main(set(sys.argv[1:]) or None)
