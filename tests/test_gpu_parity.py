"""GPU parity tests beyond the contract tier (tests/test_gpu_00_contract.py, tests/test_gpu_01_configs.py): the corner cases the rounds
added -- rays through vertices and along edges, degenerate / tiny / huge triangles, the silhouette cull, the leaf list with every leaf a
candidate, the pool schedule's stack variants on a deep tree, bench.py's four ways to run, deferred read-back, the sky panorama swap, the
reference-undefined AOV modes.  Same rules: through the C-ABI, bit-exact against reference fixtures and the oracle, and no assertion on a
time or on a schedule-dependent counter (such numbers are printed)."""
import ctypes as C
import os
import numpy as np
import pytest

import helpers
from helpers import ffi, bits, l2, frac_bit_equal, window_mismatches_without_a_tie, tie_mask, assert_same_outside_ties, golden, L2_TOL, FLT_MAX   # noqa: F401

pytestmark = pytest.mark.gpu


def test_rays_through_vertices_and_along_edges_short_barycentrics_against_the_divisions(gpu_lib, workdir, oracle, monkeypatch):
    """The triangle test's two divisions by `denom` run as six-cycle sequences with the reciprocal from the triangle record (csrc/rl_render.hip
    Barycentric), and a quotient below 2^-38 -- a ray through a vertex, along an edge, or past one by a hair -- sends the lane to the divisions
    themselves.  Random rays never get there; these do: aimed at every vertex, at points ON every edge (exact in float for the axis-aligned
    Cornell walls) and a few ulps to either side.  The same scene uploaded with RAYLIB_FAST_BARY=0 (divisions only) must give the same records
    bit for bit, and both must be the CPU oracle's."""
    from raylib_amd import binding
    obj, c = helpers.build_case("cornell", workdir)
    fast = binding.SceneSession(gpu_lib, obj, c["origin"], c["look_at"], c["fov"], c["aspect"])
    monkeypatch.setenv("RAYLIB_FAST_BARY", "0")
    slow = binding.SceneSession(gpu_lib, obj, c["origin"], c["look_at"], c["fov"], c["aspect"])
    monkeypatch.delenv("RAYLIB_FAST_BARY")
    tris, _ = fast.export_flat()
    rng = np.random.RandomState(21)
    targets = [tris["v0"], tris["v1"], tris["v2"]]
    for a, b in (("v0", "v1"), ("v1", "v2"), ("v2", "v0")):
        for w in (0.5, 0.25, 0.125, 0.75):                      # on the edge, exactly, where the coordinates allow
            targets.append((tris[a].astype(np.float64) * (1 - w) + tris[b].astype(np.float64) * w).astype(np.float32))
    targets = np.concatenate(targets)
    nudged = [targets]
    for k in (1, -1, 3, -3):                                     # ... and a few ulps off in every coordinate
        nudged.append((targets.view(np.int32) + k).view(np.float32))
    targets = np.concatenate(nudged)
    targets = targets[np.isfinite(targets).all(1)]
    rays = []
    for origin in ((0.0, 1.0, 4.0), (0.1, 0.9, 0.3), (-0.4, 1.6, -0.2)):
        o = np.broadcast_to(np.asarray(origin, np.float32), targets.shape)
        rays.append(np.concatenate([o, (targets - o).astype(np.float32)], axis=1))     # unnormalised: t = 1 at the target
        d = (targets - o).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        rays.append(np.concatenate([o, d], axis=1))
    rays = np.ascontiguousarray(np.concatenate(rays), np.float32)
    rays = rays[np.isfinite(rays).all(1)]
    outs = []
    for ses in (fast, slow):
        out = np.zeros(len(rays), ffi.HIT_DTYPE)
        assert gpu_lib.RaylibAMD_ClosestHit(ses.scene, rays.ctypes.data_as(C.POINTER(C.c_float)), len(rays), 1e-4, out.ctypes.data) == 1
        outs.append(out)
    raw = [o.view(np.uint8).reshape(len(rays), -1) for o in outs]
    assert outs[0].tobytes() == outs[1].tobytes(), "short barycentric form and divisions disagree on %d rays" % (raw[0] != raw[1]).any(1).sum()
    assert outs[0]["hit"].sum() > len(rays) // 2
    # the frame too, on both schedules' code (the leaf-list kernel reads the reciprocal from its LDS record)
    a, b = fast.render(64, 64, 4), slow.render(64, 64, 4)
    assert np.array_equal(bits(a), bits(b))
    # and the oracle.  A ray through an edge or a vertex meets two or more triangles within an ulp of the same distance, on the boundary of each: the
    # reference's barycentric test lets some of them in although the ray misses that triangle's own box, and which one its traversal returns
    # depends on its tree (the device only takes candidates whose own box the ray passes).  The oracle counts those events:
    # every ray whose record differs must have met one.
    _, _, flat = helpers.flat_for_case("cornell", workdir, oracle)
    scene = oracle.scene_create(flat, 1)
    want = oracle.closest_hit(scene, rays, 1e-4)
    differ = np.nonzero((outs[0]["hit"] != want["hit"]) | (bits(outs[0]["t"]) != bits(want["t"])))[0]
    for i in differ:
        oracle.closest_hit(scene, rays[i:i + 1], 1e-4)
        cn = oracle.counters(scene)
        assert cn["closest_hit_ties"] > 0 or cn["hits_outside_own_box"] > 0, "ray %d %s: device hit %d t %r | oracle hit %d t %r | %s" % (
            i, rays[i].tolist(), outs[0]["hit"][i], float(outs[0]["t"][i]), want["hit"][i], float(want["t"][i]), cn)
    print("%d rays at vertices / on edges: %d hits; %d differ from the oracle's tree walk, every one with a tie or a hit outside its triangle's own box on the way" % (
        len(rays), outs[0]["hit"].sum(), len(differ)))
    assert len(differ) < len(rays) // 20
    oracle.scene_destroy(scene)
    fast.close(); slow.close()


def test_triangles_too_small_or_too_large_for_the_short_barycentric_form(gpu_lib, workdir, oracle):
    """A divisor outside 2^-62 .. 2^125 (triangle edges below ~2e-5 or above ~2e9 scene units) or a degenerate triangle (denom 0): the first kind makes
    the whole scene take the divisions (DSceneView::fastBary), the second can never be hit either way.  Rays aimed into each of them: the oracle's
    records, bit for bit."""
    from raylib_amd import binding, scenes
    d = os.path.join(str(workdir), "extreme"); os.makedirs(d, exist_ok=True)
    for tag, special in (("ordinary", []),
                         ("degenerate", [((0.2, 0.2, 0.5), (0.2, 0.2, 0.5), (0.4, 0.3, 0.5)), ((0.0, 0.0, 0.6), (0.1, 0.1, 0.6), (0.2, 0.2, 0.6))]),
                         ("tiny", [((0.3, 0.3, 0.5), (0.3 + 6e-6, 0.3, 0.5), (0.3, 0.3 + 6e-6, 0.5))]),
                         ("huge", [((-3e9, -3e9, -7.0), (3e9, -3e9, -7.0), (0.0, 3e9, -7.0))])):
        tri = [((-1.0, -1.0, 0.0), (1.0, -1.0, 0.0), (0.0, 1.0, 0.0)), ((-1.0, -1.0, -2.0), (1.5, -1.0, -2.0), (0.0, 1.5, -2.5))] + special
        obj = os.path.join(d, tag + ".obj")
        with open(obj, "w") as f:
            f.write("mtllib %s.mtl\nusemtl white\n" % tag)
            for k, t in enumerate(tri):
                f.write("usemtl %s\n" % ("light" if k == 1 else "white"))      # something to see: the far triangle glows
                for v in t:
                    f.write("v %.9g %.9g %.9g\n" % v)
                f.write("f %d %d %d\n" % (3 * k + 1, 3 * k + 2, 3 * k + 3))
        with open(os.path.join(d, tag + ".mtl"), "w") as f:
            f.write(scenes.CORNELL_MTL)
        ses = binding.SceneSession(gpu_lib, obj, (0.0, 0.0, 3.0), (0.0, 0.0, 0.0), 45.0, 1.0)
        rng = np.random.RandomState(5)
        o = np.tile(np.asarray([[0.1, 0.2, 3.0]], np.float32), (4000, 1))
        tg = np.concatenate([rng.uniform(-1.2, 1.2, (3000, 3)) * (1, 1, 0), np.asarray(tri[-1][0]) + rng.uniform(-1, 1, (1000, 3)) * (1.2e-5, 1.2e-5, 0)]).astype(np.float32)
        rays = np.ascontiguousarray(np.concatenate([o, tg - o], axis=1), np.float32)
        out = np.zeros(len(rays), ffi.HIT_DTYPE)
        assert gpu_lib.RaylibAMD_ClosestHit(ses.scene, rays.ctypes.data_as(C.POINTER(C.c_float)), len(rays), 1e-4, out.ctypes.data) == 1
        flat = helpers.objflat.load_obj(obj, oracle)
        scene = oracle.scene_create(flat, 1)
        want = oracle.closest_hit(scene, rays, 1e-4)
        for f in ("hit", "t", "p", "n"):
            a, b = out[f], want[f]
            assert np.array_equal(bits(a) if a.dtype == np.float32 else a, bits(b) if b.dtype == np.float32 else b), (tag, f)
        img = ses.render(48, 48, 2)
        ref = oracle.render(scene, ffi.make_camera((0.0, 0.0, 3.0), (0.0, 0.0, 0.0), 45.0, 1.0), ffi.make_settings(48, 48, 2), seed=1)
        assert helpers.same(img[..., :3], ref[..., :3]).all(), tag
        print("%s: %d of %d rays hit" % (tag, out["hit"].sum(), len(rays)))
        oracle.scene_destroy(scene)
        ses.close()


def test_cells_outside_the_scenes_silhouette_are_not_traced_and_nothing_changes(gpu_lib, workdir, oracle, monkeypatch):
    """With a pinhole camera and no sky panorama, cells whose pixels (jitter and margin included) lie outside the projected bounding box of the scene are
    dropped from the megakernel's job list; k_resolve adds up the miss shader's constant for them (csrc/rl_runtime.inl CullCells).  The frame and the
    counters must be what they are with every cell traced (RAYLIB_CULL_CELLS=0) -- cameras far away (most cells dropped), close, inside the box's slab
    (nothing can be dropped), off-axis, with and without a sun -- and the oracle's on a window that straddles the silhouette."""
    from raylib_amd import binding
    obj, c = helpers.build_case("cornell", workdir)
    _, _, flat = helpers.flat_for_case("cornell", workdir, oracle)
    views = [((0.0, 1.0, 4.0), (0.0, 1.0, -1.0), 45.0), ((0.0, 1.0, 14.0), (0.0, 1.0, -1.0), 45.0), ((3.5, 2.5, 6.0), (0.0, 1.0, 0.0), 30.0),
             ((0.0, 1.0, 0.5), (0.0, 1.0, -1.0), 70.0), ((-6.0, 0.3, 0.0), (0.0, 1.0, 0.0), 25.0), ((0.0, 9.0, 0.01), (0.0, 0.0, 0.0), 40.0)]
    dropped_somewhere = 0
    probe = binding.SceneSession(gpu_lib, obj, (0.0, 1.0, 4.0), (0.0, 1.0, -1.0), 45.0, 1.0)
    tris, _ = probe.export_flat()
    probe.close()
    pts = np.concatenate([tris["v0"], tris["v1"], tris["v2"]])
    bounds = np.concatenate([pts.min(0), pts.max(0)]).astype(np.float32)      # the scene's box: what the root node's child boxes span
    for sun, sun_dir in (((0.0, 0.0, 0.0), (0.0, -1.0, -0.5)), ((9.0, 8.0, 7.0), (-1.0, -1.0, 0.0)), ((5.0, 5.0, 5.0), (0.0, -1.0, -0.9))):
        for (origin, look, fov) in views:
            ses = binding.SceneSession(gpu_lib, obj, origin, look, fov, 200 / 120, sun=sun, sun_dir=sun_dir)
            for (w, h, spp) in ((200, 120, 3), (67, 41, 1)):
                img = ses.render(w, h, spp)
                st1 = ses.stats().as_dict()
                monkeypatch.setenv("RAYLIB_CULL_CELLS", "0")
                ref = ses.render(w, h, spp)
                st0 = ses.stats().as_dict()
                monkeypatch.delenv("RAYLIB_CULL_CELLS")
                assert np.array_equal(bits(img), bits(ref)), (origin, sun, w, h)
                # executed + stood-for = the frame's totals, the same with and without the cull; shading events and triangle tests are all executed
                for k in ("frameRays", "frameSamples", "frameNodes", "shadedHits", "trisTested", "pixels"):
                    assert st0[k] == st1[k], (k, st0[k], st1[k], origin, sun)
                assert st1["frameSamples"] == w * h * spp and st0["cameraSamples"] == w * h * spp and st0["culledCells"] == 0 and st0["culledRays"] == 0
                # what was dropped is what the host-side rule (RaylibAMD_CullCells, same camera, same box, same sun) says: a fact, not a schedule
                cells = ((w + 7) // 8) * ((h + 7) // 8)
                flags = np.zeros(cells, np.uint8); const = np.zeros(3, np.float32)
                sun_v, sd = np.zeros(3, np.float32), np.zeros(3, np.float32)
                gpu_lib.RaylibAMD_SceneGetSun(ses.scene, sun_v.ctypes.data_as(C.POINTER(C.c_float)), sd.ctypes.data_as(C.POINTER(C.c_float)))   # the direction as the scene normalised it
                n = gpu_lib.RaylibAMD_CullCells(ses.camera, bounds.ctypes.data_as(C.POINTER(C.c_float)), sun_v.ctypes.data_as(C.POINTER(C.c_float)),
                                                sd.ctypes.data_as(C.POINTER(C.c_float)), w, h, flags.ctypes.data_as(C.POINTER(C.c_uint8)), const.ctypes.data_as(C.POINTER(C.c_float)))
                assert st1["culledCells"] == max(0, n) == int(flags.sum()) and st1["culledCells"] + st1["listedCells"] == cells, (st1["culledCells"], n, origin, sun, w, h)
                px = 0
                for cell in np.nonzero(flags)[0]:
                    cx, cy = cell % ((w + 7) // 8), cell // ((w + 7) // 8)
                    px += min(8, w - 8 * cx) * min(8, h - 8 * cy)
                assert st1["culledSamples"] == px * spp and st1["culledRays"] == px * spp * (2 if any(sun) else 1)
                assert st1["cameraSamples"] == w * h * spp - px * spp
                dropped_somewhere += int(st1["culledCells"] > 0)
            ses.close()
    # A thin lens (the console front-end's camera: aperture 0.01, reference src/main.cc:24; here 0.01 and a wide 0.3): the rectangle grows by the circle of
    # confusion, cells beyond it are dropped, and the frame and the totals are what they are with every cell traced.
    lens_dropped = 0
    for aperture, focal in ((0.01, 4.0), (0.3, 2.0), (0.3, 9.0)):
        for sun, sun_dir in (((0.0, 0.0, 0.0), (0.0, -1.0, -0.5)), ((9.0, 8.0, 7.0), (-1.0, -1.0, 0.0))):
            ses = binding.SceneSession(gpu_lib, obj, (0.5, 1.2, 9.0), (0.0, 1.0, -1.0), 40.0, 200 / 120, sun=sun, sun_dir=sun_dir, aperture=aperture, focal=focal)
            img = ses.render(200, 120, 4)
            st1 = ses.stats().as_dict()
            monkeypatch.setenv("RAYLIB_CULL_CELLS", "0")
            ref = ses.render(200, 120, 4)
            st0 = ses.stats().as_dict()
            monkeypatch.delenv("RAYLIB_CULL_CELLS")
            assert np.array_equal(bits(img), bits(ref)), (aperture, focal, sun)
            for k in ("frameRays", "frameSamples", "frameNodes", "shadedHits", "trisTested", "pixels"):
                assert st0[k] == st1[k], (k, st0[k], st1[k], aperture, focal, sun)
            lens_dropped += int(st1["culledCells"] > 0)
            ses.close()
    assert lens_dropped == 6, lens_dropped
    # A sky panorama (the console front-end renders with an HDR panorama and a lens: reference src/main.cc:24,421-425,450): the dropped cells' samples are not a
    # constant any more -- every one is its own sky texel (plus the sun) -- and k_resolve generates each sample's camera ray to look it up.  Same bits as with every
    # cell traced, same totals; and the oracle on a window of pure sky and on one across the silhouette.
    sky = helpers.scenes.sky_panorama()
    sky_dropped = 0
    for aperture, sun in ((0.0, (0.0, 0.0, 0.0)), (0.0, (9.0, 8.0, 7.0)), (0.01, (9.0, 8.0, 7.0)), (0.3, (0.0, 0.0, 0.0))):
        ses = binding.SceneSession(gpu_lib, obj, (0.5, 1.2, 9.0), (0.0, 1.0, -1.0), 40.0, 200 / 120, sun=sun, sun_dir=(-1.0, -1.0, 0.0), aperture=aperture, focal=7.0, sky_image=sky)
        for spp in (1, 5):
            img = ses.render(200, 120, spp)
            st1 = ses.stats().as_dict()
            monkeypatch.setenv("RAYLIB_CULL_CELLS", "0")
            ref = ses.render(200, 120, spp)
            st0 = ses.stats().as_dict()
            monkeypatch.delenv("RAYLIB_CULL_CELLS")
            assert np.array_equal(bits(img), bits(ref)), (aperture, sun, spp)
            for k in ("frameRays", "frameSamples", "frameNodes", "shadedHits", "trisTested", "pixels"):
                assert st0[k] == st1[k], (k, st0[k], st1[k], aperture, sun)
            sky_dropped += int(st1["culledCells"] > 0)
        if aperture == 0.0 and any(sun):
            flat.sun_illuminance = sun; flat.sun_direction = (-1.0, -1.0, 0.0)
            flat.textures.append(np.ascontiguousarray(sky, np.float32)); flat.sky_texture = len(flat.textures) - 1
            scene = oracle.scene_create(flat, 1)
            cam = ffi.make_camera((0.5, 1.2, 9.0), (0.0, 1.0, -1.0), 40.0, 200 / 120, 0.0, 7.0)   # (the focal distance scales the image plane: same rays only up to rounding)
            for (x0, y0, size) in ((0, 0, 24), (60, 30, 40)):
                same, tied, untied, err = window_mismatches_without_a_tie(oracle, scene, cam, ffi.make_settings(200, 120, 5), img, x0, y0, size)
                assert untied == 0 and same + tied == size * size and tied <= 8, (x0, y0, same, tied, untied, err)
            oracle.scene_destroy(scene)
            flat.textures.pop(); flat.sky_texture = -1
        ses.close()
    assert sky_dropped == 8, sky_dropped
    # A negative rayTMin lets a query find hits BEHIND its origin: with the light travelling from the box towards a camera that looks at the box from
    # downstream, the sun's occlusion query (origin: the camera, direction: away from the box) then meets the box at t < 0 and every traced sample loses
    # the sun -- a dropped cell filled with the sun's illuminance would be wrong.  Such a frame is not culled at all (csrc/rl_cull.cc).
    ses = binding.SceneSession(gpu_lib, obj, (0.0, 1.0, 6.0), (0.0, 1.0, -1.0), 45.0, 200 / 120, sun=(4.0, 5.0, 6.0), sun_dir=(0.0, 0.0, -1.0))
    for tmin, want_culled in ((1e-4, True), (-25.0, False)):
        img = ses.render(200, 120, 2, tmin=tmin)
        st1 = ses.stats().as_dict()
        monkeypatch.setenv("RAYLIB_CULL_CELLS", "0")
        ref = ses.render(200, 120, 2, tmin=tmin)
        st0 = ses.stats().as_dict()
        monkeypatch.delenv("RAYLIB_CULL_CELLS")
        assert np.array_equal(bits(img), bits(ref)), tmin
        assert (st1["culledCells"] > 0) == want_culled and st0["frameRays"] == st1["frameRays"], (tmin, st1["culledCells"])
        if not want_culled:
            assert (img[0, 0, :3] == 0.0).all() and (ref[0, 0, :3] == 0.0).all()      # the corner pixel: sky-less miss, sun hidden behind the camera's back
    ses.close()
    print("frames with dropped cells: %d of 36" % dropped_somewhere)
    assert dropped_somewhere >= 3, dropped_somewhere      # (a deterministic count now: the far, off-axis and side views drop cells)
    # the oracle on the far view: a window across the box's left edge
    ses = binding.SceneSession(gpu_lib, obj, (0.0, 1.0, 14.0), (0.0, 1.0, -1.0), 45.0, 200 / 120, sun=(9.0, 8.0, 7.0), sun_dir=(-1.0, -1.0, 0.0))
    img = ses.render(200, 120, 2)
    flat.sun_illuminance = (9.0, 8.0, 7.0); flat.sun_direction = (-1.0, -1.0, 0.0)
    scene = oracle.scene_create(flat, 1)
    cam = ffi.make_camera((0.0, 1.0, 14.0), (0.0, 1.0, -1.0), 45.0, 200 / 120)
    same, tied, untied, err = window_mismatches_without_a_tie(oracle, scene, cam, ffi.make_settings(200, 120, 2), img, 72, 32, 56)   # the whole silhouette and its surroundings
    print("far view with a sun: %d of %d window pixels bit-equal to the oracle, %d tie pixels" % (same, 56 * 56, tied))
    assert untied == 0 and same + tied == 56 * 56 and tied <= 8, (same, tied, untied, err)
    oracle.scene_destroy(scene)
    ses.close()


@pytest.mark.timeout(120)
def test_leaf_list_with_every_leaf_a_candidate(gpu_lib, workdir, monkeypatch):
    """Soups of large overlapping triangles, 24 leaves: with a negative rayTMin (no cut) a ray visits every leaf whose box it meets, and many meet
    all 24 -- the case in which the pick loop of the first version never ended (its end test assumed an unused slot).  Must return, and with the
    bits of the BVH4 walk."""
    from raylib_amd import binding
    monkeypatch.setenv("RAYLIB_POOL", "0")
    d = os.path.join(str(workdir), "leaflist_all"); os.makedirs(d, exist_ok=True)
    for k, (n, seed) in enumerate(((97, 5), (103, 11), (108, 23))):
        obj, nt = helpers.scenes.soup(os.path.join(d, "s%d.obj" % k), n_tris=n, seed=seed, extent=0.6, size=1.8)
        ses = binding.SceneSession(gpu_lib, obj, (0, 1, 4), (0, 1, -1), 50.0, 1.5)
        most = C.c_uint32()
        assert gpu_lib.RaylibAMD_SceneLeafListInfo(ses.scene, C.byref(most)) == 24
        for tmin in (-1e-4, -0.05, 0.0, 1e-4):
            img = ses.render(96, 64, 2, max_path=9, tmin=tmin)
            monkeypatch.setenv("RAYLIB_LEAF_LIST", "0")
            ref = ses.render(96, 64, 2, max_path=9, tmin=tmin)
            monkeypatch.delenv("RAYLIB_LEAF_LIST")
            assert helpers.same(img, ref).all(), (n, tmin)
        ses.close()


def test_pool_schedule_is_default_on_deep_bvh_and_bit_identical(mid_scene, gpu_lib, oracle, monkeypatch):
    ses, obj, n = mid_scene
    img = ses.render(96, 64, 8, max_path=6)
    st = ses.stats().as_dict()
    assert st["bvhDepth"] > 16 and st["numTriangles"] == n
    monkeypatch.setenv("RAYLIB_POOL", "0")
    base = ses.render(96, 64, 8, max_path=6)
    st0 = ses.stats().as_dict()
    assert np.array_equal(bits(img), bits(base))
    assert st["rays"] == st0["rays"] and st["shadedHits"] == st0["shadedHits"]
    assert st["pathsPerWave"] == 128 and st0["pathsPerWave"] == 64      # it really was another schedule (what ran, not how long it took)
    for k in ("3", "4"):
        monkeypatch.setenv("RAYLIB_POOL", k)
        assert np.array_equal(bits(ses.render(96, 64, 8, max_path=6)), bits(base))
    monkeypatch.delenv("RAYLIB_POOL")
    # the traversal stack's LDS part: 19 entries (default up to depth 24), all 32, and 4 (test build: the private overflow
    # array takes nearly every push)
    for mode in ("0", "1", "4"):
        monkeypatch.setenv("RAYLIB_POOL_SHORT_STACK", mode)
        assert np.array_equal(bits(ses.render(96, 64, 8, max_path=6)), bits(base)), mode
    monkeypatch.delenv("RAYLIB_POOL_SHORT_STACK")
    # sample batches (the sample buffer is summed in sample order whatever the batch size) and the BVH2 under the pool schedule
    # ... and the 8-wide tree, which this scene is too shallow to get by default
    assert st["treeWidth"] == 4 and st["nodeBytes"] == 64
    for env in (dict(RAYLIB_SAMPLE_BATCH="3"), dict(RAYLIB_BVH4="0"), dict(RAYLIB_BVH4="0", RAYLIB_SAMPLE_BATCH="1"), dict(RAYLIB_BVH8="1"), dict(RAYLIB_BVH8="1", RAYLIB_JOB_HEADS="1")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert np.array_equal(bits(ses.render(96, 64, 8, max_path=6)), bits(base)), env
        se = ses.stats().as_dict()
        assert se["treeWidth"] == (8 if "RAYLIB_BVH8" in env else 2 if "RAYLIB_BVH4" in env else 4), env
        assert se["rays"] == st0["rays"] and se["shadedHits"] == st0["shadedHits"], env
        for k in env:
            monkeypatch.delenv(k)
    # windows recomputed by the CPU oracle with the same pixel keys
    flat = helpers.objflat.load_obj(obj, oracle, texture_loader=helpers.texture_loader, sun_illuminance=(20, 20, 20), sun_direction=(-1.0, -1.0, 0.0))
    scene = oracle.scene_create(flat, 1)
    cam = ffi.make_camera((0, 1, 5), (0, 1, -1), 60.0, 96 / 64)
    stg = ffi.make_settings(96, 64, 8, max_path=6)
    for (x0, y0) in ((40, 24), (0, 0), (80, 48)):
        same, tied, untied, err = window_mismatches_without_a_tie(oracle, scene, cam, stg, img, x0, y0, 16)
        assert err < L2_TOL and untied == 0 and same + tied == 256 and tied <= 4, (x0, y0, same, tied, untied, err)


def test_bench_multi_rank_frame_assembly_on_one_gpu():
    """bench.py's N > 1 path on a 1-GPU box: BENCH_SHARE_GPU=1 puts 2 ranks on the one device and stages the gather through the
    host over gloo (RCCL refuses two ranks per device).  What is checked is the logic the real run uses -- cells dealt round-robin,
    double-buffered send, scatter plan -- by bench.py's own frame check: the assembled frame equals a one-GPU render bit for bit."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    for k in ("RAYLIB_POOL", "RAYLIB_LIB"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["frame_check"].startswith("assembled frame bit-identical to a one-GPU render") and "MISMATCH" not in d["config"]["frame_check"]
    assert "reference-rendered windows bit-identical" in d["config"]["frame_check"]
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert d["multi_gpu"]["ranks"] == 2 and len(d["multi_gpu"]["rank_trace_ms"]) == 2 and min(d["multi_gpu"]["rank_trace_ms"]) > 0
    check_multi_rank_roofline(d["roofline"], leaf_list=True)
    # the pool schedule (the configs[2]-sized scene on the 8-wide tree) through the same two-rank path
    cmd[cmd.index("29533")] = "29535"
    out = subprocess.run(cmd + ["--workload", "breakfast_300k_1080p_128spp"], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["roofline"]["kernel"] == "k_trace_pool" and d["roofline"]["algorithmic"]["tree_width"] == 8
    assert d["config"]["frame_check"].startswith("assembled frame bit-identical to a one-GPU render") and "MISMATCH" not in d["config"]["frame_check"]
    check_multi_rank_roofline(d["roofline"], leaf_list=False)


def check_multi_rank_roofline(r, leaf_list):
    """What an N > 1 bench line may say about the roofline: a fraction of a peak is at most 1 or absent; bytes the leaf-list kernel was served
    from LDS are never put over the HBM peak -- the algorithmic bytes are the tree walk's, from an untimed frame --; replayed N = 1 counters say that they are scaled."""
    assert r["frac"] is None or 0.0 < r["frac"] <= 1.0, r["frac"]
    a = r["algorithmic"]
    assert a["frac_of_hbm_peak"] is None or a["frac_of_hbm_peak"] > 0.0
    if leaf_list:
        assert a["tree_width"] == 0 and a["served_elsewhere"] is not None
        assert a["bytes_per_launch"] is None or a["bytes_per_launch"] < a["served_elsewhere"]["lds_served_bytes_per_launch"]
        assert not (r["resource"].startswith("HBM (algorithmic") and a["bytes_per_launch"] is None)
    if r["valu"] is not None:
        assert r["valu"]["scaled_from_n1"] is True and "SCALED" in r["traffic_source"] and r["valu"]["frac_of_spec_peak"] <= 1.0
    if r["frac"] is not None and r["achieved"] is not None:
        assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-9


def test_bench_default_run_times_the_boundary_and_checks_the_frame():
    """`python bench.py` as the driver runs it at N = 1: the timed call is Raylib_Render, and the frame it produced is compared, outside
    the timed region, with windows the reference build rendered (tests/golden/bench_windows.npz).  The same invocation reports the
    configs[2]-sized scene (a memory-bound megakernel) under "extra"."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RAYLIB_POOL", "RAYLIB_LIB", "BENCH_SHARE_GPU", "WORLD_SIZE", "RANK", "LOCAL_RANK", "RAYLIB_NUM_GPUS", "RAYLIB_GPU_MAP", "RAYLIB_JOB_HEADS"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["boundary"]["timed_entry"] == "Raylib_Render"
    assert d["config"]["frame_check"].endswith("reference-rendered windows bit-identical") and "MISMATCH" not in d["config"]["frame_check"]
    b = d["config"]["boundary"]
    # (times are printed by bench.py and judged by whoever reads the line: a test only checks that they are there)
    assert b["render_plus_dump_ms_per_step"] > 0 and b["raylib_render_ms_per_step"] > 0 and b["render_device_ms_per_step"] > 0
    sp = d["ms_per_step_spread"]
    assert sp["n"] == 10 and sp["min"] <= sp["median"] <= sp["max"]
    r = d["roofline"]
    assert set(r) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic", "hbm", "valu", "replayed_pmc"}
    # peaks are constants of the part, whatever ran
    assert r["peak"] in (8000.0, 1024 * 2.4)
    if r["frac"] is None:
        # no counter passes of THIS build are committed (a source change since the last profile run) and the tree walk's algorithmic bytes were served by the
        # caches faster than HBM could have: the line then names no bounding resource rather than a fraction above 1
        assert (r["replayed_pmc"] is None or r["replayed_pmc"]["stale"] is True) and r["resource"].startswith("none:") and (r["achieved"] is None or r["achieved"] > r["peak"])
    else:
        assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-9 and r["frac"] <= 1.0
    a = r["algorithmic"]
    # the Cornell frame's algorithmic bytes are the TREE WALK's (a workload property), the leaf list's LDS traffic sits next to them
    assert a["bytes_per_camera_sample"] > 0 and a["served_elsewhere"]["lds_served_bytes_per_camera_sample"] > a["bytes_per_camera_sample"]
    # `value` is made of EXECUTED queries only; what the silhouette cull left out is listed next to it (20 190 of the Cornell frame's 32 400 cells)
    wk = d["config"]["work"]
    assert wk["cells_culled"] == 20190 and wk["cells_culled"] + wk["cells_listed"] == 32400
    assert wk["rays_accounted_not_traced_per_step"] == 20190 * 64 * 64 and wk["camera_samples_executed_per_step"] == (32400 - 20190) * 64 * 64
    assert abs(d["value"] - wk["rays_executed_per_step"] / d["ms_per_step"] / 1e3) < 1e-6 * d["value"]
    assert r["job_heads"] == 8
    if r["replayed_pmc"] is not None:
        rp = r["replayed_pmc"]
        assert rp["loaded_build_id"] == d["config"]["build_id"] and rp["stale"] == (rp["build_id"] != rp["loaded_build_id"])
        assert ("STALE" in r["traffic_source"]) == rp["stale"]
        if r["valu"] is not None:
            assert r["valu"]["frac_of_spec_peak"] > 0.0 and 2.0 <= r["valu"]["mean_cost_cycles_per_inst"] < 8.0   # (the cost is a property of the replayed counters, not of this run's time)
    e = d["extra"]
    assert e["workload"] == "breakfast_300k_1080p_128spp" and e["value"] > 0 and e["scene_triangles"] > 290000
    assert e["frame_check"].endswith("reference-rendered windows bit-identical")
    assert e["roofline"]["kernel"] == "k_trace_pool" and e["roofline"]["algorithmic"]["frac_of_hbm_peak"] > 0.0
    assert e["work"]["cells_culled"] == 28800 and e["work"]["rays_accounted_not_traced_per_step"] == 28800 * 64 * 128 * 2
    # ... and the same scene from inside: nothing can be dropped, every counted ray ran
    ei = d["extra_interior"]
    assert ei["workload"] == "breakfast_interior_300k_1080p_128spp" and ei["value"] > 0 and ei["scene_triangles"] == e["scene_triangles"]
    assert ei["work"]["cells_culled"] == 0 and ei["work"]["rays_accounted_not_traced_per_step"] == 0 and ei["work"]["camera_samples_executed_per_step"] == 1920 * 1080 * 128
    assert ei["frame_check"].endswith("reference-rendered windows bit-identical") and ei["roofline"]["kernel"] == "k_trace_pool"
    # ... and with albedo maps on the walls and a fifth of the triangles as alpha-cut-out foliage cards (round 5): texels are fetched inside traversal
    et = d["extra_textured"]
    assert et["workload"] == "breakfast_textured_interior_300k_1080p_128spp" and et["value"] > 0 and et["scene_triangles"] == e["scene_triangles"]
    assert et["frame_check"].endswith("reference-rendered windows bit-identical") and et["roofline"]["kernel"] == "k_trace_pool"
    assert et["work"]["cells_culled"] == 0 and et["work"]["texel_fetches_per_ray"] > 0.5


def test_bench_library_mode_runs_n_ranks_behind_raylib_render_or_refuses():
    """`python bench.py --gpus N` without torch.distributed.run is the library mode: RAYLIB_NUM_GPUS = N behind Raylib_Render.  On this
    1-GPU box N = 2 must be REFUSED (exit code != 0, no JSON line) unless RAYLIB_GPU_MAP puts both logical ranks on the one device;
    then the line says n_gpus 2, names the gather mechanism and carries per-rank times, and the frame check holds."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RAYLIB_POOL", "RAYLIB_LIB", "BENCH_SHARE_GPU", "WORLD_SIZE", "RANK", "LOCAL_RANK", "RAYLIB_NUM_GPUS", "RAYLIB_GPU_MAP", "RAYLIB_GATHER_SELF"):
        env.pop(k, None)
    import torch
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra"]
    if torch.cuda.device_count() < 2:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")], out.stdout[-500:]
        assert "refusing to measure fewer GPUs" in out.stderr
    out = subprocess.run(cmd, env=dict(env, RAYLIB_GPU_MAP="0,0", RAYLIB_GATHER_SELF="1"), capture_output=True, text=True, timeout=300)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["boundary"]["timed_entry"] == "Raylib_Render"
    assert d["config"]["frame_check"].endswith("reference-rendered windows bit-identical")
    m = d["multi_gpu"]
    assert m["ranks"] == 2 and m["devices"] == 1 and m["gather"] in ("rccl", "peer") and len(m["rank_kernel_ms"]) == 2 and min(m["rank_kernel_ms"]) > 0
    assert m["gather"] != "rccl" or m["rccl_comm_size"] == 1
    check_multi_rank_roofline(d["roofline"], leaf_list=True)
    assert d["roofline"]["algorithmic"]["gpus"] == 2 and "n1_reference" in d["config"]
    # ... and the pool schedule on the 8-wide tree behind the same call
    out = subprocess.run(cmd + ["--workload", "breakfast_300k_1080p_128spp"], env=dict(env, RAYLIB_GPU_MAP="0,0", RAYLIB_GATHER_SELF="1"), capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["roofline"]["kernel"] == "k_trace_pool" and d["roofline"]["algorithmic"]["tree_width"] == 8
    assert d["config"]["frame_check"].endswith("reference-rendered windows bit-identical")
    check_multi_rank_roofline(d["roofline"], leaf_list=False)
    # a mismatch between --gpus and the launcher is an error, not a silent one-GPU run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra"],
                         env=dict(env, RAYLIB_NUM_GPUS="2", RAYLIB_GPU_MAP="0,0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "RAYLIB_NUM_GPUS" in bad.stderr
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_bench_rccl_gather_path_with_one_rank():
    """bench.py launched the way the driver launches N > 1 (torch.distributed.run, backend nccl = RCCL), with the one rank a 1-GPU box
    allows: process group on the device, asynchronous gather on RCCL's stream, assembly on rank 0, frame check.  The ranks' device
    selection, double-buffered send and stream ordering are the ones the multi-GPU run uses; only the peer count differs."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RAYLIB_POOL", "RAYLIB_LIB", "BENCH_SHARE_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29534",
           os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"]["frame_check"].startswith("assembled frame bit-identical to a one-GPU render") and "MISMATCH" not in d["config"]["frame_check"]
    assert "reference-rendered windows bit-identical" in d["config"]["frame_check"]


def test_deferred_readback_reaches_every_host_reader(gpu_lib, workdir):
    """Raylib_Render and Raylib_PostProcess leave the frame on the device; the host pixels are fetched when something reads them.
    Every reader must see the rendered frame: the two dump calls, the file writer (without a dump before it), an image used as a
    sky panorama of another scene, and a second render into the same handle with a different viewport (reallocation)."""
    from raylib_amd import binding
    lib = gpu_lib
    obj, c = helpers.build_case("cornell", workdir)
    ses = binding.SceneSession(lib, obj, c["origin"], c["look_at"], 45.0, 32 / 24)
    want = ses.render(32, 24, 4)                                     # SceneSession.render dumps RGBA
    st = ses.settings(32, 24, 4)
    # (1) writer straight after the render, no dump in between
    img = lib.Raylib_CreateImage(32, 24)
    lib.Raylib_Render(C.byref(st), ses.scene, ses.camera, img)
    bmp = os.path.join(str(workdir), "deferred.bmp")
    assert lib.Raylib_WriteImageToDisk(img, bmp.encode(), 0) == 1
    back = lib.Raylib_LoadImage(bmp.encode()); assert back
    got8 = np.zeros((24, 32, 4), np.float32)
    lib.RaylibAMD_DumpImageRGBA(back, got8.ctypes.data_as(C.POINTER(C.c_float)))
    assert got8[..., :3].max() > 0.0                                 # not the cleared host buffer
    ref8 = (((want[..., :3] * np.float32(255.0)).astype(np.uint32) & 0xff).astype(np.float32) / np.float32(255.0))   # image.h:62-69: no clamp
    assert np.array_equal(got8[..., :3], ref8)
    lib.Raylib_DestroyImage(back)
    # (2) both dumps after the writer
    rgba = np.zeros((24, 32, 4), np.float32)
    lib.RaylibAMD_DumpImageRGBA(img, rgba.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(bits(rgba), bits(want))
    rgb = np.zeros(24 * 32 * 3, np.float32)
    lib.Raylib_DumpImageData(img, rgb.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(bits(rgb.reshape(24, 32, 3)), bits(want[..., :3]))
    # (3) a rendered image as the sky of an empty scene: every camera ray returns a sky texel, none of them the cleared value
    lib.Raylib_Render(C.byref(st), ses.scene, ses.camera, img)       # stale again
    sc = lib.Raylib_CreateScene(); lib.Raylib_SetSkyPanorama(sc, img); lib.Raylib_FinalizeScene(sc)
    sky = lib.Raylib_CreateImage(16, 16)
    st2 = binding.RendererSettings(16, 16, 1, 5, 0.0001, 0)
    lib.Raylib_Render(C.byref(st2), sc, ses.camera, sky)
    out = np.zeros((16, 16, 4), np.float32)
    lib.RaylibAMD_DumpImageRGBA(sky, out.ctypes.data_as(C.POINTER(C.c_float)))
    texels = {tuple(p) for p in bits(want[..., :3]).reshape(-1, 3)}
    assert all(tuple(p) in texels for p in bits(out[..., :3]).reshape(-1, 3))
    assert out[..., :3].max() > 0.0
    lib.Raylib_DestroyScene(sc); lib.Raylib_DestroyImage(sky)
    # (4) the same handle rendered at another size
    st3 = ses.settings(24, 16, 1)
    lib.Raylib_Render(C.byref(st3), ses.scene, ses.camera, img)
    small = np.zeros(24 * 16 * 3, np.float32)
    lib.Raylib_DumpImageData(img, small.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(small.reshape(16, 24, 3), ses.render(24, 16, 1)[..., :3])
    lib.Raylib_DestroyImage(img)
    ses.close()


def test_sky_panorama_swapped_by_destroy_and_create(gpu_lib):
    """A front-end that swaps panoramas destroys image A and creates image B; the allocator very likely hands B the address A had.  The
    device copy of the sky is keyed on (handle, pixel version): versions are unique in the process, so B can never pass for A."""
    from raylib_amd import binding
    lib = gpu_lib
    cam = lib.Raylib_CreateCamera()
    lib.Raylib_CameraSetPosition(cam, 0.0, 0.0, 0.0); lib.Raylib_CameraSetLookAt(cam, 0.0, 0.0, -1.0); lib.Raylib_CameraSetPerspective(cam, 60.0, 1.0)
    sc = lib.Raylib_CreateScene(); lib.Raylib_FinalizeScene(sc)
    st = binding.RendererSettings(16, 16, 1, 5, 0.0001, 0)
    out = lib.Raylib_CreateImage(16, 16)
    seen = []
    for colour in ((0.75, 0.0, 0.0, 1.0), (0.0, 0.5, 0.0, 1.0), (0.0, 0.0, 0.25, 1.0)):
        px = np.tile(np.array(colour, np.float32), (8 * 16, 1))
        sky = lib.RaylibAMD_CreateImageFromData(16, 8, px.ctypes.data_as(C.POINTER(C.c_float)))
        seen.append(sky)
        lib.Raylib_SetSkyPanorama(sc, sky)
        lib.Raylib_Render(C.byref(st), sc, cam, out)
        got = np.zeros((16, 16, 4), np.float32)
        lib.RaylibAMD_DumpImageRGBA(out, got.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(got[..., :3], np.broadcast_to(np.array(colour[:3], np.float32), (16, 16, 3))), colour
        lib.Raylib_DestroyImage(sky)     # no render between this and the next panorama's SetSkyPanorama
    print("panorama handles: %s (%d distinct)" % (seen, len(set(seen))))
    lib.Raylib_DestroyImage(out); lib.Raylib_DestroyScene(sc); lib.Raylib_DestroyCamera(cam)


@pytest.mark.parametrize("name", ["cornell", "cornell_glass_sun", "pbr_maps"])
def test_reflectance_and_microsurface_aovs_match_the_oracle_definition(name, sessions, gpu_lib, oracle, workdir):
    """Modes 3 and 6 read an uninitialised tangent frame in the reference (renderer.cc:89-93,104-108), so no reference fixture exists.  The
    defined variant -- the frame built first, as TraceScene does -- is what the oracle restates; the device must give the same bits, including
    the random draws Scatter makes after the camera ray's (open aperture and shutter in the second case)."""
    ses = sessions[name]
    obj, c, flat = helpers.flat_for_case(name, workdir, oracle)
    cam = helpers.camera_for_case(c)
    scene = oracle.scene_create(flat, 1)
    ties = tie_mask(oracle, flat, ffi.make_camera(c["origin"], c["look_at"], c["fov"], c["aspect"]), 64, 64)
    for mode in (6, 3):
        want = oracle.render(scene, cam, ffi.make_settings(64, 64, 1, mode=mode), seed=1)
        assert_same_outside_ties(ses.render(64, 64, 1, mode=mode), want, ties, "mode %d of %s" % (mode, name))
