"""Host-side logic of libraylib.so that needs no GPU: the C-ABI surface, OBJ/MTL
ingestion, scene flattening, the BVH builder, image codecs, registries."""
import ctypes as C
import os
import re
import numpy as np
import pytest

import helpers
from helpers import ffi, scenes

INCLUDE = os.path.join(helpers.ROOT, "include")


def declared_symbols(header):
    txt = open(os.path.join(INCLUDE, header)).read()
    return re.findall(r"RAYLIB_API\s+[\w\s\*]+?\b(Raylib(?:AMD)?_\w+)\s*\(", txt)


def test_library_exports_every_declared_symbol(lib):
    from raylib_amd import binding
    raylib = declared_symbols("raylib.h")
    amd = declared_symbols("raylib_amd.h")
    assert len(raylib) == 33, raylib                       # the reference's 33 entry points (raylib/raylib.h:17-151)
    assert sorted(raylib) == sorted(binding.RAYLIB_H_EXPORTS)
    assert sorted(amd) == sorted(binding.RAYLIB_AMD_H_EXPORTS)
    for name in raylib + amd:
        assert getattr(lib, name) is not None


def test_settings_struct_layout():
    from raylib_amd import binding
    assert C.sizeof(binding.RendererSettings) == 24        # reference raylib_types.h:41-57 / RaylibWrapper.cs:27-38
    assert binding.RendererSettings.renderMode.offset == 20


def test_render_mode_strings_and_misc(lib):
    names = [lib.Raylib_GetRenderModeString(i) for i in range(7)]
    assert names == [b"Default", b"Albedo", b"SurfaceNormal", b"MicrosurfaceNormal", b"Texcoord", b"Emission", b"Reflectance"]
    assert lib.Raylib_GetRenderModeString(7) is None
    assert lib.Raylib_IsDenoiserSupported() == 0
    assert lib.Raylib_Denoise(None, 1, None, None, None) == 0
    assert lib.Raylib_LoadOBJModel(b"/nonexistent/file.obj") is None
    assert lib.Raylib_LoadImage(b"/nonexistent/file.png") is None


def test_handle_registries(lib):
    cam = lib.Raylib_CreateCamera()
    assert lib.Raylib_DestroyCamera(cam) == 1 and lib.Raylib_DestroyCamera(cam) == 0
    img = lib.Raylib_CreateImage(5, 3)
    out = np.ones(5 * 3 * 3, np.float32)
    lib.Raylib_DumpImageData(img, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert (out == 0).all()                                 # cleared to RGBA 0 (raylib.cc:181-186)
    assert lib.Raylib_DestroyImage(img) == 1 and lib.Raylib_DestroyImage(img) == 0
    sc = lib.Raylib_CreateScene()
    lib.Raylib_FinalizeScene(sc)
    lib.Raylib_FinalizeScene(sc)                            # idempotent
    assert lib.Raylib_DestroyScene(sc) == 1 and lib.Raylib_DestroyScene(sc) == 0


@pytest.mark.parametrize("name", list(helpers.CASES))
def test_obj_ingestion_matches_oracle_side_parser(name, lib, oracle, workdir):
    """Product OBJ/MTL loader + MTL->material rules vs the independent Python parser + oracle rule."""
    obj, c, flat = helpers.flat_for_case(name, workdir, oracle)
    ses = helpers.session_for_case(lib, name, workdir)
    tris, mats = ses.export_flat()
    assert tris.tobytes() == flat.triangles.tobytes()
    assert mats.tobytes() == flat.materials.tobytes()
    # the checker's flat scene lists the sky panorama as its last texture; the product reads the panorama through the image
    # handle when a render starts (as the reference does), so it is not one of the scene's textures
    material_textures = flat.textures[:-1] if c["sky"] else flat.textures
    assert lib.RaylibAMD_SceneNumTextures(ses.scene) == len(material_textures)
    for i, t in enumerate(material_textures):
        w, h = C.c_int32(), C.c_int32()
        lib.RaylibAMD_SceneTextureSize(ses.scene, i, C.byref(w), C.byref(h))
        assert (h.value, w.value) == t.shape[:2]
        got = np.zeros_like(t)
        lib.RaylibAMD_SceneExportTexture(ses.scene, i, got.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(got, t)                       # PNG decode == byte/255 of the generator's pixels
    ill, d = (C.c_float * 3)(), (C.c_float * 3)()
    lib.RaylibAMD_SceneGetSun(ses.scene, ill, d)
    sd = np.asarray(c["sun_dir"], np.float32)
    k = np.float32(1.0) / np.sqrt(np.float32(sd[0] * sd[0] + sd[1] * sd[1]) + np.float32(sd[2] * sd[2]))
    assert np.allclose(np.asarray(d[:]), sd * k, rtol=0, atol=1e-7)
    ses.close()


def test_textured_room_generator_loader_and_checker_agree(lib, oracle, workdir):
    """The BASELINE-size textured / alpha-cut-out room (scenes.textured: albedo maps on every wall, a fifth of the triangles as foliage cards with a cut-out map;
    SURVEY 8d) at a small tessellation: the product's loader, the checker-side parser and the generator's own arrays (helpers.big_scene, what the full-size GPU tests
    hand the oracle) give the same flat scene -- triangles, materials with their texture numbers, decoded texels."""
    from raylib_amd import binding
    d = os.path.join(str(workdir), "textured_small"); os.makedirs(d, exist_ok=True)
    obj, n = scenes.textured(os.path.join(d, "t.obj"), tess=9, displace_fraction=0.2)
    flat = helpers.objflat.load_obj(obj, oracle, texture_loader=helpers.texture_loader)
    ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1.0)
    tris, mats = ses.export_flat()
    assert len(tris) == n == 2916 and tris.tobytes() == flat.triangles.tobytes() and mats.tobytes() == flat.materials.tobytes()
    cards = flat.triangles["material"] == [m["name"] for m in helpers.objflat.parse_mtl(obj[:-4] + ".mtl")].index("foliage")
    assert 0.1 * n < cards.sum() < 0.3 * n                                  # at least a tenth of the triangles are cut-out cards
    assert (flat.materials["texAlbedo"] >= 0).sum() == 4 and lib.RaylibAMD_SceneNumTextures(ses.scene) == 4
    for i, t in enumerate(flat.textures):
        got = np.zeros_like(t)
        lib.RaylibAMD_SceneExportTexture(ses.scene, i, got.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(got, t)
    leaf = scenes.foliage_texture()
    assert set(np.unique(leaf[..., 3])) == {0, 255} and 0.3 < (leaf[..., 3] > 0).mean() < 0.7
    ses.close()
    d2 = os.path.join(str(workdir), "textured_small_arrays"); os.makedirs(d2, exist_ok=True)
    obj2, flat2 = helpers.big_scene(os.path.join(d2, "t.obj"), None, scenes.TEXTURED_MTL, oracle, 9, arrays=scenes.build_arrays_textured(9, 0.2), textures=scenes.textured_textures())
    assert open(obj2).read() == open(obj).read()                          # the checker library's fast writer prints what the Python writer prints
    assert flat2.triangles.tobytes() == flat.triangles.tobytes() and flat2.materials.tobytes() == flat.materials.tobytes()
    assert len(flat2.textures) == len(flat.textures) and all(np.array_equal(a, b) for a, b in zip(flat2.textures, flat.textures))


def test_golden_scene_matches_product_loader(lib, workdir):
    """The flat scene stored in the fixtures (what the reference rendered) is what the product loads."""
    for name in helpers.CASES:
        g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
        ses = helpers.session_for_case(lib, name, workdir)
        tris, mats = ses.export_flat()
        assert tris.tobytes() == g["triangles"].tobytes() and mats.tobytes() == g["materials"].tobytes()
        ses.close()


def test_camera_derived_state_matches_oracle(lib, oracle):
    cam_h = lib.Raylib_CreateCamera()
    lib.Raylib_CameraSetPosition(cam_h, 0.3, 1.2, 4.0)
    lib.Raylib_CameraSetLookAt(cam_h, 0.0, 0.9, -1.0)
    lib.Raylib_CameraSetPerspective(cam_h, 50.0, 1.5)
    lib.Raylib_CameraSetLens(cam_h, 0.0, 3.0)
    lib.Raylib_CameraSetMotion(cam_h, 0.0, 0.0)
    out = (C.c_float * 19)()
    lib.RaylibAMD_CameraExport(cam_h, out)
    o = np.asarray(out[:], np.float32)
    origin, top_left, horiz, vert = o[0:3], o[4:7], o[7:10], o[10:13]
    cam = ffi.make_camera((0.3, 1.2, 4.0), (0.0, 0.9, -1.0), 50.0, 1.5, 0.0, 3.0)
    uv = np.array([[0, 0], [1, 0], [0, 1], [0.25, 0.75]], np.float32)
    rays = oracle.camera_rays(cam, uv, seed=1)
    f = np.float32
    for (s, t), r in zip(uv, rays):
        d = ((top_left + f(s) * horiz).astype(np.float32) + (f(1.0) - f(t)) * vert).astype(np.float32) - origin
        d = (d - np.zeros(3, np.float32)).astype(np.float32)
        d = d * (f(1.0) / np.sqrt(f(f(d[0] * d[0]) + f(d[1] * d[1])) + f(d[2] * d[2])))
        assert np.array_equal(np.asarray(d, np.float32), r[3:6])
    # copy keeps everything
    c2 = lib.Raylib_CreateCamera()
    lib.Raylib_CameraCopy(cam_h, c2)
    out2 = (C.c_float * 19)()
    lib.RaylibAMD_CameraExport(c2, out2)
    assert out[:] == out2[:]
    lib.Raylib_DestroyCamera(cam_h); lib.Raylib_DestroyCamera(c2)


def test_bvh_builder(lib, workdir):
    from raylib_amd import binding
    for make, kw, expect in ((scenes.cornell, dict(tess=6, displace_fraction=0.2), 36 * 36), (scenes.soup, dict(n_tris=5000), 5000)):
        obj, n = make(os.path.join(str(workdir), "bvh_%s.obj" % make.__name__), **kw)
        assert n == expect
        ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1.0)
        nodes, depth, sah = C.c_uint32(), C.c_uint32(), C.c_float()
        assert lib.RaylibAMD_SceneBVHInfo(ses.scene, C.byref(nodes), C.byref(depth), C.byref(sah)) == 1
        assert lib.RaylibAMD_SceneNumTriangles(ses.scene) == n
        assert n / 4 <= nodes.value <= n and depth.value <= 64
        assert lib.RaylibAMD_SceneBVH4Info(ses.scene, None, None) == 1    # triangle scenes of 8+ triangles carry the valid wide tree
        ses.close()
    # empty and single-triangle scenes
    sc = lib.Raylib_CreateScene(); lib.Raylib_FinalizeScene(sc)
    assert lib.RaylibAMD_SceneNumTriangles(sc) == 0
    lib.Raylib_DestroyScene(sc)


def test_small_scenes_carry_a_valid_leaf_list(lib, workdir):
    """Scenes of at most 108 triangles get the leaf list k_trace walks instead of the tree (rl_bvh.cc): at most 24 leaves of at most 8
    triangles, every triangle in exactly one of them and inside its box (RaylibAMD_SceneBVH4Info checks that), none for larger scenes."""
    from raylib_amd import binding
    cases = [(scenes.cornell, {}, True), (scenes.cutout, {}, True), (scenes.pbr_maps, {}, True),
             (scenes.cornell, dict(tess=2), False), (scenes.soup, dict(n_tris=100), True), (scenes.soup, dict(n_tris=109), False)]
    for k, (make, kw, expect) in enumerate(cases):
        obj, n = make(os.path.join(str(workdir), "ll_%d.obj" % k), **kw)[:2]
        ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1.0)
        most = C.c_uint32()
        leaves = lib.RaylibAMD_SceneLeafListInfo(ses.scene, C.byref(most))
        if n >= 8:
            assert lib.RaylibAMD_SceneBVH4Info(ses.scene, None, None) == 1, (make.__name__, kw)
        if expect:
            assert 1 <= leaves <= 24 and 1 <= most.value <= 8 and n <= 108, (make.__name__, kw, n, leaves, most.value)
            assert lib.RaylibAMD_SceneBVH8Info(ses.scene, None, None, None, None) == 0    # (the 8-wide tree orders triangle slots its own way: not for leaf-list scenes)
        else:
            assert leaves == 0 and n > 108, (make.__name__, kw, n, leaves)
        ses.close()


def test_larger_scenes_carry_a_valid_eight_wide_tree(lib, workdir, monkeypatch):
    """Above 108 triangles the builder also emits the 8-wide collapse (rl_bvh.cc CollapseWide8 / EmitWide8): structurally valid on rooms, soups (long thin
    triangles, coincident vertices) and cut-out scenes, fewer nodes and fewer expected steps than the 4-wide tree, levels within what the kernel's group
    stack holds (16)."""
    from raylib_amd import binding
    cases = [(scenes.soup, dict(n_tris=109)), (scenes.soup, dict(n_tris=5000, seed=3)), (scenes.cornell, dict(tess=9, displace_fraction=0.2)),
             (scenes.cornell, dict(tess=40)), (scenes.cutout, dict(tess=12)), (scenes.colonnade, dict(tess=2))]
    for k, (make, kw) in enumerate(cases):
        obj, n = make(os.path.join(str(workdir), "w8_%d.obj" % k), **kw)[:2]
        ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1.0)
        n4, need, n8, lev, s4, s8 = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_float(), C.c_float()
        assert lib.RaylibAMD_SceneBVH4Info(ses.scene, C.byref(n4), C.byref(need)) == 1, (make.__name__, kw)
        assert lib.RaylibAMD_SceneBVH8Info(ses.scene, C.byref(n8), C.byref(lev), C.byref(s4), C.byref(s8)) == 1, (make.__name__, kw)
        assert n / 8 / 4 <= n8.value < n4.value and 1 <= lev.value <= 16, (make.__name__, kw, n, n4.value, n8.value, lev.value)
        assert 1.0 <= s8.value < s4.value, (make.__name__, kw, s4.value, s8.value)
        ses.close()
        # the planned collapse (least summed node area) is never worse than opening the largest child first, the A/B switch's tree is valid too
        monkeypatch.setenv("RAYLIB_WIDE_GREEDY", "1")
        ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1.0)
        g8, gs8 = C.c_uint32(), C.c_float()
        assert lib.RaylibAMD_SceneBVH8Info(ses.scene, C.byref(g8), None, None, C.byref(gs8)) == 1, (make.__name__, kw)
        assert s8.value <= gs8.value * (1 + 1e-5), (make.__name__, kw, s8.value, gs8.value)
        ses.close()
        monkeypatch.delenv("RAYLIB_WIDE_GREEDY")


@pytest.mark.parametrize("make,kw", [(scenes.cornell, dict(tess=24, displace_fraction=0.2)), (scenes.cornell, dict(tess=64, displace_fraction=0.2)),
                                      (scenes.soup, dict(n_tris=6000, seed=9)), (scenes.colonnade, dict(tess=3))], ids=["room_21k", "room_147k_split_leaves", "soup_6k", "colonnade_10k"])
def test_eight_wide_walk_restated_on_the_host_never_skips_the_closest_hit(make, kw, lib, oracle, workdir, monkeypatch):
    """The megakernel's 8-wide walk restated on the host (RaylibAMD_SceneWalk8Host: the same float operations as rl_render.hip NodeStep8 -- half-float planes
    through one fma each, the ray's widened factors, visiting order, groups) against the oracle's closest hit: with the exit distance fixed just behind the
    oracle's hit the walk must reach the leaf that holds it, for rays in random directions AND for rays that lie IN the planes of the scene's walls -- second
    generation rays from hit points towards a sun whose direction has exact zero components (+0 and -0), origins exactly on a plane, where (corner - o) * inv is
    0 * 1e30 and the reference's slab test is all NaNs and lets the ray through (geom/aabb.h:39-54).  (Round 5 found a variant of the error bound that culled
    exactly those rays only on the device: this test is that bug's, and runs without one.)"""
    from raylib_amd import binding
    d = os.path.join(str(workdir), "walk8_%s_%s" % (make.__name__, "_".join(str(v) for v in kw.values()))); os.makedirs(d, exist_ok=True)
    obj = make(os.path.join(d, "w.obj"), **kw)[0]
    if kw.get("tess") == 64:
        monkeypatch.setenv("RAYLIB_W8_SPLIT", "1")          # (rl_bvh.cc SplitLeaves: an option of the builder, off by default)
    ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1.0)
    monkeypatch.delenv("RAYLIB_W8_SPLIT", raising=False)
    assert lib.RaylibAMD_SceneBVH8Info(ses.scene, None, None, None, None) == 1
    flat = helpers.objflat.load_obj(obj, oracle)
    sc = oracle.scene_create(flat, 1)
    lo, hi = flat.triangles["v0"].min(0), flat.triangles["v0"].max(0)
    rng = np.random.RandomState(5)
    n = 6000
    o = rng.uniform(lo + 0.02 * (hi - lo), hi - 0.02 * (hi - lo), (n, 3)).astype(np.float32)
    dd = rng.normal(size=(n, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    rays = np.concatenate([o, dd.astype(np.float32)], 1).astype(np.float32)
    first = oracle.closest_hit(sc, rays, 1e-4)
    p = first["p"][first["hit"] == 1][:3000]
    groups = [("random", rays)]
    for k, sun in enumerate(((1.0, 1.0, -0.0), (-1.0, 0.0, 1.0), (0.0, -1.0, -1.0), (1.0, 0.0, 0.0), (-0.0, -0.0, -1.0), (0.0, 1.0, -0.0))):
        sd = np.asarray(sun, np.float32)
        sd = np.where(sd == 0, sd, sd / np.float32(np.sqrt(float((sd * sd).sum())))).astype(np.float32)     # (keeps the signed zeros)
        groups.append(("in-plane %d" % k, np.concatenate([p, np.repeat(sd[None], len(p), 0)], 1).astype(np.float32)))
    for name, R in groups:
        R = np.ascontiguousarray(R, np.float32)
        h = oracle.closest_hit(sc, R, 1e-4)
        hit = h["hit"] == 1
        tmax = np.where(hit, h["t"] * np.float32(1.00002), np.float32(3.4e38)).astype(np.float32)
        out_t = np.zeros(len(R), np.float32); steps = np.zeros(len(R), np.uint32)
        assert lib.RaylibAMD_SceneWalk8Host(ses.scene, R.ctypes.data_as(C.POINTER(C.c_float)), len(R), 1e-4, tmax.ctypes.data_as(C.POINTER(C.c_float)),
                                            out_t.ctypes.data_as(C.POINTER(C.c_float)), steps.ctypes.data_as(C.POINTER(C.c_uint32))) == 1
        # (rays that graze the triangle they hit -- |n . d| tiny: the reference's float plane formula and the restatement's double-precision test can disagree about
        #  such a hit; the box arithmetic, which is what is tested, reaches the leaf either way -- are left out)
        grazing = np.abs((h["n"].astype(np.float64) * R[:, 3:6]).sum(1)) < 1e-3
        missed = hit & ~grazing & ~(out_t <= h["t"] * np.float32(1.0001) + np.float32(1e-5))     # (the restatement's distances are double precision, the reference's float)
        assert not missed.any(), "%s: the walk skipped the closest hit of %d of %d rays, e.g. %s (oracle t %s)" % (name, missed.sum(), hit.sum(), R[missed][0], h["t"][missed][0])
        assert (name != "random" or hit.sum() > len(R) // 10) and steps.mean() >= 1
    oracle.scene_destroy(sc)
    ses.close()


def test_bvh_build_is_the_same_tree_for_any_thread_count(lib, workdir, monkeypatch):
    """Above 65536 primitives the build uses every host thread (top levels: parallel binning; sub-trees: tasks);
    the flat tree must be byte-identical to the single-threaded one (traversal order, hence tie behaviour, depends on it)."""
    from raylib_amd import binding
    obj, n = scenes.cornell(os.path.join(str(workdir), "bvh_mt.obj"), tess=64, displace_fraction=0.2)
    assert n > 65536
    seen = {}
    for threads in ("1", "2", "5", "8"):
        monkeypatch.setenv("RAYLIB_BUILD_THREADS", threads)
        ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1.0)
        nodes, depth, sah = C.c_uint32(), C.c_uint32(), C.c_float()
        assert lib.RaylibAMD_SceneBVHInfo(ses.scene, C.byref(nodes), C.byref(depth), C.byref(sah)) == 1
        n4, need = C.c_uint32(), C.c_uint32()
        assert lib.RaylibAMD_SceneBVH4Info(ses.scene, C.byref(n4), C.byref(need)) == 1          # the wide tree exists and is valid
        assert nodes.value / 4 < n4.value < nodes.value and depth.value / 2 <= need.value <= 3 * depth.value
        seen[threads] = (lib.RaylibAMD_SceneBVHHash(ses.scene), nodes.value, depth.value, sah.value, n4.value, need.value)
        ses.close()
    assert len(set(seen.values())) == 1, seen


def test_transform_rules(lib, workdir):
    """rotate -> scale -> translate on positions, rotation only on normals; ignored after finalize
    (reference raylib.cc:71-90, static_mesh.cc:54-78)."""
    obj, _ = scenes.cornell(os.path.join(str(workdir), "xf.obj"))
    h = lib.Raylib_LoadOBJModel(obj.encode())
    lib.Raylib_TransformOBJModel(h, 1.0, 2.0, 3.0, 90.0, 0.0, 0.0, 2.0, 2.0, 2.0)
    sc = lib.Raylib_CreateScene(); lib.Raylib_AddOBJModelToScene(sc, h); lib.Raylib_FinalizeOBJModel(h)
    lib.Raylib_TransformOBJModel(h, 100.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0)   # ignored: locked
    lib.Raylib_FinalizeScene(sc)
    from raylib_amd import binding
    tris = np.zeros(36, ffi.TRI_DTYPE)
    lib.RaylibAMD_SceneExportTriangles(sc, tris.ctypes.data)
    base = helpers.objflat.load_obj(obj, ffi.load_oracle()).triangles
    # yaw 90: (x, y, z) -> (-z*?, y, x*?) with the reference matrix: x' = ch*x - sh*z, z' = sh*x + ch*z (ch ~ 0, sh = 1)
    v = base["v0"]
    want = np.stack([-v[:, 2], v[:, 1], v[:, 0]], axis=1) * 2.0 + np.array([1.0, 2.0, 3.0])
    assert np.allclose(tris["v0"], want, atol=1e-5)
    n = base["n0"]
    assert np.allclose(tris["n0"], np.stack([-n[:, 2], n[:, 1], n[:, 0]], axis=1), atol=1e-6)
    lib.Raylib_DestroyScene(sc); lib.Raylib_UnloadOBJModel(h)


def test_image_codecs_roundtrip(lib, workdir):
    rgba = np.zeros((6, 7, 4), np.float32)
    rng = np.random.RandomState(2)
    rgba[..., :3] = rng.randint(0, 256, (6, 7, 3)) / np.float32(255.0)
    rgba[..., 3] = 1.0
    ih = lib.RaylibAMD_CreateImageFromData(7, 6, rgba.ctypes.data_as(C.POINTER(C.c_float)))
    for ftype, ext in ((0, "bmp"), (2, "png")):
        path = os.path.join(str(workdir), "rt." + ext).encode()
        assert lib.Raylib_WriteImageToDisk(ih, path, ftype) == 1
        back = lib.Raylib_LoadImage(path)
        assert back
        got = np.zeros((6, 7, 4), np.float32)
        lib.RaylibAMD_DumpImageRGBA(back, got.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(got, rgba)                    # row 0 stays the top row through both codecs
        lib.Raylib_DestroyImage(back)
    assert lib.Raylib_WriteImageToDisk(ih, os.path.join(str(workdir), "x.jpg").encode(), 1) == 1   # baseline JPEG (tests/test_image_codecs.py)
    assert lib.Raylib_WriteImageToDisk(ih, b"/tmp/x.png", 3) == 0      # invalid type (raylib.cc:316)
    assert lib.Raylib_WriteImageToDisk(None, b"/tmp/x.png", 2) == 0
    lib.Raylib_DestroyImage(ih)


def test_radiance_hdr_decode(lib, workdir):
    """Radiance RGBE (sky panoramas): flat and new-style RLE scanlines decode to mantissa * 2^(e-136), alpha 1, row 0 = top."""
    rng = np.random.RandomState(9)
    w, h = 16, 5
    rgbe = rng.randint(1, 256, (h, w, 4)).astype(np.uint8)
    rgbe[..., 3] = rng.randint(120, 140, (h, w))
    rgbe[0, 0] = (0, 0, 0, 0)
    want = np.zeros((h, w, 4), np.float32)
    f = np.ldexp(np.float32(1.0), rgbe[..., 3].astype(np.int32) - 136).astype(np.float32)
    for c in range(3):
        want[..., c] = np.where(rgbe[..., 3] > 0, rgbe[..., c].astype(np.float32) * f, 0)
    want[..., 3] = 1.0
    header = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)
    flat = header + rgbe.tobytes()
    rle = bytearray(header)
    for y in range(h):
        rle += bytes([2, 2, w >> 8, w & 255])
        for ch in range(4):
            row = rgbe[y, :, ch]
            rle += bytes([8]) + row[:8].tobytes() + bytes([128 + 8, row[8]])    # 8 literals, then a run of 8
            rgbe[y, 8:, ch] = row[8]
    f = np.ldexp(np.float32(1.0), rgbe[..., 3].astype(np.int32) - 136).astype(np.float32)
    want_rle = want.copy()
    for c in range(3):
        want_rle[..., c] = np.where(rgbe[..., 3] > 0, rgbe[..., c].astype(np.float32) * f, 0)
    for name, blob, expect in (("flat.hdr", flat, want), ("rle.hdr", bytes(rle), want_rle)):
        path = os.path.join(str(workdir), name)
        open(path, "wb").write(blob)
        ih = lib.Raylib_LoadImage(path.encode())
        assert ih, name
        got = np.zeros((h, w, 4), np.float32)
        lib.RaylibAMD_DumpImageRGBA(ih, got.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(got, expect), name
        lib.Raylib_DestroyImage(ih)


def test_postprocess_matches_oracle(lib, oracle):
    rng = np.random.RandomState(4)
    rgba = np.zeros((9, 11, 4), np.float32)
    rgba[..., :3] = rng.gamma(1.0, 0.8, (9, 11, 3)).astype(np.float32)
    rgba[0, 0, :3] = 0.0; rgba[1, 1, :3] = (9.0, 12.0, 3.0)
    rgba[..., 3] = 1.0
    ih = lib.RaylibAMD_CreateImageFromData(11, 9, rgba.ctypes.data_as(C.POINTER(C.c_float)))
    lib.Raylib_PostProcess(ih)
    got = np.zeros_like(rgba)
    lib.RaylibAMD_DumpImageRGBA(ih, got.ctypes.data_as(C.POINTER(C.c_float)))
    want = oracle.postprocess(rgba)
    assert np.array_equal(helpers.bits(got), helpers.bits(want))
    lib.Raylib_DestroyImage(ih)


def test_procedural_elements_registry(lib):
    """Materials / elements made by the library are accepted by Raylib_AddSceneElement; anything else is refused."""
    import ctypes as C
    f3 = lambda *v: (C.c_float * 3)(*v)
    m = lib.RaylibAMD_CreateMaterial(0, f3(2.0, 0.5, -1.0), 0.0, 0.0, None, 0.0, None, 0.0)
    assert m and lib.RaylibAMD_CreateMaterial(9, None, 0, 0, None, 0, None, 0) is None
    sp = lib.RaylibAMD_CreateSphere(0.0, 0.0, 0.0, 1.0, m)
    cu = lib.RaylibAMD_CreateCube(f3(0, 0, 0), f3(1, 1, 1), 0.0, f3(0, 0, 0), m)
    tr = lib.RaylibAMD_CreateTriangle(f3(0, 0, 0), f3(1, 0, 0), f3(0, 1, 0), f3(0, 0, 1), f3(0, 0, 1), f3(0, 0, 1), None, m)
    assert sp and cu and tr
    assert lib.RaylibAMD_CreateSphere(0.0, 0.0, 0.0, 1.0, 12345) is None          # not a material of this library
    sc = lib.Raylib_CreateScene()
    for e in (sp, cu, tr):
        lib.Raylib_AddSceneElement(sc, e)
    lib.Raylib_AddSceneElement(sc, 0xdeadbeef)                                     # foreign pointer: ignored with a log line
    lib.Raylib_FinalizeScene(sc)
    assert lib.RaylibAMD_SceneNumTriangles(sc) == 1 and lib.RaylibAMD_SceneNumMaterials(sc) == 3
    mats = np.zeros(3, ffi.MAT_DTYPE)
    lib.RaylibAMD_SceneExportMaterials(sc, mats.ctypes.data)
    assert np.array_equal(mats["albedo"][0], np.array([1.0, 0.5, 0.0], np.float32))   # Lambertian saturates (material.h:79-82)
    nodes, depth, sah = C.c_uint32(), C.c_uint32(), C.c_float()
    assert lib.RaylibAMD_SceneBVHInfo(sc, C.byref(nodes), C.byref(depth), C.byref(sah)) == 1 and nodes.value >= 2
    assert lib.Raylib_DestroyScene(sc) == 1
    for e in (sp, cu, tr):
        assert lib.RaylibAMD_DestroySceneElement(e) == 1 and lib.RaylibAMD_DestroySceneElement(e) == 0
    assert lib.RaylibAMD_DestroyMaterial(m) == 1 and lib.RaylibAMD_DestroyMaterial(m) == 0


def test_cell_math(lib):
    from raylib_amd import tiling
    for (w, h) in ((64, 64), (40, 28), (1920, 1080), (7, 9)):
        assert lib.RaylibAMD_NumCells(w, h) == tiling.num_cells(w, h)
        assert lib.RaylibAMD_CellBufferFloats(w, h, 0, 1) == w * h * 4
        for world in (2, 3, 8):
            tot = 0
            for r in range(world):
                assert lib.RaylibAMD_CellBufferFloats(w, h, r, world) == tiling.local_cells(w, h, r, world) * 256
                tot += tiling.local_cells(w, h, r, world)
            assert tot == tiling.num_cells(w, h)


def test_render_without_device_fails_loudly(lib, workdir, capfd):
    """No CPU fallback: on a box without a HIP device the render entry reports failure."""
    if lib.RaylibAMD_DeviceAvailable():
        pytest.skip("a device is present")
    ses = helpers.session_for_case(lib, "cornell", workdir)
    from raylib_amd import binding
    st = ses.settings(8, 8, 1)
    assert lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, 0, 1, None) == 0
    img = ses.render(8, 8, 1)
    assert (img[..., :3] == 0).all()
    assert "FAILED" in capfd.readouterr().err
    assert lib.Raylib_Initialize() == 0
    ses.close()


def test_obj_number_reader_is_correctly_rounded(lib):
    """The OBJ parser reads numbers with exact fast paths in front of strtof (csrc/rl_obj_loader.cc ParseFloat); every value
    must be the float32 nearest to the decimal (ties to even) -- checked with rational arithmetic, not with another parser."""
    import random, struct
    from fractions import Fraction

    def nearest_f32(txt):
        x = Fraction(txt)
        if x == 0:
            return 0.0
        approx = np.float32(float(x))                       # within one ulp; pick the true nearest among its neighbours
        cands = {float(approx), float(np.nextafter(approx, np.float32(np.inf))), float(np.nextafter(approx, np.float32(-np.inf)))}
        best = min(cands, key=lambda c: (abs(Fraction(c) - x), struct.unpack("<I", struct.pack("<f", c))[0] & 1))
        return best

    rnd = random.Random(11)
    texts = ["0", "-0", "1", "0.1", "16777216", "16777217", "16777217.000000001", "0.333333343267440796", "1.00000005960464477539",
             "8388608.5", "8388609.5", "0.000000000116415321826934814453125", "123456789012345.6789", "-2.5", "+7.25", "00012.5000"]
    for _ in range(20000):
        digits = rnd.randint(1, 16)
        m = rnd.randrange(10 ** digits)
        frac = rnd.randint(0, min(22, digits + 6))
        t = str(m).rjust(frac + 1, "0")
        t = (t[:-frac] + "." + t[-frac:]) if frac else t
        texts.append(("-" if rnd.random() < 0.3 else "") + t)
    # decimal midpoints of adjacent floats (the double-rounding trap): exact halfway strings
    for _ in range(300):
        f = np.float32(rnd.uniform(0.001, 1000.0))
        g = np.nextafter(f, np.float32(np.inf))
        mid = (Fraction(float(f)) + Fraction(float(g))) / 2
        num, den = mid.numerator, mid.denominator            # den is a power of two: finite decimal expansion
        k = den.bit_length() - 1
        texts.append(str(Fraction(num * 5 ** k, 10 ** k).numerator).rjust(k + 1, "0")[:-k] + "." + str(num * 5 ** k).rjust(k + 1, "0")[-k:] if k else str(num))
    bad = []
    for t in texts:
        got = lib.RaylibAMD_ParseFloat(t.encode())
        want = nearest_f32(t.lstrip("+"))
        if struct.pack("<f", got) != struct.pack("<f", want) and not (got == 0.0 and want == 0.0):
            bad.append((t, got, want))
    assert not bad, bad[:5]


def test_obj_reader_result_does_not_depend_on_its_chunking(lib, workdir, monkeypatch):
    """The OBJ text is parsed in chunks of whole lines on several threads (files from 4 MB on).  State that a one-pass reader carries
    from line to line -- element counts behind relative indices and the defined-before-use check, the material in force, mtllib
    before/after usemtl, the o/g shape number, dropped polygons -- must come out the same for every chunking, including chunks
    that hold a single line or nothing."""
    from raylib_amd import binding
    d = os.path.join(str(workdir), "chunks"); os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "a.mtl"), "w") as f:
        f.write("newmtl red\nKd 0.6 0.1 0.1\nNs 10\nnewmtl mirror\nKd 0.9 0.9 0.9\nillum 3\n")
    with open(os.path.join(d, "b.mtl"), "w") as f:
        f.write("newmtl late\nKd 0.1 0.2 0.7\nNs 50\n")
    rng = np.random.RandomState(5)
    lines = ["# faces before any shape or material", "usemtl red   # not known yet: fallback material", "v 0 0 0", "v 1 0 0", "v 0 1 0", "f 1 2 3",
             "f 1 2 4", "v 1 1 0", "mtllib a.mtl", "usemtl red", "f -4 -3 -1 -2", "o first", "o empty", "g second",
             "vt 0.25 0.5", "vn 0 0 1", "f 1/1/1 2//1 3/1", "f 1/2/1 2/1/1 3/1/1", "vt 0.75 0.125",
             "usemtl nosuch", "f 1 2 4 3 1", "usemtl", "f 2 3 4", "f 1 2", "usemtl late", "f 1 3 4", "mtllib missing.mtl b.mtl a.mtl", "usemtl late", "g third", "f 4/1/1 3/-1/-1 1/-2/5",
             "f 1 2 99999"]
    for i in range(400):                                   # bulk, so that forced chunkings cut everywhere
        x, y, z = rng.rand(3)
        lines += ["v %.6f %.6f %.6f" % (x, y, z), "v %.6f %.6f %.6f" % (x + 0.1, y, z), "v %.6f %.6f %.6f" % (x, y + 0.1, z)]
        if i % 7 == 0: lines.append("vn %.4f %.4f %.4f" % tuple(rng.rand(3)))
        if i % 5 == 0: lines.append("usemtl %s" % ["red", "mirror", "late", "gone"][i // 5 % 4])
        if i % 11 == 0: lines.append("o part%d" % i)
        lines.append("f -3//-1 -2//-1 -1//-1" if i % 3 else "f -3 -2 -1")
    obj = os.path.join(d, "tricky.obj")
    with open(obj, "w") as f:
        f.write("\n".join(lines) + "\n")
    seen = {}
    for chunks, threads in (("1", "1"), ("2", "2"), ("3", "8"), ("17", "4"), ("400", "8"), ("100000", "3")):
        monkeypatch.setenv("RAYLIB_PARSE_CHUNKS", chunks)
        monkeypatch.setenv("RAYLIB_BUILD_THREADS", threads)
        ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1.0)
        tris, mats = ses.export_flat()
        seen[chunks] = (tris.tobytes(), mats.tobytes())
        if chunks == "1":
            first = tris
        ses.close()
    assert len(set(seen.values())) == 1
    # spot checks of the one-pass rules on the single-chunk result
    assert len(first) == 1 + 1 + 2 + 2 + 3 + 1 + 1 + 1 + 400     # "f 1 2" (two corners) and "f 1 2 99999" (beyond the file's last vertex) are dropped
    mat = first["material"]
    assert list(mat[:12]) == [3, 3, 0, 0, 0, 0, 3, 3, 3, 3, 3, 2]   # red = 0, mirror = 1, late = 2 (only once b.mtl was read), fallback = 3: name not
                                                                    # known yet, "usemtl nosuch", bare "usemtl", "late" before its mtllib
    assert list(first["shape"][:12]) == [0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2]   # before any o/g; "second" ("first" and "empty" hold no faces: dropped); "third"
    assert np.array_equal(first["v2"][1], np.float32([1, 1, 0]))            # "f 1 2 4": vertex 4 is defined after the face that names it
    assert np.array_equal(first["st"][5][:2], np.float32([0.75, 0.125]))    # vt 2, likewise
    # unit square 1 2 4 3 as "f -4 -3 -1 -2": both diagonals are equally long -> tinyobjloader's `else` branch [0,1,3] [1,2,3]
    assert np.array_equal(first["v0"][2], np.float32([0, 0, 0])) and np.array_equal(first["v2"][2], np.float32([0, 1, 0]))
    assert np.array_equal(first["v0"][3], np.float32([1, 0, 0])) and np.array_equal(first["v1"][3], np.float32([1, 1, 0]))
    # the independent Python reader (oracle/objflat.py) agrees on every triangle and material
    flat = helpers.objflat.load_obj(obj, helpers.ffi.load_oracle())
    assert first.tobytes() == flat.triangles.tobytes()


@pytest.mark.parametrize("bad", ["f 1 2 0", "f 1/0/1 2/1/1 3/1/1", "f 1 2 -9", "f 1 x 3", "f 1 2 3 # trailing comment"])
def test_obj_face_statements_that_fail_the_load(bad, lib, workdir):
    """tinyobjloader's parseTriple / fixIndex return false on index 0 (also what atoi makes of a word that is not a number -- a '#'
    does not end a face statement) and on a relative index before the first element; LoadObj then fails and the reference refuses the
    model (loader/obj_loader.cc:91-95)."""
    d = os.path.join(str(workdir), "badobj"); os.makedirs(d, exist_ok=True)
    obj = os.path.join(d, "bad.obj")
    with open(obj, "w") as f:
        f.write("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\nf 1 2 3\n" + bad + "\n")
    assert not lib.Raylib_LoadOBJModel(obj.encode())
    with pytest.raises(helpers.objflat.ObjLoadError):
        helpers.objflat.load_obj(obj, helpers.ffi.load_oracle())


def test_chunked_obj_reader_matches_oracle_side_parser_on_a_multi_megabyte_file(lib, oracle, workdir, monkeypatch):
    """A file above the 4 MB threshold goes through the default chunking (several chunks, several threads); the flat scene must be
    the one the independent Python parser produces from the same text."""
    from raylib_amd import binding
    obj, n = scenes.cornell(os.path.join(str(workdir), "chunked_big.obj"), tess=24, displace_fraction=0.2)
    assert os.path.getsize(obj) > (4 << 20), os.path.getsize(obj)
    monkeypatch.setenv("RAYLIB_BUILD_THREADS", "4")
    monkeypatch.delenv("RAYLIB_PARSE_CHUNKS", raising=False)
    flat = helpers.objflat.load_obj(obj, oracle)
    ses = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, 1.0)
    tris, mats = ses.export_flat()
    ses.close()
    assert len(tris) == n == len(flat.triangles)
    assert tris.tobytes() == flat.triangles.tobytes() and mats.tobytes() == flat.materials.tobytes()


def test_null_arguments_are_survivable(lib):
    """Errors are return codes, never a crash across the C ABI (reference raylib.cc checks only the destroy paths; a null handle there is UB,
    here every export takes null handles / null pointers and returns).  Run under ASan + UBSan by tools/asan_host_check.sh."""
    from raylib_amd import binding
    st = binding.RendererSettings(16, 16, 1, 5, 1e-4, 0)
    calls = [
     ("Raylib_LoadOBJModel", (None,)), ("Raylib_LoadOBJModel", (b"/nonexistent.obj",)), ("Raylib_TransformOBJModel", (0, 0,0,0, 0,0,0, 1,1,1)), ("Raylib_FinalizeOBJModel", (0,)), ("Raylib_UnloadOBJModel", (0,)),
     ("Raylib_LoadImage", (None,)), ("Raylib_LoadImage", (b"/nonexistent.png",)), ("Raylib_AddSceneElement", (0, 0)), ("Raylib_AddOBJModelToScene", (0, 0)), ("Raylib_SetSkyPanorama", (0, 0)),
     ("Raylib_SetSunIlluminance", (0, 1.0, 1.0, 1.0)), ("Raylib_SetSunDirection", (0, 0.0, -1.0, 0.0)), ("Raylib_FinalizeScene", (0,)), ("Raylib_DestroyScene", (0,)),
     ("Raylib_CameraSetPosition", (0, 0.0, 0.0, 0.0)), ("Raylib_CameraSetLookAt", (0, 0.0, 0.0, 0.0)), ("Raylib_CameraSetPerspective", (0, 45.0, 1.0)), ("Raylib_CameraSetLens", (0, 0.0, 1.0)), ("Raylib_CameraSetMotion", (0, 0.0, 0.0)),
     ("Raylib_CameraCopy", (0, 0)), ("Raylib_DestroyCamera", (0,)), ("Raylib_DumpImageData", (0, None)), ("Raylib_DestroyImage", (0,)), ("Raylib_Render", (None, 0, 0, 0)), ("Raylib_Render", (C.byref(st), 0, 0, 0)),
     ("Raylib_Denoise", (0, 0, 0, 0, 0)), ("Raylib_PostProcess", (0,)), ("Raylib_GetRenderModeString", (99,)), ("Raylib_WriteImageToDisk", (0, None, 0)), ("Raylib_WriteImageToDisk", (0, b"/tmp/x.png", 7)),
     ("RaylibAMD_RenderDevice", (None, 0, 0, 0, 1, None)), ("RaylibAMD_RenderDevice", (C.byref(st), 0, 0, 0, 1, None)), ("RaylibAMD_SceneNumTriangles", (0,)), ("RaylibAMD_SceneBVHHash", (0,)),
     ("RaylibAMD_ImageSize", (0, None, None)), ("RaylibAMD_DumpImageRGBA", (0, None)), ("RaylibAMD_DestroyMaterial", (0,)), ("RaylibAMD_DestroySceneElement", (0,)), ("RaylibAMD_GetLastStats", (None,)),
     ("RaylibAMD_ParseFloat", (None,)), ("RaylibAMD_CameraExport", (0, None)), ("RaylibAMD_OBJModelSetTexture", (0, None, 0, 0)),
    ]
    # with live objects but null partners
    img = lib.Raylib_CreateImage(8, 8); sc = lib.Raylib_CreateScene(); cam = lib.Raylib_CreateCamera()
    calls += [("Raylib_Render", (C.byref(st), sc, 0, img)), ("Raylib_Render", (C.byref(st), 0, cam, img)), ("Raylib_Render", (C.byref(st), sc, cam, 0)), ("Raylib_SetSkyPanorama", (sc, 0)),
              ("Raylib_AddOBJModelToScene", (sc, 0)), ("Raylib_AddSceneElement", (sc, 0)), ("Raylib_DumpImageData", (img, None)), ("Raylib_WriteImageToDisk", (img, None, 0)), ("Raylib_CameraCopy", (cam, 0)), ("Raylib_CameraCopy", (0, cam))]
    for name, args in calls:
        getattr(lib, name)(*args)
    for h, fn in ((img, lib.Raylib_DestroyImage), (sc, lib.Raylib_DestroyScene), (cam, lib.Raylib_DestroyCamera)):
        assert fn(h) == 1


def test_cells_dropped_from_the_job_list_cannot_see_the_box(lib, oracle):
    """csrc/rl_cull.cc: with a pinhole camera and no sky panorama the renderer leaves out of the megakernel's job list every 8 x 8 cell none of whose camera
    rays can meet the scene's bounding box, and fills it with the miss shader's constant.  The decision is a rectangle on the image plane with a margin;
    what must hold is the geometry: NO ray of a dropped cell -- any pixel of it, any jitter in (-1, 1) pixel -- meets the box.  Random cameras and boxes;
    the rays are the oracle's (reference camera.h:44-53), the box test is exact arithmetic on them in float64."""
    pass
    rng = np.random.RandomState(11)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    dropped_cases = ineligible = lens_cases = 0
    for case in range(60):
        w, h = int(rng.randint(40, 400)), int(rng.randint(30, 300))
        lo = rng.uniform(-2, 0, 3); hi = lo + rng.uniform(0.2, 3.0, 3)
        centre = (lo + hi) / 2
        dist = float(rng.uniform(0.5, 30.0))
        direction = rng.normal(size=3); direction /= np.linalg.norm(direction)
        origin = centre + direction * dist
        look = centre + rng.normal(size=3) * float(rng.uniform(0.0, 1.5))
        fov, aspect = float(rng.uniform(15, 90)), w / h
        # every third camera has a thin lens (the console front-end's aperture is 0.01, reference src/main.cc:24; here up to a fifth of the box's size), focused
        # in front of, inside or behind the box: the dropped cells must be clear of the circle of confusion too
        aperture = float(rng.uniform(0.005, 0.4)) if case % 3 == 2 else 0.0
        focal = float(rng.uniform(0.3, 2.0)) * dist
        cam = lib.Raylib_CreateCamera()
        lib.Raylib_CameraSetPosition(cam, *[float(x) for x in origin]); lib.Raylib_CameraSetLookAt(cam, *[float(x) for x in look])
        lib.Raylib_CameraSetPerspective(cam, fov, aspect)
        lib.Raylib_CameraSetLens(cam, aperture, focal)
        bounds = np.concatenate([lo, hi]).astype(np.float32)
        cx, cy = (w + 7) // 8, (h + 7) // 8
        empty = np.zeros(cx * cy, np.uint8)
        const = np.zeros(3, np.float32)
        n = lib.RaylibAMD_CullCells(cam, fp(bounds), None, None, w, h, empty.ctypes.data_as(C.POINTER(C.c_uint8)), fp(const))
        lib.Raylib_DestroyCamera(cam)
        inside = bool(np.all(origin > lo - 1e-3 - aperture) and np.all(origin < hi + 1e-3 + aperture))
        if n < 0:
            ineligible += 1
            continue
        assert not inside
        assert n == int(empty.sum()) and (const == 0).all()
        if n == 0:
            continue
        dropped_cases += 1
        ocam = ffi.make_camera(tuple(float(x) for x in origin), tuple(float(x) for x in look), fov, aspect, aperture, focal)
        ys, xs = np.nonzero(empty.reshape(cy, cx))
        uv = []
        for (cyy, cxx) in zip(ys, xs):
            # the cell's four corner pixels and its centre, each at the jitter's extremes and at none
            for px in (8 * cxx, min(8 * cxx + 7, w - 1), 8 * cxx + 3):
                for py in (8 * cyy, min(8 * cyy + 7, h - 1), 8 * cyy + 4):
                    if px >= w or py >= h:
                        continue
                    for jx in (-0.999, 0.0, 0.999):
                        for jy in (-0.999, 0.0, 0.999):
                            uv.append(((px + jx) / w, (py + jy) / h))
        uv = np.asarray(uv, np.float32)
        if aperture > 0.0:
            # lens points are drawn per ray: many draws per (u, v), plus the rim of the lens by hand (the reference's r = sqrt(u1) reaches it only in the limit)
            lens_cases += 1
            uv = np.tile(uv[:: max(1, len(uv) // 4000)], (12, 1))
        rays = oracle.camera_rays(ocam, uv, seed=1 + case)[:, :6].astype(np.float64)
        if aperture > 0.0:
            cam19 = np.zeros(19, np.float32)
            c2 = lib.Raylib_CreateCamera()
            lib.Raylib_CameraSetPosition(c2, *[float(x) for x in origin]); lib.Raylib_CameraSetLookAt(c2, *[float(x) for x in look])
            lib.Raylib_CameraSetPerspective(c2, fov, aspect); lib.Raylib_CameraSetLens(c2, aperture, focal)
            lib.RaylibAMD_CameraExport(c2, fp(cam19)); lib.Raylib_DestroyCamera(c2)
            O, R, TL, Hh, Vv, cu, cv = cam19[0:3].astype(np.float64), float(cam19[3]), cam19[4:7].astype(np.float64), cam19[7:10].astype(np.float64), cam19[10:13].astype(np.float64), cam19[13:16].astype(np.float64), cam19[16:19].astype(np.float64)
            sub = uv[: len(uv) // 12].astype(np.float64)
            rim = []
            for ang in np.linspace(0.0, 2 * np.pi, 8, endpoint=False):
                L = O + R * (np.cos(ang) * cu + np.sin(ang) * cv)
                F = TL + sub[:, :1] * Hh + (1.0 - sub[:, 1:2]) * Vv
                dd = F - L
                rim.append(np.concatenate([np.broadcast_to(L, dd.shape), dd / np.linalg.norm(dd, axis=1, keepdims=True)], axis=1))
            rays = np.concatenate([rays] + rim)
        o, d = rays[:, :3], rays[:, 3:]
        blo, bhi = bounds[:3].astype(np.float64), bounds[3:].astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            t0, t1 = (blo - o) / d, (bhi - o) / d
        tn = np.nanmax(np.minimum(t0, t1), axis=1); tf = np.nanmin(np.maximum(t0, t1), axis=1)
        hits = (tf >= np.maximum(tn, 0.0))
        assert not hits.any(), "case %d: %d rays of dropped cells meet the box" % (case, hits.sum())
    assert dropped_cases >= 20 and lens_cases >= 5, (dropped_cases, ineligible, lens_cases)
    # a camera inside the box (a corner beside or behind it): no rectangle bounds the box, the frame is not eligible
    cam = lib.Raylib_CreateCamera()
    lib.Raylib_CameraSetPosition(cam, 0.0, 1.0, 0.5); lib.Raylib_CameraSetLookAt(cam, 0.0, 1.0, -1.0); lib.Raylib_CameraSetPerspective(cam, 70.0, 200 / 120)
    assert lib.RaylibAMD_CullCells(cam, fp(np.asarray([-1, 0, -1, 1, 2, 1], np.float32)), None, None, 200, 120, None, None) == -1
    lib.Raylib_DestroyCamera(cam)
    # with a sun whose ray from the camera misses the box the constant is the sun's illuminance; when it may hit the box nothing is dropped
    cam = lib.Raylib_CreateCamera()
    lib.Raylib_CameraSetPosition(cam, 0.0, 1.0, 14.0); lib.Raylib_CameraSetLookAt(cam, 0.0, 1.0, -1.0); lib.Raylib_CameraSetPerspective(cam, 45.0, 200 / 120)
    bounds = np.asarray([-1, 0, -1, 1, 2, 1], np.float32)
    empty = np.zeros(25 * 15, np.uint8); const = np.zeros(3, np.float32)
    sun = np.asarray([9, 8, 7], np.float32)
    sdir = np.asarray([-1, -1, 0], np.float32) / np.float32(np.sqrt(2))
    assert lib.RaylibAMD_CullCells(cam, fp(bounds), fp(sun), fp(sdir), 200, 120, empty.ctypes.data_as(C.POINTER(C.c_uint8)), fp(const)) > 100
    assert (const == sun).all()
    sdir = np.asarray([0, 0, 1], np.float32)      # the sun behind the box as seen from the camera: its ray runs through the box
    assert lib.RaylibAMD_CullCells(cam, fp(bounds), fp(sun), fp(sdir), 200, 120, empty.ctypes.data_as(C.POINTER(C.c_uint8)), fp(const)) == -1
    lib.Raylib_DestroyCamera(cam)


def test_sanitizer_stub_covers_the_device_interface():
    """tools/nodevice_stub.cc stands in for the HIP translation units in the ASan / UBSan build of the host side (tools/asan_host_check.sh): every Device*
    function rl_host.h declares needs a definition there, or that build stops linking the day an entry point is added (it did, twice)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    decl = open(os.path.join(root, "software-raytracing_amd", "csrc", "rl_host.h")).read()
    stub = open(os.path.join(root, "tools", "nodevice_stub.cc")).read()
    declared = set(re.findall(r"^\s*(?:bool|void|void\*|int|int32_t)\s+(Device[A-Za-z0-9_]+)\s*\(", decl, re.M))
    defined = set(re.findall(r"\b(Device[A-Za-z0-9_]+)\s*\(", stub))
    assert len(declared) >= 10, declared
    assert declared <= defined, sorted(declared - defined)
