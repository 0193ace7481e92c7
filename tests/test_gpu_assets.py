"""Asset ingestion on the GPU path (reference loader/obj_loader.cc:247-292 PreloadImages, render/image.cc:152-231,
render/renderer.cc:159-181): a scene whose MTL names its maps as TGA, BMP, JPEG, PNG and Radiance HDR files, and a sky panorama loaded
from an .hdr file through Raylib_LoadImage -> Raylib_SetSkyPanorama.  The files are written by encoders that are not the product's
(Pillow; a numpy RGBE packer), the texels the oracle renders with are decoded by decoders that are not the product's either (Pillow --
for JPEG driven the way FreeImage drives libjpeg, see tests/test_image_codecs.py -- and the RGBE formula), and the HIP path, which
decodes every file itself, must produce the oracle's image bit for bit."""
import ctypes as C
import io
import os
import numpy as np
import pytest

import helpers
from helpers import ffi, scenes, objflat

PIL = pytest.importorskip("PIL")
from PIL import Image   # noqa: E402

pytestmark = pytest.mark.gpu


def rgbe_pack(img):
    """float RGB (H, W, 3) -> RGBE bytes (H, W, 4), Greg Ward's float2rgbe."""
    m = img.max(-1)
    mant, exp = np.frexp(m)
    scale = np.where(m > 1e-32, mant * 256.0 / np.maximum(m, 1e-38), 0.0)
    out = np.zeros(img.shape[:2] + (4,), np.uint8)
    out[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m > 1e-32, exp + 128, 0).astype(np.uint8)
    return out


def rgbe_unpack(rgbe):
    """What FreeImage's HDR plugin (and the product) make of RGBE: mantissa * 2^(e - 136), alpha 1."""
    f = np.ldexp(np.float32(1.0), rgbe[..., 3].astype(np.int32) - 136).astype(np.float32)
    out = np.ones(rgbe.shape[:2] + (4,), np.float32)
    for c in range(3):
        out[..., c] = np.where(rgbe[..., 3] > 0, rgbe[..., c].astype(np.float32) * f, np.float32(0))
    return out


def write_hdr(path, rgbe):
    h, w = rgbe.shape[:2]
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w) + rgbe.tobytes())


def byte_image(u8):
    return (u8.astype(np.float32) / np.float32(255.0)).astype(np.float32)


def test_scene_with_tga_bmp_jpeg_png_hdr_maps_and_an_hdr_sky(gpu_lib, oracle, workdir):
    from raylib_amd import binding
    from test_image_codecs import _turbo_fast_decode
    d = os.path.join(str(workdir), "assets"); os.makedirs(d, exist_ok=True)
    tex = scenes.pbr_textures()
    ext = {"pbr_albedo": ".tga", "pbr_normal": ".bmp", "pbr_rough": ".jpg", "pbr_metal": ".png", "pbr_emit": ".hdr"}
    obj, _ = scenes.pbr_maps(os.path.join(d, "assets.obj"), tess=2, ext=ext)         # writes only the .png
    expect = {}
    # TGA, 32 bits with alpha (top-left origin)
    Image.fromarray(tex["pbr_albedo"], "RGBA").save(os.path.join(d, "pbr_albedo.tga"), format="TGA")
    expect["pbr_albedo.tga"] = byte_image(np.asarray(Image.open(os.path.join(d, "pbr_albedo.tga")).convert("RGBA")))
    # BMP, 24 bits: alpha 1 after ConvertTo32Bits (render/image.cc:203-226)
    Image.fromarray(tex["pbr_normal"][..., :3], "RGB").save(os.path.join(d, "pbr_normal.bmp"), format="BMP")
    rgb = np.asarray(Image.open(os.path.join(d, "pbr_normal.bmp")).convert("RGB"))
    expect["pbr_normal.bmp"] = byte_image(np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], -1))
    # JPEG (lossy): an enlarged roughness map so that the 8x8 blocks hold something; decoded the FreeImage way (IFAST, no fancy up-sampling)
    big = np.kron(tex["pbr_rough"][..., :3], np.ones((4, 4, 1), np.uint8))
    bio = io.BytesIO(); Image.fromarray(big, "RGB").save(bio, "JPEG", quality=88, subsampling=2)
    open(os.path.join(d, "pbr_rough.jpg"), "wb").write(bio.getvalue())
    rgb = _turbo_fast_decode(bio.getvalue())
    expect["pbr_rough.jpg"] = byte_image(np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], -1))
    expect["pbr_metal.png"] = byte_image(tex["pbr_metal"])
    # Radiance HDR emissive map: values above 1
    emit = rgbe_pack(tex["pbr_emit"][..., :3].astype(np.float64) / 40.0)
    write_hdr(os.path.join(d, "pbr_emit.hdr"), emit)
    expect["pbr_emit.hdr"] = rgbe_unpack(emit)
    # the sky: an .hdr panorama on disk
    sky_rgbe = rgbe_pack(scenes.sky_panorama()[..., :3].astype(np.float64))
    write_hdr(os.path.join(d, "sky.hdr"), sky_rgbe)
    sky = rgbe_unpack(sky_rgbe)

    lib = gpu_lib
    cam = dict(origin=(0.1, 1.1, 6.0), look_at=(0, 0.95, -1), fov=50.0)
    ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], 72 / 48, sun=(2, 2, 2), sun_dir=(0.1, -0.3, -1.0))
    sky_h = lib.Raylib_LoadImage(os.path.join(d, "sky.hdr").encode())
    assert sky_h
    lib.Raylib_SetSkyPanorama(ses.scene, sky_h)
    # every map arrived: what the product decoded is what the independent decoders say
    assert lib.RaylibAMD_SceneNumTextures(ses.scene) == 5
    flat = objflat.load_obj(obj, oracle, texture_loader=lambda p: expect.get(os.path.basename(p)), sun_illuminance=(2, 2, 2), sun_direction=(0.1, -0.3, -1.0))
    assert len(flat.textures) == 5
    for i, t in enumerate(flat.textures):
        w, h = C.c_int32(), C.c_int32()
        lib.RaylibAMD_SceneTextureSize(ses.scene, i, C.byref(w), C.byref(h))
        assert (h.value, w.value) == t.shape[:2], i
        got = np.zeros_like(t)
        lib.RaylibAMD_SceneExportTexture(ses.scene, i, got.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(got, t), "texture %d differs from its independent decode" % i
    flat.textures.append(np.ascontiguousarray(sky, np.float32)); flat.sky_texture = len(flat.textures) - 1
    scene = oracle.scene_create(flat, 1)
    ocam = ffi.make_camera(cam["origin"], cam["look_at"], cam["fov"], 72 / 48)
    for spp, mode in ((1, 0), (4, 0), (1, 1), (1, 5)):
        got = ses.render(72, 48, spp, mode=mode)
        want = oracle.render(scene, ocam, ffi.make_settings(72, 48, spp, mode=mode), seed=1)
        differ = ~helpers.same(got[..., :3], want[..., :3]).all(-1)
        excused = 0
        for (py, px) in zip(*np.nonzero(differ)):
            oracle.render_region(scene, ocam, ffi.make_settings(72, 48, spp, mode=mode), int(px), int(py), 1, 1, seed=1)
            cn = oracle.counters(scene)
            assert cn["closest_hit_ties"] > 0 or cn["hits_outside_own_box"] > 0, (spp, mode, px, py, got[py, px], want[py, px])
            excused += 1
        print("spp %d mode %d: %d excused tie pixels of %d" % (spp, mode, excused, 72 * 48))
        assert excused <= 6, excused
    assert got[..., :3].max() > 0
    oracle.scene_destroy(scene)
    lib.Raylib_DestroyImage(sky_h)
    ses.close()


def test_more_textures_than_the_pool_kernel_keeps_in_lds(gpu_lib, oracle, workdir, monkeypatch):
    """64 materials with a map each (+ their 64 converted copies: 128 texture descriptors, RL_LDS_TEXTURES is 56): the pool kernel then reads descriptors from
    global memory (rl_render.hip TexTable), a scene of four maps from its LDS copy -- both must give the oracle's image, and the pool kernel's frame must be
    k_trace's bit for bit.  Every second map has holes (alpha 0 texels): the cut-out test inside the walk takes its texture from the per-triangle table."""
    from raylib_amd import binding
    d = os.path.join(str(workdir), "many_maps"); os.makedirs(d, exist_ok=True)
    rng = np.random.RandomState(11)
    N = 64
    mtl, objs, tex = [], [], {}
    for i in range(N):
        name = "m%02d" % i
        img = rng.randint(40, 255, (4, 4, 4)).astype(np.uint8)
        img[..., 3] = 255
        if i % 2:
            img[rng.randint(0, 4, 5), rng.randint(0, 4, 5), 3] = 0          # holes
        tex["map_%02d.png" % i] = img
        mtl.append("newmtl %s\nNs 10\nKd 0.8 0.8 0.8\nKs 0 0 0\nmap_Kd map_%02d.png\nillum 2\n" % (name, i))
        x0, y0 = -1.0 + 0.25 * (i % 8), 0.0 + 0.25 * (i // 8)
        z = -0.9 + 0.02 * i                                                  # a staircase of cards: rays that pass a hole meet the next card
        objs.append((name, name, [scenes._quad((x0, y0, z), (x0 + 0.25, y0, z), (x0 + 0.25, y0 + 0.25, z), (x0, y0 + 0.25, z))]))
    objs += [o for o in scenes.cornell_objects() if o[0] in ("floor", "backwall", "light")]
    obj, n = scenes.write_obj(os.path.join(d, "many.obj"), objs, scenes.CORNELL_MTL, tess=3, extra_mtl="\n" + "\n".join(mtl))
    scenes.write_textures(d, tex)
    assert n >= 256      # (the pool schedule's class)
    cam = dict(origin=(0.0, 1.0, 3.5), look_at=(0.0, 1.0, -1.0), fov=45.0)
    loader = lambda p: scenes.texture_as_float(tex[os.path.basename(p)]) if os.path.basename(p) in tex else None
    flat = objflat.load_obj(obj, oracle, texture_loader=loader)
    assert len(flat.textures) == N
    scene = oracle.scene_create(flat, 1)
    ocam = ffi.make_camera(cam["origin"], cam["look_at"], cam["fov"], 1.0)
    ses = binding.SceneSession(gpu_lib, obj, cam["origin"], cam["look_at"], cam["fov"], 1.0)
    assert gpu_lib.RaylibAMD_SceneNumTextures(ses.scene) == N
    frames = {}
    for pool in ("2", "0"):
        monkeypatch.setenv("RAYLIB_POOL", pool)
        frames[pool] = ses.render(96, 96, 8)
        st = ses.stats().as_dict()
        assert st["pathsPerWave"] == (128 if pool == "2" else 64) and st["texFetches"] > 0
    monkeypatch.delenv("RAYLIB_POOL")
    assert np.array_equal(helpers.bits(frames["2"]), helpers.bits(frames["0"])), "the pool kernel (descriptors from global memory) and k_trace differ"
    want = oracle.render(scene, ocam, ffi.make_settings(96, 96, 8), seed=1)
    differ = ~helpers.same(frames["2"][..., :3], want[..., :3]).all(-1)
    untied = 0
    for (py, px) in zip(*np.nonzero(differ)):
        oracle.render_region(scene, ocam, ffi.make_settings(96, 96, 8), int(px), int(py), 1, 1, seed=1)
        cn = oracle.counters(scene)
        if not (cn["closest_hit_ties"] > 0 or cn["hits_outside_own_box"] > 0):
            untied += 1
    assert untied == 0 and differ.sum() <= 40, (int(differ.sum()), untied)
    oracle.scene_destroy(scene)
    ses.close()
