"""JPEG and TGA ingestion (csrc/rl_jpeg.cc).

The reference reads textures through FreeImage with flags 0, i.e. libjpeg with JDCT_IFAST and no fancy up-sampling.  FreeImage
is not in this image, so the codec is pinned against libjpeg-turbo driven the same way: Pillow's decoder in "draft" mode sets
dct_method = JDCT_FASTEST (= IFAST) and do_fancy_upsampling = FALSE.  Bit-exact 8-bit texels are required."""
import ctypes as C
import io
import os
import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image, features   # noqa: E402


def _load_through_abi(lib, path):
    h = lib.Raylib_LoadImage(path.encode())
    assert h, path
    cw, ch = C.c_uint32(), C.c_uint32()
    assert lib.RaylibAMD_ImageSize(h, C.byref(cw), C.byref(ch)) == 1
    w, hh = cw.value, ch.value
    buf = np.zeros((hh, w, 4), np.float32)
    lib.RaylibAMD_DumpImageRGBA(h, buf.ctypes.data_as(C.POINTER(C.c_float)))
    lib.Raylib_DestroyImage(h)
    return buf


def _turbo_fast_decode(data):
    im = Image.open(io.BytesIO(data))
    im.decoderconfig = (1, 1)          # (scale, draft): JpegDecode.c -> do_fancy_upsampling = FALSE, dct_method = JDCT_FASTEST
    im.load()
    return np.asarray(im.convert("RGB"), np.uint8)


def _picture(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 120 * np.sin(x / 7.0 + seed), 127 + 120 * np.cos(y / 5.0), (x * 3 + y * 2) % 256], -1)
    img += rng.normal(0, 18, img.shape)                       # texture, so every AC band is populated
    img[h // 3: h // 2, w // 4: w // 2] = (250, 5, 5)         # saturated flat patch with hard edges (clamping paths)
    return Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), "RGB")


CASES = [
    dict(size=(64, 48), subsampling=0, quality=90),
    dict(size=(61, 47), subsampling=1, quality=85),                       # 4:2:2, ragged
    dict(size=(97, 33), subsampling=2, quality=75),                       # 4:2:0, ragged
    dict(size=(40, 40), subsampling=2, quality=100),
    dict(size=(50, 31), subsampling=2, quality=30, optimize=True),
    dict(size=(72, 56), subsampling=2, quality=80, progressive=True),
    dict(size=(33, 65), subsampling=0, quality=92, progressive=True, optimize=True),
    dict(size=(80, 24), subsampling=1, quality=60, progressive=True),
    dict(size=(64, 64), subsampling=2, quality=85, restart_marker_blocks=3),
    dict(size=(45, 45), subsampling=0, quality=85, restart_marker_rows=1, progressive=True),
    dict(size=(31, 29), grey=True, quality=88),
    dict(size=(48, 40), grey=True, quality=70, progressive=True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join("%s%s" % (k[:4], v) for k, v in c.items()))
def test_jpeg_decoder_matches_libjpeg_fast_path(lib, workdir, case):
    if not features.check("jpg"):
        pytest.skip("Pillow without libjpeg")
    kw = dict(case)
    w, h = kw.pop("size")
    grey = kw.pop("grey", False)
    im = _picture(w, h, seed=w * 131 + h)
    if grey:
        im = im.convert("L")
    bio = io.BytesIO()
    im.save(bio, "JPEG", **kw)
    data = bio.getvalue()
    path = os.path.join(str(workdir), "t_%d_%d_%d.jpg" % (w, h, len(data)))
    open(path, "wb").write(data)
    want = _turbo_fast_decode(data)
    got = _load_through_abi(lib, path)
    assert got.shape == (h, w, 4)
    got8 = np.rint(got * 255.0).astype(np.int32)
    assert (got[..., 3] == 1.0).all()
    assert np.array_equal(got[..., :3], (want.astype(np.float32) / np.float32(255.0))), \
        "max texel difference %d" % np.abs(got8[..., :3] - want.astype(np.int32)).max()


def test_tga_decoder(lib, workdir):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (19, 23, 4), dtype=np.uint8)
    for mode, rle in (("RGBA", False), ("RGBA", True), ("RGB", True), ("RGB", False), ("L", False), ("L", True)):
        im = Image.fromarray(img, "RGBA").convert(mode)
        path = os.path.join(str(workdir), "t_%s_%d.tga" % (mode, rle))
        im.save(path, compression="tga_rle" if rle else None)
        want = np.asarray(Image.open(path).convert("RGBA"), np.uint8)
        got = _load_through_abi(lib, path)
        assert np.array_equal(got, want.astype(np.float32) / np.float32(255.0)), (mode, rle)


def test_unsupported_image_is_refused(lib, workdir):
    path = os.path.join(str(workdir), "junk.jpg")
    open(path, "wb").write(b"\xff\xd8\xff\xe0 not a jpeg at all")
    assert not lib.Raylib_LoadImage(path.encode())


def test_jpeg_writer_roundtrip(lib, workdir):
    """Raylib_WriteImageToDisk(..., Jpg): the console front-end saves its results as .jpg (reference src/main.cc:480-510).
    A lossy writer's bytes are not part of parity; the file must be a baseline JPEG that Pillow and the library's own
    decoder read back to within the quantisation error of quality 75 / 4:2:0."""
    for (w, h) in ((64, 48), (37, 29), (16, 16), (1, 1)):
        src = np.asarray(_picture(max(w, 8), max(h, 8), seed=w + h), np.uint8)[:h, :w]
        rgba = np.concatenate([src.astype(np.float32) / np.float32(255.0), np.ones((h, w, 1), np.float32)], -1)
        rgba = np.ascontiguousarray(rgba)
        img = lib.RaylibAMD_CreateImageFromData(w, h, rgba.ctypes.data_as(C.POINTER(C.c_float)))
        path = os.path.join(str(workdir), "w_%d_%d.jpg" % (w, h))
        assert lib.Raylib_WriteImageToDisk(img, path.encode(), 1) == 1
        lib.Raylib_DestroyImage(img)
        pil = Image.open(path)
        assert pil.format == "JPEG" and pil.size == (w, h)
        back = np.asarray(pil.convert("RGB"), np.float32)
        mine = _load_through_abi(lib, path)[..., :3] * 255.0
        # yardstick: libjpeg's own encoder at the same settings (quality 75, 4:2:0) on the same picture
        bio = io.BytesIO(); Image.fromarray(src).save(bio, "JPEG", quality=75, subsampling=2)
        ref = np.asarray(Image.open(io.BytesIO(bio.getvalue())).convert("RGB"), np.float32)
        ref_mse = float(((ref - src.astype(np.float32)) ** 2).mean())
        for got, slack in ((back, 1.1), (mine, 1.4)):     # the fast path (replicated chroma) reconstructs edges a little worse
            mse = float(((got - src.astype(np.float32)) ** 2).mean())
            assert mse <= slack * ref_mse + 2.0, (w, h, mse, ref_mse)
        assert os.path.getsize(path) <= 1.25 * len(bio.getvalue()) + 64
        fast = _turbo_fast_decode(open(path, "rb").read()).astype(np.float32)
        assert np.array_equal(np.rint(mine), fast)       # the library's decoder on the library's file = libjpeg's fast path, bit for bit


def test_corrupt_files_are_refused_or_decoded_without_harm(lib, tmp_path):
    """What the sanitizer fuzz (tools/fuzz_codecs.py) found, pinned: a header that claims an enormous picture must be refused before anything
    is allocated for it; a Huffman table naming an impossible magnitude category and coefficients far outside the 8-bit range must decode
    to *something* (as libjpeg does) without undefined behaviour.  Plus a short mutation run over every format."""
    import struct, subprocess, sys, zlib
    def load(name, data):
        p = tmp_path / name; p.write_bytes(data)
        h = lib.Raylib_LoadImage(str(p).encode())
        if h: lib.Raylib_DestroyImage(h)
        return bool(h)
    # PNG whose IHDR claims 60000 x 60000 with a 20-byte IDAT
    def chunk(t, b): return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 60000, 60000, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 16)) + chunk(b"IEND", b"")
    assert not load("huge.png", png)
    # BMP / TGA / HDR headers with absurd sizes
    bmp = bytearray(b"BM" + b"\0" * 52); struct.pack_into("<I", bmp, 10, 54); struct.pack_into("<ii", bmp, 18, 0x7fffffff, -0x80000000); struct.pack_into("<H", bmp, 28, 24)
    assert not load("huge.bmp", bytes(bmp))
    tga = bytearray(18); tga[2] = 10; struct.pack_into("<HH", tga, 12, 65535, 65535); tga[16] = 32
    assert not load("huge.tga", bytes(tga) + b"\xff\0\0\0\0")
    assert not load("huge.hdr", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2000000000 +X 2000000000\n" + b"\0" * 64)
    # a JPEG whose DC table maps the shortest code to "category 255", and one with every entropy-coded byte set to 0x7f
    im = _picture(40, 32, 3); b = io.BytesIO(); im.save(b, "JPEG", quality=85); good = bytearray(b.getvalue())
    i = good.find(b"\xff\xc4"); assert i > 0
    bad = bytearray(good); n = (bad[i + 2] << 8) | bad[i + 3]
    for k in range(i + 4 + 17, i + 2 + n): bad[k] = 255                      # every symbol of the first table becomes 255
    load("dc255.jpg", bytes(bad))                                            # either outcome is fine; it must return
    sos = good.find(b"\xff\xda"); bad = bytearray(good)
    for k in range(sos + 14, len(bad) - 2): bad[k] = 0x7f
    load("flood.jpg", bytes(bad))
    # mutation run (the sanitizer build runs the same script for tens of thousands of files)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_codecs.py"), "400", "5"], capture_output=True, text=True, timeout=300)
    assert "no crash" in out.stdout, out.stdout[-500:] + out.stderr[-1500:]
