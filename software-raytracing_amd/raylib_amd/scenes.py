"""Deterministic synthetic OBJ+MTL scene generators.

The reference's named scenes (Cornell Box, Breakfast Room, Sponza, San Miguel) are
downloaded by its Setup.ps1:42-79 and are not available offline, so every
BASELINE.json config runs on a synthetic stand-in of matching triangle count that
goes through the same OBJ path (Raylib_LoadOBJModel).  Cameras / suns follow the
reference's scenes.json and src/main.cc:64-155.

All coordinates are float32 and are printed with 9 significant digits, so any
conforming OBJ parser recovers the same float32 values.
"""
import math
import os
import numpy as np

WHITE, RED, GREEN, LIGHT, MIRROR, GLASS, CUTOUT = "white", "red", "green", "light", "mirror", "glass", "cutout"

CORNELL_MTL = """# synthetic Cornell box materials (Ns 10, Ks 0 => roughness sqrt(2/(10*0+2)) = 1)
newmtl white
Ns 10
Ka 0 0 0
Kd 0.725 0.71 0.68
Ks 0 0 0
illum 2

newmtl red
Ns 10
Kd 0.63 0.065 0.05
Ks 0 0 0
illum 2

newmtl green
Ns 10
Kd 0.14 0.45 0.091
Ks 0 0 0
illum 2

newmtl light
Ns 10
Kd 0.78 0.78 0.78
Ks 0 0 0
Ke 17 12 4
illum 2

newmtl mirror
Ns 10
Kd 0.9 0.9 0.9
Ks 0 0 0
illum 3

newmtl glass
Ns 10
Kd 0 0 0
Ks 0 0 0
Tf 0.95 0.97 0.95
Ni 1.5
illum 4
"""


def _f(x):
    return "%.9g" % float(np.float32(x))


def _quad(p0, p1, p2, p3):
    return np.array([p0, p1, p2, p3], np.float32)


def _box(center, size, yaw_deg):
    """6 outward-facing quads of a box rotated about +y."""
    cx, cy, cz = center
    hx, hy, hz = size[0] / 2, size[1] / 2, size[2] / 2
    c, s = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))

    def P(x, y, z):
        return (cx + c * x + s * z, cy + y, cz - s * x + c * z)

    quads = [
        _quad(P(-hx, hy, -hz), P(-hx, hy, hz), P(hx, hy, hz), P(hx, hy, -hz)),      # top
        _quad(P(-hx, -hy, -hz), P(hx, -hy, -hz), P(hx, -hy, hz), P(-hx, -hy, hz)),  # bottom
        _quad(P(-hx, -hy, hz), P(hx, -hy, hz), P(hx, hy, hz), P(-hx, hy, hz)),      # front (+z)
        _quad(P(hx, -hy, -hz), P(-hx, -hy, -hz), P(-hx, hy, -hz), P(hx, hy, -hz)),  # back
        _quad(P(hx, -hy, hz), P(hx, -hy, -hz), P(hx, hy, -hz), P(hx, hy, hz)),      # right
        _quad(P(-hx, -hy, -hz), P(-hx, -hy, hz), P(-hx, hy, hz), P(-hx, hy, -hz)),  # left
    ]
    return quads


def cornell_objects(tall_material=MIRROR, short_material=WHITE):
    """[(shape name, material, [quads])] -- 18 quads = 36 triangles."""
    return [
        ("floor", WHITE, [_quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1))]),
        ("ceiling", WHITE, [_quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1))]),
        ("backwall", WHITE, [_quad((-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1))]),
        ("leftwall", RED, [_quad((-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1))]),
        ("rightwall", GREEN, [_quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1))]),
        ("light", LIGHT, [_quad((-0.24, 1.98, -0.22), (0.23, 1.98, -0.22), (0.23, 1.98, 0.16), (-0.24, 1.98, 0.16))]),
        # the boxes float 1/1024 above the floor: coplanar duplicate surfaces would make the closest hit a tie,
        # which the reference resolves by the shape of its randomly built BVH (reference geom/bvh.cc:43,92)
        ("shortbox", short_material, _box((0.33, 0.3 + 1.0 / 1024, 0.35), (0.6, 0.6, 0.6), -17.0)),
        ("tallbox", tall_material, _box((-0.33, 0.6 + 1.0 / 1024, -0.3), (0.6, 1.2, 0.6), 17.0)),
    ]


def _tessellate(quad, k):
    """Split a quad into k*k cells -> (2*k*k, 3, 3) float64 positions and (2*k*k, 3, 2) float64 UVs, cell by cell (row j, column i),
    two triangles per cell: (a, b, c), (a, c, d)."""
    p0, p1, p2, p3 = [quad[i].astype(np.float64) for i in range(4)]
    j, i = np.mgrid[0:k, 0:k]
    u0, u1, v0, v1 = (i / k).ravel(), ((i + 1) / k).ravel(), (j / k).ravel(), ((j + 1) / k).ravel()

    def P(u, v):
        u, v = u[:, None], v[:, None]
        return ((1 - u) * (1 - v)) * p0 + (u * (1 - v)) * p1 + (u * v) * p2 + ((1 - u) * v) * p3

    a, b, c, d = P(u0, v0), P(u1, v0), P(u1, v1), P(u0, v1)
    tri = np.empty((k * k, 2, 3, 3), np.float64)
    tri[:, 0, 0], tri[:, 0, 1], tri[:, 0, 2] = a, b, c
    tri[:, 1, 0], tri[:, 1, 1], tri[:, 1, 2] = a, c, d
    uv = np.empty((k * k, 2, 3, 2), np.float64)
    uv[:, 0, 0] = np.stack([u0, v0], 1); uv[:, 0, 1] = np.stack([u1, v0], 1); uv[:, 0, 2] = np.stack([u1, v1], 1)
    uv[:, 1, 0] = np.stack([u0, v0], 1); uv[:, 1, 1] = np.stack([u1, v1], 1); uv[:, 1, 2] = np.stack([u0, v1], 1)
    return tri.reshape(-1, 3, 3), uv.reshape(-1, 3, 2)


def build_arrays(objects, tess=1, displace_fraction=0.0, displace_seed=7, room=((-1, 1), (0, 2), (-1, 1))):
    """The scene as arrays, in file order: positions (N, 3, 3) float32, UVs (N, 3, 2) float32, flat normals (N, 3) float32 and the
    object number of every triangle (N,), plus [(name, material)] per object.  These ARE the numbers write_obj prints (with 9
    significant digits, which a float32 survives), so a flat scene built from them is what any conforming reader makes of the file."""
    tris, uvs, owner, meta = [], [], [], []
    for oi, (name, material, quads) in enumerate(objects):
        meta.append((name, material))
        for quad in quads:
            t, uv = _tessellate(quad, tess)
            tris.append(t); uvs.append(uv); owner.append(np.full(len(t), oi, np.int32))
    tri = np.concatenate(tris).astype(np.float32)
    uv = np.concatenate(uvs).astype(np.float32)
    owner = np.concatenate(owner)
    n = len(tri)
    if displace_fraction > 0:
        # One stream of uniform doubles, consumed triangle by triangle: one draw decides whether the triangle moves, a moved one
        # takes three more for its new place inside the room (keeps its size and facing).
        rng = np.random.RandomState(displace_seed)
        R = rng.random_sample(4 * n)
        flags = (R < displace_fraction).tobytes()      # one byte per draw: "a triangle that starts here moves"
        at = np.empty(n, np.int64)
        pos = 0
        starts = []
        for k in range(n):
            starts.append(pos)
            pos += 4 if flags[pos] else 1
        at[:] = starts
        del starts
        moved = R[at] < displace_fraction
        m = np.nonzero(moved)[0]
        lo = np.array([r[0] + 0.05 for r in room]); hi = np.array([r[1] - 0.05 for r in room])
        target = (lo + (hi - lo) * np.stack([R[at[m] + 1], R[at[m] + 2], R[at[m] + 3]], 1)).astype(np.float32)
        centre = tri[m].mean(axis=1)
        tri[m] = (tri[m] - centre[:, None, :] + target[:, None, :]).astype(np.float32)
    e1 = tri[:, 1].astype(np.float64) - tri[:, 0]
    e2 = tri[:, 2].astype(np.float64) - tri[:, 0]
    nrm = np.cross(e1, e2)
    length = np.sqrt(np.einsum("ij,ij->i", nrm, nrm))
    nrm = (nrm / np.maximum(length, 1e-30)[:, None]).astype(np.float32)
    was_moved = np.zeros(n, bool)
    if displace_fraction > 0:
        was_moved[m] = True
    return dict(tri=tri, uv=uv, normal=nrm, owner=owner, objects=meta, moved=was_moved)


def write_obj(path, objects, mtl_text, tess=1, with_normals=True, with_uvs=True,
              displace_fraction=0.0, displace_seed=7, room=((-1, 1), (0, 2), (-1, 1)), extra_mtl=""):
    """Write <path>.obj and <path>.mtl; return the .obj path and the triangle count."""
    base = os.path.splitext(path)[0]
    A = build_arrays(objects, tess, displace_fraction, displace_seed, room)
    write_obj_text(base + ".obj", A, os.path.basename(base) + ".mtl", with_normals, with_uvs)
    with open(base + ".mtl", "w") as f:
        f.write(mtl_text + extra_mtl)
    return base + ".obj", len(A["tri"])


def write_obj_text(obj_path, A, mtl_name, with_normals=True, with_uvs=True):
    """The OBJ text of build_arrays' scene: per triangle three `v`, (three `vt`), (one `vn`) and one `f`; `o` / `usemtl` per object."""
    tri, uv, nrm, owner = A["tri"].astype(np.float64), A["uv"].astype(np.float64), A["normal"].astype(np.float64), A["owner"]
    out = ["# synthetic scene (raylib_amd.scenes)\nmtllib %s\n" % mtl_name]
    last = -1
    tl, ul, nl, ol = tri.tolist(), uv.tolist(), nrm.tolist(), owner.tolist()
    for k in range(len(tl)):
        if ol[k] != last:
            last = ol[k]
            out.append("o %s\nusemtl %s\n" % A["objects"][last])
        vi = 3 * k + 1
        for p in tl[k]:
            out.append("v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        if with_uvs:
            for t in ul[k]:
                out.append("vt %.9g %.9g\n" % (t[0], t[1]))
        if with_normals:
            out.append("vn %.9g %.9g %.9g\n" % (nl[k][0], nl[k][1], nl[k][2]))
        idx = []
        for c in range(3):
            w = str(vi + c)
            if with_uvs or with_normals:
                w += "/" + (str(vi + c) if with_uvs else "")
            if with_normals:
                w += "/" + str(k + 1)
            idx.append(w)
        out.append("f %s\n" % " ".join(idx))
    with open(obj_path, "w") as f:
        f.write("".join(out))


# ---------------------------------------------------------------------------------
# BASELINE.json configs

def cornell(path, tess=1, displace_fraction=0.0, tall_material=MIRROR, short_material=WHITE, **kw):
    """C1/C2: 36-triangle Cornell box (tess=1).  C3: tess=91 + displace 0.2 -> 298 116 triangles."""
    return write_obj(path, cornell_objects(tall_material, short_material), CORNELL_MTL, tess=tess,
                     displace_fraction=displace_fraction, **kw)


def colonnade_objects(n_columns=24, segments=24):
    """C4 'Sponza-size' stand-in: a long hall with two rows of faceted columns."""
    objs = [
        ("floor", WHITE, [_quad((-14, 0, 6), (14, 0, 6), (14, 0, -6), (-14, 0, -6))]),
        ("wall_n", WHITE, [_quad((-14, 0, -6), (14, 0, -6), (14, 8, -6), (-14, 8, -6))]),
        ("wall_s", RED, [_quad((14, 0, 6), (-14, 0, 6), (-14, 8, 6), (14, 8, 6))]),
        ("wall_w", GREEN, [_quad((-14, 0, 6), (-14, 0, -6), (-14, 8, -6), (-14, 8, 6))]),
        ("wall_e", WHITE, [_quad((14, 0, -6), (14, 0, 6), (14, 8, 6), (14, 8, -6))]),
    ]
    for ci in range(n_columns):
        row = ci % 2
        x = -12.0 + (ci // 2) * (24.0 / max(1, n_columns // 2 - 1))
        z = -3.0 if row == 0 else 3.0
        quads = []
        r = 0.45
        for s in range(segments):
            a0, a1 = 2 * math.pi * s / segments, 2 * math.pi * (s + 1) / segments
            x0, z0, x1, z1 = x + r * math.cos(a0), z + r * math.sin(a0), x + r * math.cos(a1), z + r * math.sin(a1)
            quads.append(_quad((x1, 0, z1), (x0, 0, z0), (x0, 6, z0), (x1, 6, z1)))
        objs.append(("column%02d" % ci, MIRROR if ci % 7 == 3 else WHITE, quads))
    return objs


def colonnade(path, tess=6, **kw):
    """C4: 24 columns x 24 facets + 5 walls = 581 quads; tess=6 -> 41 832, tess=12 -> 167 328 triangles."""
    return write_obj(path, colonnade_objects(), CORNELL_MTL, tess=tess, **kw)


# ---------------------------------------------------------------------------------
# The BASELINE-size rooms WITH textures and alpha cut-outs ("tessellated room + instanced foliage cards with an alpha-cut-out
# texture"; Breakfast Room, Sponza and San Miguel all use map_Kd): every wall of the tessellated room carries an albedo map, and the fifth of
# the triangles that build_arrays displaces into the room are FOLIAGE CARDS -- one object with a cut-out albedo map (alpha 0 / 255, about half
# of it holes), UVs spread so that a card spans a few leaves.  A material with an albedo map is alpha-tested inside Triangle::Hit for every
# candidate (reference geom/triangle.cc:54, render/material.cc:397-404, render/texture.cc:30-53): walls pass (alpha 1), cards cut rays' paths
# open -- any-hit work in the traversal of a deep tree, texel traffic, divergence.

WALL_W, WALL_R, WALL_G, FOLIAGE = "white", "red", "green", "foliage"

TEXTURED_MTL = """# tessellated room with albedo maps; foliage cards with an alpha cut-out
newmtl white
Ns 10
Kd 0.725 0.71 0.68
Ks 0 0 0
map_Kd wall_white.png
illum 2

newmtl red
Ns 10
Kd 0.63 0.065 0.05
Ks 0 0 0
map_Kd wall_red.png
illum 2

newmtl green
Ns 10
Kd 0.14 0.45 0.091
Ks 0 0 0
map_Kd wall_green.png
illum 2

newmtl light
Ns 10
Kd 0.78 0.78 0.78
Ks 0 0 0
Ke 17 12 4
illum 2

newmtl mirror
Ns 10
Kd 0.9 0.9 0.9
Ks 0 0 0
illum 3

newmtl foliage
Ns 10
Kd 0.5 0.6 0.3
Ks 0 0 0
map_Kd foliage.png
illum 2
"""
FOLIAGE_UV_SCALE = 24.0


def wall_texture(tint, size=128):
    """Opaque RGBA8 'plaster and bricks': a brick bond in two tones of `tint` (0..255 per channel) with darker joints; alpha 255 everywhere."""
    yy, xx = np.mgrid[0:size, 0:size]
    row = yy // 16
    col = (xx + 16 * (row % 2)) // 32
    joint = ((yy % 16) < 2) | (((xx + 16 * (row % 2)) % 32) < 2)
    tone = 200 + 17 * ((row * 7 + col * 13) % 4)
    img = np.zeros((size, size, 4), np.uint8)
    for c in range(3):
        v = (tint[c] * tone) // 255
        img[..., c] = np.where(joint, (v * 5) // 8, v)
    img[..., 3] = 255
    return img


def foliage_texture(size=64):
    """RGBA8 leaves: elliptic blobs of greens on a fully transparent ground (alpha 0 / 255 only; a little under half of the texels are leaf)."""
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    img = np.zeros((size, size, 4), np.uint8)
    rng = np.random.RandomState(11)
    for _ in range(30):
        cx, cy = rng.uniform(0, size, 2)
        a, b = rng.uniform(5, 11), rng.uniform(2.5, 5)
        th = rng.uniform(0, np.pi)
        g = (70 + rng.randint(0, 120), 110 + rng.randint(0, 130), 30 + rng.randint(0, 60))
        for ox in (-size, 0, size):            # leaves wrap around the edges: the sampler repeats the texture (render/texture.cc:38-41)
            for oy in (-size, 0, size):
                dx, dy = xx - cx - ox, yy - cy - oy
                u, v = dx * np.cos(th) + dy * np.sin(th), -dx * np.sin(th) + dy * np.cos(th)
                inside = (u / a) ** 2 + (v / b) ** 2 <= 1.0
                img[inside, 0], img[inside, 1], img[inside, 2], img[inside, 3] = g[0], g[1], g[2], 255
    return img


def textured_textures():
    """file name -> (H, W, 4) uint8, what `textured` writes next to the OBJ."""
    return {"wall_white.png": wall_texture((185, 181, 173)), "wall_red.png": wall_texture((161, 17, 13)), "wall_green.png": wall_texture((36, 115, 23)),
            "foliage.png": foliage_texture()}


def build_arrays_textured(tess=1, displace_fraction=0.2, displace_seed=7):
    """build_arrays of the Cornell room with the displaced triangles gathered into ONE extra object, the foliage cards (material `foliage`, UVs spread by
    FOLIAGE_UV_SCALE so that the sampler's repeat wrap lays several leaves over a card).  Same dictionary as build_arrays."""
    A = build_arrays(cornell_objects(), tess, displace_fraction, displace_seed)
    mv = A["moved"]
    keep = np.nonzero(~mv)[0]; cards = np.nonzero(mv)[0]
    order = np.concatenate([keep, cards])
    uv = A["uv"].copy()
    uv[cards] = (uv[cards].astype(np.float64) * FOLIAGE_UV_SCALE).astype(np.float32)
    owner = A["owner"].copy()
    objects = list(A["objects"])
    if len(cards):
        owner[cards] = len(objects)
        objects.append(("foliage", FOLIAGE))
    return dict(tri=A["tri"][order], uv=uv[order], normal=A["normal"][order], owner=owner[order], objects=objects, moved=mv[order])


def write_textures(directory, textures):
    for name, img in textures.items():
        write_png_rgba(os.path.join(directory, name), img)


def textured(path, tess=91, displace_fraction=0.2):
    """The configs[2]-sized room with albedo maps on its walls and a fifth of its triangles as alpha-cut-out foliage cards (tess = 91: 298 116 triangles,
    59 477 of them cards).  Writes <path>.obj, .mtl and the four PNG maps; returns the .obj path and the triangle count."""
    base = os.path.splitext(path)[0]
    A = build_arrays_textured(tess, displace_fraction)
    write_obj_text(base + ".obj", A, os.path.basename(base) + ".mtl")
    with open(base + ".mtl", "w") as f:
        f.write(TEXTURED_MTL)
    write_textures(os.path.dirname(os.path.abspath(path)), textured_textures())
    return base + ".obj", len(A["tri"])


# Cameras / suns per config (reference scenes.json, src/main.cc:64-155)
CONFIG_CAMERAS = {
    "cornell":   dict(origin=(0.0, 1.0, 4.0), look_at=(0.0, 1.0, -1.0), fov=45.0, sun=(0.0, 0.0, 0.0), sun_dir=(0.0, -1.0, -0.5)),
    "breakfast": dict(origin=(0.0, 1.0, 5.0), look_at=(0.0, 1.0, -1.0), fov=60.0, sun=(20.0, 20.0, 20.0), sun_dir=(-1.0, -1.0, 0.0)),
    # the camera INSIDE the tessellated room (what Breakfast Room / Sponza / San Miguel are: interiors, reference Setup.ps1:42-79, scenes.json): the open front is
    # behind it, every pixel looks at geometry, no cell of the frame can be dropped, paths leave through the holes the displaced triangles left and through the front
    "breakfast_interior": dict(origin=(0.15, 1.1, 0.9), look_at=(-0.1, 0.9, -1.0), fov=60.0, sun=(20.0, 20.0, 20.0), sun_dir=(-1.0, -1.0, 0.0)),
    "sponza":    dict(origin=(10.0, 2.0, 0.0), look_at=(0.0, 3.0, 0.0), fov=60.0, sun=(20.0, 20.0, 20.0), sun_dir=(0.0, -1.0, -0.5)),
}


# ---------------------------------------------------------------------------------
# Textured / cut-out test scene (exercises map_Kd, sRGB decode, alpha test inside
# traversal and the sky panorama; the other map slots: pbr_maps below)

CUTOUT_MTL = """
newmtl cutout
Ns 10
Kd 0.8 0.8 0.8
Ks 0 0 0
map_Kd leaf.png
illum 2
"""


def leaf_texture(size=16):
    """RGBA8 checkerboard: opaque green / transparent cells plus a gradient, (H, W, 4) uint8."""
    img = np.zeros((size, size, 4), np.uint8)
    for y in range(size):
        for x in range(size):
            on = ((x // 2) + (y // 2)) % 2 == 0
            img[y, x] = (40 + 10 * x, 120 + 8 * y, 30 + 3 * (x + y), 255 if on else 0)
    return img


def write_png_rgba(path, img):
    """Minimal PNG writer (zlib), so fixtures do not depend on PIL."""
    import struct
    import zlib
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].astype(np.uint8).tobytes() for y in range(h))

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def texture_as_float(img_u8):
    """byte/255 in float32, as reference render/image.h:37-43."""
    return (img_u8.astype(np.float32) / np.float32(255.0)).astype(np.float32)


def cutout(path, tess=1):
    """Cornell room with a cut-out card (alpha-tested map_Kd) in front of the boxes."""
    objs = cornell_objects()
    objs.append(("card", CUTOUT, [_quad((-0.6, 0.2, 0.7), (0.6, 0.2, 0.7), (0.6, 1.4, 0.7), (-0.6, 1.4, 0.7))]))
    obj, n = write_obj(path, objs, CORNELL_MTL, tess=tess, extra_mtl=CUTOUT_MTL)
    write_png_rgba(os.path.join(os.path.dirname(os.path.abspath(path)), "leaf.png"), leaf_texture())
    return obj, n


# ---------------------------------------------------------------------------------
# PBR test scene: every MicrofacetMaterial input the MTL path can set (reference
# loader/obj_loader.cc:372-395, render/material.cc:290-431) away from the Cornell
# defaults -- roughness in (0,1) from `Pr` and from the Phong pair `Ns`/`Ks`, metallic
# > 0, a normal map through `norm` and through the `map_bump` fallback, `map_Pr`,
# `map_Pm`, `map_Ke` (the (U,U) / .b quirk of Emitted), an albedo map with partial
# alpha (GetAlbedo multiplies by it) and a surface with roughness < 0.1 (IsMirrorLike
# in the albedo AOV).

FLOOR_PR, PHONG, MAPPED, BUMPED, GLOSSY, BRUSHED, EMITMAP = "floor_pr", "phong", "mapped", "bumped", "glossy", "brushed", "emitmap"

PBR_MTL = """
newmtl floor_pr
Kd 0.6 0.5 0.4
Pr 0.35
Pm 0.7

newmtl phong
Kd 0.7 0.7 0.7
Ns 96
Ks 0.5 0.5 0.5
illum 2

newmtl mapped
Kd 0.8 0.8 0.8
Ns 40
Ks 0.3 0.2 0.1
map_Kd pbr_albedo%(ext_albedo)s
norm pbr_normal%(ext_normal)s
map_Pr pbr_rough%(ext_rough)s
map_Pm pbr_metal%(ext_metal)s

newmtl bumped
Kd 0.3 0.5 0.7
Pr 0.6
Pm 0.2
map_bump pbr_normal%(ext_normal)s

newmtl glossy
Kd 0.9 0.6 0.3
Pr 0.05
Pm 1

newmtl brushed
Kd 0.95 0.93 0.88
Pr 0.25
Pm 1.0
Ke 0.02 0.01 0.0

newmtl emitmap
Kd 0.2 0.2 0.2
Ke 1 1 1
map_Ke pbr_emit%(ext_emit)s
"""


def pbr_textures():
    """name -> (H, W, 4) uint8.  Sizes differ per map and are not powers of two."""
    out = {}
    h, w = 12, 20
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.zeros((h, w, 4), np.uint8)
    a[..., 0] = 60 + 9 * xx; a[..., 1] = 200 - 11 * yy; a[..., 2] = 90 + 5 * ((xx + yy) % 7)
    a[..., 3] = np.where((xx + 2 * yy) % 5 == 0, 180, 255)          # partial alpha, all above the 0.5 cut-out
    out["pbr_albedo"] = a
    h, w = 16, 16
    yy, xx = np.mgrid[0:h, 0:w]
    n = np.zeros((h, w, 4), np.uint8)
    n[..., 0] = np.clip(128 + 70 * np.sin(xx * 0.9), 0, 255); n[..., 1] = np.clip(128 + 60 * np.cos(yy * 0.7), 0, 255)
    n[..., 2] = 230; n[..., 3] = 255
    out["pbr_normal"] = n
    h, w = 9, 13
    yy, xx = np.mgrid[0:h, 0:w]
    r = np.zeros((h, w, 4), np.uint8)
    r[..., 0] = 8 + (17 * xx + 23 * yy) % 230                       # 0.03 .. 0.93: both sides of IsMirrorLike's 0.1
    r[..., 1] = 255 - r[..., 0]; r[..., 2] = 40; r[..., 3] = 255    # .g / .b must not be what is read
    out["pbr_rough"] = r
    h, w = 7, 5
    yy, xx = np.mgrid[0:h, 0:w]
    m = np.zeros((h, w, 4), np.uint8)
    m[..., 0] = (xx * 63) % 256; m[..., 1] = 10; m[..., 2] = 250; m[..., 3] = 255
    m[0, 0, 0] = 255
    out["pbr_metal"] = m
    h, w = 10, 10
    yy, xx = np.mgrid[0:h, 0:w]
    e = np.zeros((h, w, 4), np.uint8)
    e[..., 0] = 255; e[..., 1] = 12 * yy; e[..., 2] = 25 * ((xx + yy) % 10); e[..., 3] = 255   # only .b reaches the image
    out["pbr_emit"] = e
    return out


def random_pbr_mtl(rng):
    """PBR_MTL's material names with random constants and a random subset of the map statements (fuzzing)."""
    maps = (("map_Kd", "pbr_albedo"), ("norm", "pbr_normal"), ("map_bump", "pbr_normal"), ("map_Pr", "pbr_rough"),
            ("map_Pm", "pbr_metal"), ("map_Ke", "pbr_emit"))
    out = []
    for name in (FLOOR_PR, PHONG, MAPPED, BUMPED, GLOSSY, BRUSHED, EMITMAP):
        out.append("newmtl %s" % name)
        out.append("Kd %s %s %s" % tuple(_f(x) for x in rng.uniform(0.05, 1.0, 3)))
        if rng.rand() < 0.5:
            out.append("Pr %s" % _f(rng.choice([0.02, 0.08, 0.1, 0.3, 0.7, 1.0, 1.5]) if rng.rand() < 0.5 else rng.uniform(0.01, 1.0)))
        else:
            out.append("Ns %s" % _f(rng.choice([1.0, 10.0, 96.0, 400.0, 1000.0])))
            out.append("Ks %s %s %s" % tuple(_f(x) for x in rng.uniform(0.0, 1.0, 3)))
        if rng.rand() < 0.6:
            out.append("Pm %s" % _f(rng.choice([0.0, 0.5, 1.0, 1.7]) if rng.rand() < 0.5 else rng.uniform(0.0, 1.0)))
        if rng.rand() < 0.3:
            out.append("Ke %s %s %s" % tuple(_f(x) for x in rng.uniform(0.0, 3.0, 3)))
        for stmt, tex in maps:
            if rng.rand() < 0.35:
                out.append("%s %s.png" % (stmt, tex))
        out.append("")
    return "\n".join(out) + "\n"


def pbr_maps(path, tess=1, ext=None, mtl=None):
    """Cornell-shaped room whose surfaces carry the materials of PBR_MTL (or of `mtl`, same material names).  `ext` maps a
    texture name to the file extension its MTL statement uses (default .png for all); the caller writes non-PNG files itself."""
    ext = dict(ext or {})
    objs = [
        ("floor", FLOOR_PR, [_quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1))]),
        ("ceiling", WHITE, [_quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1))]),
        ("backwall", PHONG, [_quad((-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1))]),
        ("leftwall", MAPPED, [_quad((-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1))]),
        ("rightwall", BUMPED, [_quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1))]),
        ("light", LIGHT, [_quad((-0.24, 1.98, -0.22), (0.23, 1.98, -0.22), (0.23, 1.98, 0.16), (-0.24, 1.98, 0.16))]),
        ("sign", EMITMAP, [_quad((-0.5, 1.2, -0.97), (0.5, 1.2, -0.97), (0.5, 1.8, -0.97), (-0.5, 1.8, -0.97))]),
        ("shortbox", GLOSSY, _box((0.33, 0.3 + 1.0 / 1024, 0.35), (0.6, 0.6, 0.6), -17.0)),
        ("tallbox", BRUSHED, _box((-0.33, 0.6 + 1.0 / 1024, -0.3), (0.6, 1.2, 0.6), 17.0)),
        ("panel", MAPPED, [_quad((0.15, 0.05, 0.95), (0.85, 0.05, 0.8), (0.85, 0.75, 0.8), (0.15, 0.75, 0.95))]),
    ]
    names = {"ext_" + k[4:]: ext.get(k, ".png") for k in ("pbr_albedo", "pbr_normal", "pbr_rough", "pbr_metal", "pbr_emit")}
    obj, n = write_obj(path, objs, CORNELL_MTL, tess=tess, extra_mtl=(PBR_MTL % names) if mtl is None else mtl)
    d = os.path.dirname(os.path.abspath(path))
    for name, img in pbr_textures().items():
        if ext.get(name, ".png") == ".png":
            write_png_rgba(os.path.join(d, name + ".png"), img)
    return obj, n


# ---------------------------------------------------------------------------------
# Hall of mirrors: the Cornell room closed by a front wall, every wall and both boxes `illum 3` (Mirror, reference
# loader/obj_loader.cc:365-367: albedo = min(0.95, Kd)) -- a Mirror always scatters with pdf 1 (render/material.h:149-163), so from a
# camera inside nearly every path lives until the maxPathLength cut (renderer.cc:120-123) and its vertices all carry weight
# (0.95^200 = 3.5e-5 of a 17-unit light is far above a float's resolution): the test scene for long paths.

MIRROR_HALL_MTL = """newmtl white
Kd 0.95 0.95 0.95
illum 3

newmtl red
Kd 0.95 0.9 0.88
illum 3

newmtl green
Kd 0.9 0.95 0.89
illum 3

newmtl mirror
Kd 0.93 0.93 0.95
illum 3

newmtl light
Ns 10
Kd 0.78 0.78 0.78
Ks 0 0 0
Ke 17 12 4
illum 2
"""
MIRROR_HALL_CAMERA = dict(origin=(0.15, 1.1, 0.9), look_at=(-0.1, 0.9, -1.0), fov=60.0, sun=(0.0, 0.0, 0.0), sun_dir=(0.0, -1.0, -0.5))


def mirror_hall(path, tess=1):
    objs = cornell_objects()
    objs.append(("frontwall", WHITE, [_quad((1, 0, 1), (-1, 0, 1), (-1, 2, 1), (1, 2, 1))]))
    return write_obj(path, objs, MIRROR_HALL_MTL, tess=tess)


def soup(path, n_tris=10000, seed=3, extent=4.0, size=0.35):
    """Random triangle soup (no normals, no UVs) for closest-hit / BVH-vs-brute-force tests."""
    rng = np.random.RandomState(seed)
    base = os.path.splitext(path)[0]
    with open(base + ".obj", "w") as f:
        f.write("o soup\n")
        for i in range(n_tris):
            c = rng.uniform(-extent, extent, 3)
            p = (c + rng.uniform(-size, size, (3, 3))).astype(np.float32)
            for q in p:
                f.write("v %s %s %s\n" % (_f(q[0]), _f(q[1]), _f(q[2])))
            f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    return base + ".obj", n_tris


def sky_panorama(w=64, h=32):
    """Float RGBA equirectangular test panorama (smooth gradient + a bright spot)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 4), np.float32)
    img[..., 0] = 0.2 + 0.6 * xx / w
    img[..., 1] = 0.3 + 0.5 * yy / h
    img[..., 2] = 0.9 - 0.4 * xx / w
    img[..., 3] = 1.0
    img[h // 4, w // 3, :3] = (8.0, 7.0, 5.0)
    return img
