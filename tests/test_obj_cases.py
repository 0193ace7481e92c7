"""OBJ / MTL edge cases against expectations WRITTEN OUT BY HAND (reference loader/obj_loader.cc:133-245,354-397).

The reference reads OBJ through tinyobjloader v2.0.0rc10, which is not vendored and not in this image, so no reference-held
vector exists for this stage.  Instead of "two parsers by the same author agree", the flat scene tests/golden/obj_cases/polygons.obj
must turn into is spelled out below, statement by statement, from tinyobjloader's documented semantics (quads cut along the shorter
diagonal, larger polygons ear-clipped, negative and forward indices, `usemtl` before `mtllib`, MTL colour defaults, texture options,
duplicate material names) followed by the reference's own MTL -> material rules.  Both the product's loader (through the C-ABI) and
the checker-side reader (oracle/objflat.py) are held to it."""
import os
import numpy as np
import pytest

import helpers
from helpers import ffi

CASES = os.path.join(helpers.GOLDEN, "obj_cases")

V = {1: (0, 0, 0), 2: (1, 0, 0), 3: (0, 1, 0), 4: (3, 1, 0), 5: (3, 0, 0), 6: (1, 1, 0), 7: (4, 0, 0), 8: (4, 4, 0), 9: (2, 1, 0), 10: (0, 4, 0),
     11: (10, 0, 0), 12: (12, 0, 0), 13: (13, 1, 0), 14: (12, 2, 0), 15: (10, 2, 0), 16: (9, 1, 0), 17: (0, 0, 5)}
VT1, VT2, NONE = (0.5, 0.25), (0.125, 0.75), (0.0, 0.0)
FALLBACK = 9
# (vertex numbers, uv per corner, material, shape)
TRIANGLES = [
    ((1, 2, 3), (NONE,) * 3, FALLBACK, 0),       # `usemtl one_component` before any mtllib: unknown name -> no material -> Lambertian(0.5)
    ((1, 2, 3), (NONE,) * 3, 0, 1),              # f 1 2 4 3: |v4-v1|^2 = 10 is not < |v3-v2|^2 = 2 -> [0,1,3] [1,2,3]
    ((2, 4, 3), (NONE,) * 3, 0, 1),
    ((1, 5, 6), (NONE,) * 3, 0, 1),              # f 1 5 6 3: |v6-v1|^2 = 2 < |v3-v5|^2 = 10 -> [0,1,2] [0,2,3]
    ((1, 6, 3), (NONE,) * 3, 0, 1),
    ((7, 8, 9), (NONE,) * 3, 2, 2),              # concave pentagon 1 7 8 9 10, reflex corner at v9: the candidate ear (1,7,8) contains v9 ->
    ((9, 10, 1), (NONE,) * 3, 2, 2),             # skipped; ear (7,8,9); (7,9,10) is reflex -> skipped; ear (9,10,1); what is left: (1,7,9).
    ((1, 7, 9), (NONE,) * 3, 2, 2),              # `usemtl dup` finds the FIRST of the two definitions
    ((11, 12, 13), (VT1,) * 3, 7, 3),            # convex hexagon through negative indices: the fan around its first corner
    ((11, 13, 14), (VT1,) * 3, 7, 3),
    ((11, 14, 15), (VT1,) * 3, 7, 3),
    ((11, 15, 16), (VT1,) * 3, 7, 3),
    ((1, 2, 3), (NONE,) * 3, 8, 4),              # v//vn
    ((1, 2, 3), (VT1,) * 3, 8, 4),               # v/vt: flat normal
    ((1, 2, 17), (VT1, VT1, VT2), 8, 4),         # v17 and vt2 are defined after this face
    ((1, 2, 3), (NONE,) * 3, FALLBACK, 4),       # usemtl with a name no library defines
]
M = ffi.MAT_MICROFACET
# type, then the fields that type reads
MATERIALS = [
    dict(type=M, albedo=(0.5, 0, 0), roughness=1.0, metallic=0.0, emissive=(0, 0, 0)),                       # `Kd 0.5`: missing components are 0; Ns 1, Ks 0 -> sqrt(2/2)
    dict(type=M, albedo=(0.25, 0.5, 0.75), roughness=float(np.sqrt(np.float32(2) / np.float32(10))), metallic=0.0, emissive=(0, 0, 0)),
    dict(type=M, albedo=(0.1, 0.1, 0.1), roughness=1.0, metallic=0.0, emissive=(0, 0, 0)),
    dict(type=M, albedo=(0.9, 0.9, 0.9), roughness=1.0, metallic=0.0, emissive=(0, 0, 0)),                   # second `dup`: kept in the list, never found by usemtl
    dict(type=ffi.MAT_DIELECTRIC, ior=1.33, transmission=(0.9, 0.8, 0.0)),                                   # illum 4, Kd 0, no map_Kd
    dict(type=M, albedo=(0.5, 0, 0), roughness=1.0, metallic=0.0, emissive=(0, 0, 0)),                       # illum 4 but Kd != 0: opaque
    dict(type=M, albedo=(0.6, 0.6, 0.6), roughness=1.0, metallic=0.0, emissive=(0, 0, 0), texAlbedo=-1),     # map_Kd without Kd -> 0.6; the name is "missing file.png"
    dict(type=ffi.MAT_MIRROR, albedo=(0.95, 0.5, 0.25)),                                                     # min(0.95, Kd)
    dict(type=M, albedo=(0.95, 0.2, 0.2), roughness=1.0, metallic=0.0, emissive=(1, 2, 0)),                  # clamps; `Ke 1 2`
    dict(type=ffi.MAT_LAMBERTIAN, albedo=(0.5, 0.5, 0.5)),
]


def check(tris, mats):
    assert len(tris) == len(TRIANGLES) and len(mats) == len(MATERIALS)
    f = np.float32
    for i, (vs, uvs, mat, shape) in enumerate(TRIANGLES):
        t = tris[i]
        for k, name in enumerate(("v0", "v1", "v2")):
            assert np.array_equal(t[name], f(V[vs[k]])), (i, name, t[name])
        assert np.array_equal(t["st"], f(np.concatenate(uvs))), (i, t["st"])
        for name in ("n0", "n1", "n2"):
            assert np.array_equal(t[name], f((0, 0, 1))), (i, name, t[name])      # every face lies in z = 0, counter-clockwise, or names vn 1
        assert (int(t["material"]), int(t["shape"])) == (mat, shape), (i, t["material"], t["shape"])
    for i, want in enumerate(MATERIALS):
        for k, v in want.items():
            got = mats[i][k]
            assert np.array_equal(np.asarray(got, np.float32), np.asarray(v, np.float32)) if k != "type" and not k.startswith("tex") else int(got) == v, (i, k, got, v)


def test_checker_side_reader(oracle):
    flat = helpers.objflat.load_obj(os.path.join(CASES, "polygons.obj"), oracle, texture_loader=lambda p: None)
    check(flat.triangles, flat.materials)


def test_product_loader(lib):
    from raylib_amd import binding
    ses = binding.SceneSession(lib, os.path.join(CASES, "polygons.obj"), (0, 1, 4), (0, 1, -1), 45.0, 1.0)
    tris, mats = ses.export_flat()
    ses.close()
    check(tris, mats)


def test_concave_polygon_is_covered_exactly_once():
    """Property of the ear clipper on random simple polygons (star-shaped around an inner point, so some corners are reflex): n - 2
    triangles, all with the polygon's orientation, areas summing to the polygon's area."""
    rng = np.random.RandomState(3)
    for _ in range(200):
        n = int(rng.randint(5, 12))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        if np.diff(np.concatenate([ang, [ang[0] + 2 * np.pi]])).max() > np.pi * 0.9:
            continue
        r = rng.uniform(0.3, 1.0, n)
        P = np.stack([r * np.cos(ang), r * np.sin(ang), np.zeros(n)], 1).astype(np.float32)
        tris = helpers.objflat.triangulate(P)
        assert len(tris) == n - 2

        def area(a, b, c):
            return 0.5 * ((b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]))
        total = sum(0.5 * (P[i][0] * P[(i + 1) % n][1] - P[i][1] * P[(i + 1) % n][0]) for i in range(n))
        parts = [area(P[a].astype(np.float64), P[b].astype(np.float64), P[c].astype(np.float64)) for a, b, c in tris]
        assert min(parts) > -1e-6 and abs(sum(parts) - total) < 1e-5
