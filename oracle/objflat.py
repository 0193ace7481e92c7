"""TEST INFRASTRUCTURE -- OBJ/MTL -> FlatScene for the CPU checkers.

The reference parses OBJ with tinyobjloader v2.0.0rc10 (Setup.ps1:39-40), which is
un-vendored and absent here, so this restates the published behaviour the
reference relies on at its call sites (loader/obj_loader.cc:91-99,133-234):
  * one shape per `o` / `g` statement (shapes without faces are dropped),
  * 1-based / negative relative indices in `f v/vt/vn`,
  * faces with >3 vertices are fan-triangulated (triangulate = true is the
    ObjReaderConfig default),
  * MTL defaults: Kd/Ks/Ke/Tf = 0, Ns = 1, Ni = 1, illum = 0, Pr = Pm = 0;
    Kd = 0.6 when map_Kd is given without Kd,
and then applies the reference's own rules: flat face normal when a vertex has no
normal (obj_loader.cc:199-203), UV = 0 when absent (:163-173), material-less faces
-> Lambertian(0.5) (:113,206-211), MTL -> material via oracle_material_from_mtl
(obj_loader.cc:354-397).  Parity for this stage is UNPINNED against the real
tinyobjloader; fixtures are pre-triangulated with explicit indices, where any
conforming parser agrees.
"""
import os
import numpy as np

from . import ffi


def parse_mtl(path):
    mats, cur = {}, None
    order = []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            k = tok[0]
            if k == "newmtl":
                cur = dict(name=tok[1], Kd=[0, 0, 0], Ks=[0, 0, 0], Ke=[0, 0, 0], Tf=[0, 0, 0], Ns=1.0, Ni=1.0,
                           illum=0, Pr=0.0, Pm=0.0, map_Kd="", map_Pr="", map_Pm="", map_Ke="", norm="", bump="", has_kd=False)
                mats[tok[1]] = cur
                order.append(tok[1])
            elif cur is None:
                continue
            elif k in ("Kd", "Ks", "Ke"):
                cur[k] = [float(x) for x in tok[1:4]]
                if k == "Kd":
                    cur["has_kd"] = True
            elif k in ("Tf", "Kt"):
                cur["Tf"] = [float(x) for x in tok[1:4]]
            elif k in ("Ns", "Ni", "Pr", "Pm"):
                cur[k] = float(tok[1])
            elif k == "illum":
                cur["illum"] = int(tok[1])
            elif k in ("map_Kd", "map_Pr", "map_Pm", "map_Ke", "norm"):
                cur[k] = tok[-1]
            elif k in ("map_bump", "map_Bump", "bump"):
                cur["bump"] = tok[-1]
    for m in mats.values():
        if m["map_Kd"] and not m["has_kd"]:
            m["Kd"] = [0.6, 0.6, 0.6]
    return [mats[n] for n in order]


def load_obj(path, oracle, texture_loader=None, sun_illuminance=(0, 0, 0), sun_direction=(0.0, -1.0, -0.5)):
    """Return a ffi.FlatScene for `path`.  `oracle` is ffi.load_oracle() (for the MTL rule).
    texture_loader(filename) -> float32 (H, W, 4) array, row 0 = top."""
    V, VT, VN = [], [], []
    shapes = []  # list of list of (tri vertex refs, material name)
    cur_shape, cur_mat = None, None
    mtl = []
    base = os.path.dirname(os.path.abspath(path))

    def fix(i, n):
        i = int(i)
        return i - 1 if i > 0 else n + i

    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            k = tok[0]
            if k == "v":
                V.append([float(x) for x in tok[1:4]])
            elif k == "vt":
                VT.append([float(tok[1]), float(tok[2]) if len(tok) > 2 else 0.0])
            elif k == "vn":
                VN.append([float(x) for x in tok[1:4]])
            elif k in ("o", "g"):
                cur_shape = []
                shapes.append(cur_shape)
            elif k == "usemtl":
                cur_mat = tok[1]
            elif k == "mtllib":
                mtl = parse_mtl(os.path.join(base, tok[1]))
            elif k == "f":
                if cur_shape is None:
                    cur_shape = []
                    shapes.append(cur_shape)
                refs = []
                for s in tok[1:]:
                    parts = s.split("/")
                    vi = fix(parts[0], len(V))
                    ti = fix(parts[1], len(VT)) if len(parts) > 1 and parts[1] else -1
                    ni = fix(parts[2], len(VN)) if len(parts) > 2 and parts[2] else -1
                    refs.append((vi, ti, ni))
                for j in range(1, len(refs) - 1):
                    cur_shape.append(((refs[0], refs[j], refs[j + 1]), cur_mat))
    shapes = [s for s in shapes if s]

    # materials: MTL order, then the fallback Lambertian(0.5)
    names = [m["name"] for m in mtl]
    textures, tex_index = [], {}

    def tex(fn):
        if not fn or texture_loader is None:
            return -1
        if fn not in tex_index:
            img = texture_loader(os.path.join(base, fn))
            if img is None:
                return -1
            tex_index[fn] = len(textures)
            textures.append(img)
        return tex_index[fn]

    mats = np.zeros(len(mtl) + 1, ffi.MAT_DTYPE)
    for i, m in enumerate(mtl):
        t_albedo = tex(m["map_Kd"])
        rec = oracle.material_from_mtl(m["Kd"], m["Ks"], m["Ke"], m["Tf"], m["Ns"], m["Ni"], m["illum"], m["Pr"], m["Pm"],
                                       bool(m["map_Kd"]))
        mats[i] = rec
        if rec["type"] == ffi.MAT_MICROFACET:
            mats[i]["texAlbedo"] = t_albedo
            n = tex(m["norm"])
            mats[i]["texNormal"] = n if n >= 0 else tex(m["bump"])
            mats[i]["texRoughness"] = tex(m["map_Pr"])
            mats[i]["texMetallic"] = tex(m["map_Pm"])
            mats[i]["texEmissive"] = tex(m["map_Ke"])
    fb = len(mtl)
    mats[fb]["type"] = ffi.MAT_LAMBERTIAN
    mats[fb]["albedo"] = (0.5, 0.5, 0.5)
    for k in ("texAlbedo", "texNormal", "texRoughness", "texMetallic", "texEmissive"):
        mats[fb][k] = -1

    Vf = np.asarray(V, np.float32).reshape(-1, 3)
    VTf = np.asarray(VT, np.float32).reshape(-1, 2)
    VNf = np.asarray(VN, np.float32).reshape(-1, 3)
    n_tri = sum(len(s) for s in shapes)
    tris = np.zeros(n_tri, ffi.TRI_DTYPE)
    k = 0
    for si, s in enumerate(shapes):
        for refs, mname in s:
            P = [Vf[r[0]] for r in refs]
            tris[k]["v0"], tris[k]["v1"], tris[k]["v2"] = P
            st = []
            for r in refs:
                st += list(VTf[r[1]]) if r[1] >= 0 else [0.0, 0.0]
            tris[k]["st"] = st
            if all(r[2] >= 0 for r in refs):
                tris[k]["n0"], tris[k]["n1"], tris[k]["n2"] = [VNf[r[2]] for r in refs]
            else:
                tris[k]["n0"] = tris[k]["n1"] = tris[k]["n2"] = _flat_normal(P)
            tris[k]["material"] = names.index(mname) if mname in names else fb
            tris[k]["shape"] = si
            k += 1
    return ffi.FlatScene(tris, mats, textures, num_shapes=len(shapes),
                         sun_illuminance=sun_illuminance, sun_direction=sun_direction)


def _flat_normal(P):
    """normalize(cross(p1-p0, p2-p0)) in float32 with the reference's operation order
    (core/vec3.h:124-130 cross, :48-53 Normalize = multiply by 1/length)."""
    f = np.float32
    a = [f(P[1][i] - P[0][i]) for i in range(3)]
    b = [f(P[2][i] - P[0][i]) for i in range(3)]
    c = [f(f(a[1] * b[2]) - f(a[2] * b[1])), f(-f(f(a[0] * b[2]) - f(a[2] * b[0]))), f(f(a[0] * b[1]) - f(a[1] * b[0]))]
    ln = np.sqrt(f(f(f(c[0] * c[0]) + f(c[1] * c[1])) + f(c[2] * c[2])))
    k = f(1.0) / f(ln)
    return [f(c[0] * k), f(c[1] * k), f(c[2] * k)]
