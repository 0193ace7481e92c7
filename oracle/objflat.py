"""TEST INFRASTRUCTURE -- OBJ/MTL -> FlatScene for the CPU checkers.

The reference parses OBJ with tinyobjloader v2.0.0rc10 (Setup.ps1:39-40), which is
un-vendored and absent here, so this restates the published behaviour the
reference relies on at its call sites (loader/obj_loader.cc:91-99,133-234):
  * one shape per `o` / `g` statement (shapes without faces are dropped),
  * 1-based / negative relative indices in `f v/vt/vn`; index 0, a word that is not a
    number, or a relative index before the first element fail the whole load; a
    positive index may name an element defined further down the file,
  * faces with >3 vertices are triangulated (triangulate = true is the ObjReaderConfig
    default): quads along the shorter diagonal, larger polygons by ear clipping,
  * MTL: '#' comments only at line start, missing colour components = 0, texture name =
    rest of the line after the options, first definition of a material name wins,
    `mtllib a b` reads the first file that opens; defaults Kd/Ks/Ke/Tf = 0, Ns = 1,
    Ni = 1, illum = 0, Pr = Pm = 0; Kd = 0.6 when map_Kd is given without Kd,
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


_BLANKS = " \t"


def _words(text):
    return [w for w in text.replace("\t", " ").split(" ") if w]


def _real(words, i):
    """tinyobjloader parseReal: word i as a number, 0 when absent or not a number."""
    if i >= len(words):
        return 0.0
    try:
        return float(words[i])
    except ValueError:
        return 0.0


_OPT_ARGS = {"-blendu": 1, "-blendv": 1, "-clamp": 1, "-boost": 1, "-bm": 1, "-type": 1, "-texres": 1, "-imfchan": 1, "-colorspace": 1,
             "-mm": 2, "-o": 3, "-s": 3, "-t": 3}


def _texture_name(rest):
    """Options first (their arguments are swallowed word by word), then the REST OF THE LINE is the name."""
    while True:
        rest = rest.lstrip(_BLANKS)
        if not rest:
            return ""
        w = _words(rest)[0]
        if w not in _OPT_ARGS:
            return rest
        rest = rest[len(w):]
        for _ in range(_OPT_ARGS[w]):
            rest = rest.lstrip(_BLANKS)
            ws = _words(rest)
            if ws:
                rest = rest[len(ws[0]):]


def parse_mtl(path):
    """tinyobjloader LoadMtl as far as the reference's material rule reads it.  Returns None when the file cannot be opened."""
    if not os.path.isfile(path):
        return None
    order = []
    scratch = {}

    def fresh(name):
        return dict(name=name, Kd=[0, 0, 0], Ks=[0, 0, 0], Ke=[0, 0, 0], Tf=[0, 0, 0], Ns=1.0, Ni=1.0,
                    illum=0, Pr=0.0, Pm=0.0, map_Kd="", map_Pr="", map_Pm="", map_Ke="", norm="", bump="", has_kd=False)

    cur = fresh("")
    with open(path, newline="") as f:
        for line in f.read().split("\n"):
            line = line.rstrip("\r").rstrip(_BLANKS).lstrip(_BLANKS)
            if not line or line[0] == "#":
                continue
            k = _words(line)[0]
            rest = line[len(k):]
            if k == "newmtl":
                name = rest.lstrip(_BLANKS)
                cur = fresh(name)
                if name:
                    order.append(cur)
                continue
            if not rest:
                continue
            w = _words(rest)
            if k in ("Kd", "Ks", "Ke"):
                cur[k] = [_real(w, 0), _real(w, 1), _real(w, 2)]
                if k == "Kd":
                    cur["has_kd"] = True
            elif k in ("Tf", "Kt"):
                cur["Tf"] = [_real(w, 0), _real(w, 1), _real(w, 2)]
            elif k in ("Ns", "Ni", "Pr", "Pm"):
                cur[k] = _real(w, 0)
            elif k == "illum":
                try:
                    cur["illum"] = int(w[0])
                except ValueError:
                    cur["illum"] = 0
            elif k == "map_Kd":
                cur["map_Kd"] = _texture_name(rest)
                if not cur["has_kd"]:
                    cur["Kd"] = [0.6, 0.6, 0.6]
            elif k in ("map_Pr", "map_Pm", "map_Ke", "norm"):
                cur[k] = _texture_name(rest)
            elif k in ("map_bump", "map_Bump", "bump"):
                cur["bump"] = _texture_name(rest)
    return order


class ObjLoadError(Exception):
    """What makes tinyobjloader's LoadObj return false (the reference then refuses the model, obj_loader.cc:91-95)."""


def _pnpoly3(vx, vy, tx, ty):
    c = False
    j = 2
    for i in range(3):
        if ((vy[i] > ty) != (vy[j] > ty)) and (tx < np.float32(np.float32(np.float32(vx[j] - vx[i]) * np.float32(ty - vy[i])) / np.float32(vy[j] - vy[i])) + vx[i]):
            c = not c
        j = i
    return c


def triangulate(P):
    """Corner triples tinyobjloader v2.0.0rc10 cuts a polygon into (float32 arithmetic as there): quads along the shorter
    diagonal, larger polygons by ear clipping in the projection plane of the first non-degenerate corner."""
    f = np.float32
    n = len(P)
    P = [[f(c) for c in p] for p in P]
    if n == 3:
        return [(0, 1, 2)]
    if n == 4:
        d02 = d13 = f(0)
        for a in range(3):
            e02, e13 = f(P[2][a] - P[0][a]), f(P[3][a] - P[1][a])
            d02 = f(d02 + f(e02 * e02)); d13 = f(d13 + f(e13 * e13))
        return [(0, 1, 2), (0, 2, 3)] if d02 < d13 else [(0, 1, 3), (1, 2, 3)]
    axes = [1, 2]
    for k in range(n):
        p0, p1, p2 = P[k % n], P[(k + 1) % n], P[(k + 2) % n]
        e0 = [f(p1[a] - p0[a]) for a in range(3)]; e1 = [f(p2[a] - p1[a]) for a in range(3)]
        cx = abs(f(f(e0[1] * e1[2]) - f(e0[2] * e1[1]))); cy = abs(f(f(e0[2] * e1[0]) - f(e0[0] * e1[2]))); cz = abs(f(f(e0[0] * e1[1]) - f(e0[1] * e1[0])))
        eps = f(1.1920929e-07)
        if cx > eps or cy > eps or cz > eps:
            if not (cx > cy and cx > cz):
                axes[0] = 0
                if cz > cx and cz > cy:
                    axes[1] = 1
            break
    area = f(0)
    for k in range(n):
        a, b = P[k], P[(k + 1) % n]
        area = f(area + f(f(f(a[axes[0]] * b[axes[1]]) - f(a[axes[1]] * b[axes[0]])) * f(0.5)))
    rest = list(range(n))
    out = []
    guess, remaining, previous = 0, n, n
    while len(rest) > 3 and remaining > 0:
        m = len(rest)
        if guess >= m:
            guess -= m
        if previous != m:
            previous, remaining = m, m
        else:
            remaining -= 1
        ind = [rest[(guess + k) % m] for k in range(3)]
        vx = [P[i][axes[0]] for i in ind]; vy = [P[i][axes[1]] for i in ind]
        e0x, e0y, e1x, e1y = f(vx[1] - vx[0]), f(vy[1] - vy[0]), f(vx[2] - vx[1]), f(vy[2] - vy[1])
        cross = f(f(e0x * e1y) - f(e0y * e1x))
        if f(cross * area) < 0:
            guess += 1
            continue
        if any(_pnpoly3(vx, vy, P[rest[(guess + o) % m]][axes[0]], P[rest[(guess + o) % m]][axes[1]]) for o in range(3, m)):
            guess += 1
            continue
        out.append(tuple(ind))
        del rest[(guess + 1) % m]
    for k in range(1, len(rest) - 1):
        out.append((rest[0], rest[k], rest[k + 1]))
    return out


def load_obj(path, oracle, texture_loader=None, sun_illuminance=(0, 0, 0), sun_direction=(0.0, -1.0, -0.5)):
    """Return a ffi.FlatScene for `path`.  `oracle` is ffi.load_oracle() (for the MTL rule).
    texture_loader(filename) -> float32 (H, W, 4) array, row 0 = top.  Raises ObjLoadError where tinyobjloader fails."""
    V, VT, VN = [], [], []
    shapes = []  # list of list of (polygon corner refs, material name)
    cur_shape, cur_mat = None, None
    mtl, mtl_index = [], {}
    base = os.path.dirname(os.path.abspath(path)) if os.path.dirname(path) else ""

    def fix(word, n):
        try:
            i = int(word)
        except ValueError:
            i = 0                                    # atoi
        if i == 0 or (i < 0 and n + i < 0):
            raise ObjLoadError("face index %r" % word)
        return i - 1 if i > 0 else n + i

    with open(path, newline="") as f:
        for line in f.read().split("\n"):
            line = line.rstrip("\r")
            tok = _words(line)
            if not tok or tok[0].startswith("#"):
                continue
            k = tok[0]
            if k == "v":
                V.append([_real(tok, 1), _real(tok, 2), _real(tok, 3)])
            elif k == "vt":
                VT.append([_real(tok, 1), _real(tok, 2)])
            elif k == "vn":
                VN.append([_real(tok, 1), _real(tok, 2), _real(tok, 3)])
            elif k in ("o", "g"):
                cur_shape = []
                shapes.append(cur_shape)
            elif k == "usemtl":
                cur_mat = mtl_index.get(tok[1], None) if len(tok) > 1 else None
            elif k == "mtllib":
                for name in tok[1:]:                 # the first file that can be opened
                    got = parse_mtl(os.path.join(base, name))
                    if got is None:
                        continue
                    for m in got:
                        mtl_index.setdefault(m["name"], len(mtl))   # the first definition of a name wins
                        mtl.append(m)
                    break
            elif k == "f":
                if cur_shape is None:
                    cur_shape = []
                    shapes.append(cur_shape)
                refs = []
                for s in tok[1:]:                    # every word is a corner, '#' included
                    parts = s.split("/")
                    vi = fix(parts[0], len(V))
                    ti = fix(parts[1], len(VT)) if len(parts) > 1 and parts[1] != "" else -1
                    if len(parts) > 1 and parts[1] == "" and len(parts) < 3:
                        raise ObjLoadError("face index %r" % s)
                    ni = fix(parts[2], len(VN)) if len(parts) > 2 else -1
                    refs.append((vi, ti, ni))
                if len(refs) >= 3:
                    cur_shape.append((refs, cur_mat))
    # positive indices may name elements defined further down; beyond the end of the file the reference reads out of bounds
    # (faces dropped here and in the product)
    Vf = np.asarray(V, np.float32).reshape(-1, 3)
    VTf = np.asarray(VT, np.float32).reshape(-1, 2)
    VNf = np.asarray(VN, np.float32).reshape(-1, 3)
    flat_shapes = []
    for s in shapes:
        tris = []
        for refs, mat in s:
            if any(r[0] >= len(Vf) for r in refs):
                continue
            for (a, b, c) in triangulate([Vf[r[0]] for r in refs]):
                tris.append(((refs[a], refs[b], refs[c]), mat))
        if tris:
            flat_shapes.append(tris)
    shapes = flat_shapes

    # materials: MTL order, then the fallback Lambertian(0.5)
    textures, tex_index = [], {}

    def tex(fn):
        if not fn or texture_loader is None or not base:
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

    n_tri = sum(len(s) for s in shapes)
    tris = np.zeros(n_tri, ffi.TRI_DTYPE)
    k = 0
    for si, s in enumerate(shapes):
        for refs, mname in s:
            P = [Vf[r[0]] for r in refs]
            tris[k]["v0"], tris[k]["v1"], tris[k]["v2"] = P
            st = []
            for r in refs:
                st += list(VTf[r[1]]) if 0 <= r[1] < len(VTf) else [0.0, 0.0]
            tris[k]["st"] = st
            if all(0 <= r[2] < len(VNf) for r in refs):
                tris[k]["n0"], tris[k]["n1"], tris[k]["n2"] = [VNf[r[2]] for r in refs]
            else:
                tris[k]["n0"] = tris[k]["n1"] = tris[k]["n2"] = _flat_normal(P)
            tris[k]["material"] = mname if mname is not None else fb
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
