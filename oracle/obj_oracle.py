"""OBJ/MTL scene-load ORACLE.  TEST INFRASTRUCTURE ONLY (never imported by the product).

Pure-Python restatement of what the reference's host seam does when it loads a scene:

* ``parse_obj`` / ``parse_mtl`` restate the subset of tinyobjloader v1.0.6
  (``external/tinyobjloader/tiny_obj_loader.h``, pinned by the vendored header) that the
  reference exercises: ``tryParseDouble`` (:474-587, digit-by-digit double accumulation, then a
  cast to float -- NOT a correctly rounded strtof), ``parseTriple``/``fixIndex`` (:425-429,
  :691-731), the ``v/vn/vt/f/usemtl/mtllib/g/o`` branches of ``LoadObj`` (:1424-1730) including
  the polygon -> triangle-fan rule (:908-931) and the ``usemtl`` flush into the same shape
  (:1528-1555), and ``LoadMtl``'s ``newmtl/Kd/Ke`` handling (:965-1330).
* ``load_model`` restates ``PathTracer::loadModel`` (``main.cpp:1101-1190``): materials ->
  ``Material{diffuse, emission}`` + appended grey(0.5) fallback, per-face ``matIndices`` (``-1`` ->
  fallback), first-occurrence vertex de-duplication with float ``==`` semantics, white vertex
  colour, V flipped.

Pinned against the real tinyobjloader through ``oracle/_ref`` (``tests/golden/ref_seam.npz``).

One documented extension (the reference has undefined behaviour there, ``main.cpp:1162-1164``):
if a face vertex has no normal (``normal_index == -1``, e.g. ``teapot.obj``) the flat geometric
normal of its triangle is synthesised.
"""
from __future__ import annotations

import math
import os

import numpy as np

_POW_LUT = [1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001]


def try_parse_double(s: str):
    """tiny_obj_loader.h:474-587.  Returns a Python float or None on failure."""
    n = len(s)
    if n == 0:
        return None
    i = 0
    sign = "+"
    if s[i] in "+-":
        sign = s[i]
        i += 1
    elif s[i].isdigit():
        pass
    else:
        return None
    mantissa = 0.0
    exponent = 0
    read = 0
    while i < n and s[i] in "0123456789":
        mantissa *= 10
        mantissa += ord(s[i]) - 0x30
        i += 1
        read += 1
    if read == 0:
        return None
    done = i >= n
    if not done:
        if s[i] == ".":
            i += 1
            read = 1
            while i < n and s[i] in "0123456789":
                mantissa += (ord(s[i]) - 0x30) * (_POW_LUT[read] if read < 8 else math.pow(10.0, -read))
                read += 1
                i += 1
        elif s[i] in "eE":
            pass
        else:
            done = True
    if not done and i < n and s[i] in "eE":
        i += 1
        exp_sign = "+"
        if i < n and s[i] in "+-":
            exp_sign = s[i]
            i += 1
        elif i < n and s[i] in "0123456789":
            pass
        else:
            return None
        read = 0
        while i < n and s[i] in "0123456789":
            exponent = exponent * 10 + (ord(s[i]) - 0x30)
            i += 1
            read += 1
        exponent *= 1 if exp_sign == "+" else -1
        if read == 0:
            return None
    val = math.ldexp(mantissa * math.pow(5.0, exponent), exponent) if exponent else mantissa
    return (1 if sign == "+" else -1) * val


class _Tok:
    """Cursor over one line, mimicking the `const char **token` walkers."""

    def __init__(self, s: str):
        self.s = s
        self.i = 0

    def skip_ws(self, chars=" \t"):
        while self.i < len(self.s) and self.s[self.i] in chars:
            self.i += 1

    def word(self, stops=" \t\r"):
        j = self.i
        while j < len(self.s) and self.s[j] not in stops:
            j += 1
        w = self.s[self.i:j]
        self.i = j
        return w

    def real(self, default=0.0):
        self.skip_ws()
        w = self.word()
        v = try_parse_double(w)
        return np.float32(default if v is None else v)

    def eol(self):
        return self.i >= len(self.s) or self.s[self.i] in "\r\n"


def _atoi(s: str, i: int) -> int:
    # C atoi: skip whitespace, optional sign, digits
    n = len(s)
    while i < n and s[i] in " \t\n\v\f\r":
        i += 1
    sgn = 1
    if i < n and s[i] in "+-":
        sgn = -1 if s[i] == "-" else 1
        i += 1
    v = 0
    while i < n and s[i].isdigit():
        v = v * 10 + (ord(s[i]) - 48)
        i += 1
    return sgn * v


def _fix_index(idx: int, n: int) -> int:
    if idx > 0:
        return idx - 1
    if idx == 0:
        return 0
    return n + idx


def _parse_triple(t: _Tok, vsize, vnsize, vtsize):
    v = vt = vn = -1
    v = _fix_index(_atoi(t.s, t.i), vsize)
    t.word("/ \t\r")
    if t.i >= len(t.s) or t.s[t.i] != "/":
        return v, vt, vn
    t.i += 1
    if t.i < len(t.s) and t.s[t.i] == "/":
        t.i += 1
        vn = _fix_index(_atoi(t.s, t.i), vnsize)
        t.word("/ \t\r")
        return v, vt, vn
    vt = _fix_index(_atoi(t.s, t.i), vtsize)
    t.word("/ \t\r")
    if t.i >= len(t.s) or t.s[t.i] != "/":
        return v, vt, vn
    t.i += 1
    vn = _fix_index(_atoi(t.s, t.i), vnsize)
    t.word("/ \t\r")
    return v, vt, vn


def _lines(text: str):
    # safeGetline: \n, \r\n and bare \r all end a line
    return text.replace("\r\n", "\n").replace("\r", "\n").split("\n")


def _sscanf_s(s: str) -> str:
    # sscanf("%s"): skip leading whitespace, read until whitespace
    return s.split()[0] if s.split() else ""


_TEX_OPTION_WORDS = {"-blendu": 1, "-blendv": 1, "-clamp": 1, "-boost": 1, "-bm": 1, "-type": 1, "-imfchan": 1, "-texres": 1,
                     "-mm": 2, "-o": 3, "-s": 3, "-t": 3}


def texture_name(rest: str) -> str:
    """ParseTextureNameAndOption (tiny_obj_loader.h:757-850) for its file name only: options are skipped with their
    arities, the last plain word is the name.  (-o/-s/-t take three numbers there; one to three here, as the MTL
    specification allows, never the last word of the statement.)"""
    words = rest.split()
    name, i = "", 0
    while i < len(words):
        n = _TEX_OPTION_WORDS.get(words[i])
        if n is None:
            name = words[i]
        elif n < 3:
            i += n
        else:
            k = 0
            while k < 3 and i + 2 < len(words) and words[i + 1][0] in "+-.0123456789":
                i += 1
                k += 1
        i += 1
    return name


def parse_mtl(text: str, materials: list, material_map: dict):
    """LoadMtl (tiny_obj_loader.h:965-1330) restricted to name/diffuse/emission."""
    def fresh():
        return {"name": "", "diffuse": [np.float32(0)] * 3, "emission": [np.float32(0)] * 3, "diffuse_texname": ""}

    cur = fresh()

    for line in _lines(text):
        line = line.rstrip(" \t")
        if not line:
            continue
        t = line.lstrip(" \t")
        if not t or t[0] == "#":
            continue
        if t.startswith("newmtl") and len(t) > 6 and t[6] in " \t":
            if cur["name"]:
                material_map.setdefault(cur["name"], len(materials))
                materials.append(cur)
            cur = fresh()
            cur["name"] = _sscanf_s(t[7:])
            continue
        if len(t) > 2 and t[0] == "K" and t[1] in "de" and t[2] in " \t":
            tok = _Tok(t)
            tok.i = 2
            rgb = [tok.real(), tok.real(), tok.real()]
            cur["diffuse" if t[1] == "d" else "emission"] = rgb
            continue
        if t.startswith("map_Kd") and len(t) > 6 and t[6] in " \t":
            cur["diffuse_texname"] = texture_name(t[7:])
            continue
    material_map.setdefault(cur["name"], len(materials))
    materials.append(cur)


def parse_obj(path: str):
    """LoadObj(attrib, shapes, materials, err, filename, mtl_basedir, triangulate=true)."""
    base = os.path.dirname(path)
    mtl_dir = (base if base else "") + "/"
    with open(path, "r", newline="") as f:
        text = f.read()
    v, vn, vt = [], [], []
    shapes = []
    materials, material_map = [], {}
    shape = {"indices": [], "material_ids": []}
    face_group = []
    material = -1

    def export(shape, face_group, material):
        if not face_group:
            return False
        for face in face_group:
            i0 = face[0]
            i2 = face[1] if len(face) > 1 else None
            for k in range(2, len(face)):
                i1 = i2
                i2 = face[k]
                shape["indices"] += [i0, i1, i2]
                shape["material_ids"].append(material)
        return True

    for line in _lines(text):
        if not line:
            continue
        t = line.lstrip(" \t")
        if not t or t[0] == "#":
            continue
        c1 = t[1] if len(t) > 1 else "\0"
        c2 = t[2] if len(t) > 2 else "\0"
        if t[0] == "v" and c1 in " \t":
            tok = _Tok(t); tok.i = 2
            v += [tok.real(), tok.real(), tok.real()]
            continue
        if t[0] == "v" and c1 == "n" and c2 in " \t":
            tok = _Tok(t); tok.i = 3
            vn += [tok.real(), tok.real(), tok.real()]
            continue
        if t[0] == "v" and c1 == "t" and c2 in " \t":
            tok = _Tok(t); tok.i = 3
            vt += [tok.real(), tok.real()]
            continue
        if t[0] == "f" and c1 in " \t":
            tok = _Tok(t); tok.i = 2
            tok.skip_ws()
            face = []
            while not tok.eol():
                face.append(_parse_triple(tok, len(v) // 3, len(vn) // 3, len(vt) // 2))
                tok.skip_ws(" \t\r")
            face_group.append(face)
            continue
        if t.startswith("usemtl") and len(t) > 6 and t[6] in " \t":
            name = _sscanf_s(t[7:])
            new_id = material_map.get(name, -1)
            if new_id != material:
                export(shape, face_group, material)
                face_group = []
                material = new_id
            continue
        if t.startswith("mtllib") and len(t) > 6 and t[6] in " \t":
            for fn in t[7:].split(" "):
                p = mtl_dir + fn
                if os.path.isfile(p):
                    with open(p, "r", newline="") as mf:
                        parse_mtl(mf.read(), materials, material_map)
                    break
            continue
        if t[0] in "go" and c1 in " \t":
            if export(shape, face_group, material):
                shapes.append(shape)
            shape = {"indices": [], "material_ids": []}
            face_group = []
            continue
    if export(shape, face_group, material) or shape["indices"]:
        shapes.append(shape)
    return {
        "vertices": np.array(v, dtype=np.float32),
        "normals": np.array(vn, dtype=np.float32),
        "texcoords": np.array(vt, dtype=np.float32),
        "shapes": shapes,
        "materials": materials,
    }


def load_model(path: str):
    """PathTracer::loadModel (main.cpp:1101-1190).  Returns dict of numpy arrays:
    vertices (n,11) f32, indices (m,) u32, materials (k,6) f32, mat_indices (m/3,) u32, diffuse_maps (k strings)."""
    obj = parse_obj(path)
    model = model_from_tinyobj(obj)
    # material_t::diffuse_texname, which loadModel itself never reads: relative names against the OBJ's directory
    base = os.path.dirname(path)
    mtl_dir = (base if base else "") + "/"
    model["diffuse_maps"] = tuple(
        (t if not t or t[0] == "/" else mtl_dir + t) for t in (m.get("diffuse_texname", "") for m in obj["materials"])) + ("",)  # the grey fallback material has none
    return model


def tinyobj_from_arrays(vertices, normals, texcoords, indices, material_ids, materials):
    """The dict parse_obj returns, rebuilt from flat arrays as the reference's own tinyobjloader dumped them
    (tests/golden/ref_seam.npz: index_t triples are (vertex, normal, texcoord); one shape)."""
    idx = np.asarray(indices, dtype=np.int64).reshape(-1, 3)
    mats = np.asarray(materials, dtype=np.float32).reshape(-1, 6)
    return {
        "vertices": np.asarray(vertices, dtype=np.float32).reshape(-1),
        "normals": np.asarray(normals, dtype=np.float32).reshape(-1),
        "texcoords": np.asarray(texcoords, dtype=np.float32).reshape(-1),
        "shapes": [{"indices": [(int(a), int(c), int(b)) for a, b, c in idx],  # oracle order: (v, vt, vn)
                    "material_ids": [int(x) for x in np.asarray(material_ids).reshape(-1)]}],
        "materials": [{"name": str(i), "diffuse": list(m[:3]), "emission": list(m[3:])} for i, m in enumerate(mats)],
    }


def model_from_tinyobj(obj):
    """The loop of PathTracer::loadModel (main.cpp:1122-1189) over a parse_obj / tinyobj_from_arrays result."""
    mats = [list(m["diffuse"]) + list(m["emission"]) for m in obj["materials"]]
    mat_count = len(mats)
    mats.append([0.5, 0.5, 0.5, 0.0, 0.0, 0.0])
    V, N, T = obj["vertices"], obj["normals"], obj["texcoords"]
    unique = {}
    verts, indices, mat_indices = [], [], []
    for shape in obj["shapes"]:
        for m in shape["material_ids"]:
            mat_indices.append(mat_count if m == -1 else m)
        idx = shape["indices"]
        for tri in range(0, len(idx), 3):
            flat = None
            for (vi, ti, ni) in idx[tri:tri + 3]:
                pos = V[3 * vi:3 * vi + 3]
                if ni != -1:
                    nrm = N[3 * ni:3 * ni + 3]
                else:
                    if flat is None:  # documented extension: flat geometric normal
                        p = [V[3 * a:3 * a + 3].astype(np.float32) for (a, _, _) in idx[tri:tri + 3]]
                        e1, e2 = p[1] - p[0], p[2] - p[0]
                        c = np.array([e1[1] * e2[2] - e2[1] * e1[2], e1[2] * e2[0] - e2[2] * e1[0],
                                      e1[0] * e2[1] - e2[0] * e1[1]], dtype=np.float32)
                        l2 = np.float32(np.float32(c[0] * c[0] + c[1] * c[1]) + c[2] * c[2])
                        if l2 > 0:
                            inv = np.float32(1.0) / np.sqrt(l2, dtype=np.float32)
                            flat = (c * inv).astype(np.float32)
                        else:
                            flat = np.array([0, 0, 1], dtype=np.float32)
                    nrm = flat
                if ti != -1:
                    uv = (T[2 * ti], np.float32(1.0) - T[2 * ti + 1])
                else:
                    uv = (np.float32(0), np.float32(0))
                rec = (float(pos[0]), float(pos[1]), float(pos[2]), float(nrm[0]), float(nrm[1]), float(nrm[2]),
                       1.0, 1.0, 1.0, float(uv[0]), float(uv[1]))
                # float == semantics: -0.0 == 0.0 (tuples of Python floats compare/hash that way)
                k = unique.get(rec)
                if k is None:
                    k = len(verts)
                    unique[rec] = k
                    verts.append(rec)
                indices.append(k)
    return {
        "vertices": np.array(verts, dtype=np.float32).reshape(-1, 11),
        "indices": np.array(indices, dtype=np.uint32),
        "materials": np.array(mats, dtype=np.float32).reshape(-1, 6),
        "mat_indices": np.array(mat_indices, dtype=np.uint32),
    }
