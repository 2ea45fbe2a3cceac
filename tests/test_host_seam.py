"""CPU: host seam (scene load + camera) -- oracle AND product -- against the reference's own
tinyobjloader / glm (tests/golden/ref_seam.npz, produced by oracle/_ref from the headers under
/root/reference/external, see oracle/ref_seam_dump.cpp + oracle/make_golden.py)."""
import os
import textwrap

import numpy as np
import pytest

from conftest import GOLDEN, REFERENCE_MODELS, bits

REF = np.load(os.path.join(GOLDEN, "ref_seam.npz"))
HAVE_MODELS = os.path.isdir(REFERENCE_MODELS)


def _mats(which, fn_look, fn_persp, fn_inv, row):
    v = fn_look(row[0:3], row[3:6], row[6:9])
    p = fn_persp(*[np.float32(x) for x in row[9:13]])
    p = np.array(p, dtype=np.float32)
    p[5] *= np.float32(-1)
    return v, p, fn_inv(v), fn_inv(p)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_camera_matrices_match_glm(impl, O, pa):
    look, persp, inv, refcam = ((O.lookat, O.perspective, O.inverse4, O.reference_camera) if impl == "oracle" else
                                (pa.look_at, pa.perspective, pa.inverse4, pa.reference_camera))
    for row in REF["cameras"]:  # main.cpp:1465-1480 at several extents
        W, H = int(row[0]), int(row[1])
        v, p = refcam(W, H)
        want = row[2:].reshape(4, 16)
        for got, w in zip((v, p, inv(v), inv(p)), want):
            assert np.array_equal(bits(got), bits(w)), (impl, W, H)
    for row in REF["camkat"]:  # generic poses
        got = _mats(impl, look, persp, inv, row)
        for g, w in zip(got, row[13:].reshape(4, 16)):
            assert np.array_equal(bits(g), bits(w)), impl


def test_reference_camera_facts(pa):
    """view = translation by (0,-1,-3.5); proj diag with aspect 1."""
    v, p = pa.reference_camera(512, 512)
    assert list(v[12:15]) == [0.0, -1.0, -3.5]
    assert p[0] == np.float32(2.41421342) and p[5] == np.float32(-2.41421342)
    assert p[10] == np.float32(-1.02020204) and p[11] == -1.0 and p[14] == np.float32(-0.202020213)


def _tinyobj_arrays(obj):
    idx = np.array([(a, c, b) for s in obj["shapes"] for (a, b, c) in s["indices"]], dtype=np.int32).reshape(-1, 3)
    mid = np.array([x for s in obj["shapes"] for x in s["material_ids"]], dtype=np.int32)
    mats = np.array([list(x["diffuse"]) + list(x["emission"]) for x in obj["materials"]], dtype=np.float32).reshape(-1, 6)
    return idx, mid, mats


@pytest.mark.skipif(not HAVE_MODELS, reason="/root/reference/models only exists in the build container")
@pytest.mark.parametrize("model", ["cornell", "plane", "teapot", "viking_room"])
def test_obj_oracle_matches_tinyobjloader(model):
    import obj_oracle as oo
    obj = oo.parse_obj(os.path.join(REFERENCE_MODELS, model + ".obj"))
    assert np.array_equal(bits(obj["vertices"]), bits(REF[model + "_v"]))
    assert np.array_equal(bits(obj["normals"]), bits(REF[model + "_vn"]))
    assert np.array_equal(bits(obj["texcoords"]), bits(REF[model + "_vt"]))
    idx, mid, mats = _tinyobj_arrays(obj)
    assert np.array_equal(idx, REF[model + "_idx"])
    assert np.array_equal(mid, REF[model + "_mid"])
    assert np.array_equal(bits(mats), bits(REF[model + "_mat"]))
    assert len(obj["shapes"]) == REF[model + "_shapes"].shape[0]


@pytest.mark.skipif(not HAVE_MODELS, reason="/root/reference/models only exists in the build container")
@pytest.mark.parametrize("model", ["cornell", "plane", "teapot", "viking_room"])
def test_product_loader_matches_oracle_on_reference_models(model, pa):
    import obj_oracle as oo
    path = os.path.join(REFERENCE_MODELS, model + ".obj")
    got, want = pa.load_obj(path), oo.load_model(path)
    for k in ("vertices", "indices", "materials", "mat_indices"):
        assert np.array_equal(getattr(got, k).view(np.uint32), want[k].view(np.uint32)), (model, k)


@pytest.mark.parametrize("model", ["cornell", "plane", "teapot", "viking_room"])
def test_scene_from_tinyobj_fixture_product_equals_oracle(model, pa):
    """loadModel's de-duplication loop alone, over the arrays the reference's own tinyobjloader produced (fixture): product
    (pt_scene_from_tinyobj) == oracle, bit for bit; where the reference's files exist also == pt_load_obj on the file.
    This is how the GPU box gets viking_room / teapot (BASELINE configs[2], [3]): as data, not as files."""
    import obj_oracle as oo
    from conftest import reference_mesh_arrays
    arrays = reference_mesh_arrays(model)
    got = pa.scene_from_tinyobj(*arrays)
    want = oo.model_from_tinyobj(oo.tinyobj_from_arrays(*arrays))
    for k in ("vertices", "indices", "materials", "mat_indices"):
        assert np.array_equal(getattr(got, k).view(np.uint32), want[k].view(np.uint32)), (model, k)
    if HAVE_MODELS:
        ref = pa.load_obj(os.path.join(REFERENCE_MODELS, model + ".obj"))
        for k in ("vertices", "indices", "materials", "mat_indices"):
            assert np.array_equal(getattr(got, k).view(np.uint32), getattr(ref, k).view(np.uint32)), (model, k)
    # SURVEY.md section 8(c) facts
    facts = {"cornell": (32, 64, 5), "plane": (2, None, 1), "teapot": (6320, None, 1), "viking_room": (3828, 4725, 1)}[model]
    assert got.n_tris == facts[0] and got.materials.shape[0] == facts[2]
    if facts[1] is not None:
        assert got.vertices.shape[0] == facts[1]


def test_scene_from_tinyobj_rejects_bad_arrays(pa):
    v = np.zeros(9, dtype=np.float32)
    idx = np.array([[0, -1, -1], [1, -1, -1], [7, -1, -1]], dtype=np.int32)
    with pytest.raises(pa.PathTracerError):
        pa.scene_from_tinyobj(v, [], [], idx, [-1], np.zeros((0, 6), np.float32))  # vertex 7 does not exist
    idx[2, 0] = 2
    with pytest.raises(pa.PathTracerError):
        pa.scene_from_tinyobj(v, [], [], idx, [3], np.zeros((0, 6), np.float32))  # material 3 does not exist
    sc = pa.scene_from_tinyobj(v, [], [], idx, [-1], np.zeros((0, 6), np.float32))
    assert sc.n_tris == 1 and list(sc.mat_indices) == [0] and list(sc.vertices[0, 3:6]) == [0.0, 0.0, 1.0]  # degenerate -> (0,0,1)


def test_generated_cornell_equals_reference_tinyobj_output(pa, tmp_path):
    """The Cornell box emitted by pathtracer_amd.scenes parses to exactly what the reference's
    tinyobjloader makes of the reference's models/cornell.obj (fixture), signs of zero included."""
    import obj_oracle as oo
    path = pa.scenes.write_cornell_box(str(tmp_path))
    obj = oo.parse_obj(path)
    assert np.array_equal(bits(obj["vertices"]), bits(REF["cornell_v"]))
    assert np.array_equal(bits(obj["normals"]), bits(REF["cornell_vn"]))
    idx, mid, mats = _tinyobj_arrays(obj)
    assert np.array_equal(idx, REF["cornell_idx"]) and np.array_equal(mid, REF["cornell_mid"])
    assert np.array_equal(bits(mats), bits(REF["cornell_mat"]))
    # loadModel facts quoted in SURVEY.md section 4 / 8(c)
    sc = pa.load_obj(path)
    want = oo.load_model(path)
    for k in ("vertices", "indices", "materials", "mat_indices"):
        assert np.array_equal(getattr(sc, k).view(np.uint32), want[k].view(np.uint32)), k
    assert sc.vertices.shape == (64, 11) and sc.indices.size == 96 and sc.materials.shape == (5, 6)
    assert list(sc.indices[:12]) == [0, 1, 2, 0, 2, 3, 4, 5, 6, 4, 6, 7]
    assert list(sc.mat_indices) == [2, 2] + [3] * 26 + [0, 0, 1, 1]
    assert list(sc.materials[4]) == [0.5, 0.5, 0.5, 0.0, 0.0, 0.0]
    assert list(sc.materials[1]) == [0.0, 0.0, 0.0, 15.0, 15.0, 15.0]


EDGE_OBJ = textwrap.dedent("""\
    # edge cases: CRLF-free text; relative indices, pentagon fan, exponents, missing normals/uvs
    mtllib edge.mtl other.mtl
    v 0 0 0
    v 1.5e0 0 0
    v 1 1 -2.5E-1
    v 0 1 +.5
    v -0.000000 2 0.125
    vt 0.25 0.75
    vt 1 0
    vn 0 0 1
    usemtl glow
    f 1/1/1 2/2/1 3/1/1 4/2/1 5/1/1
    usemtl nosuch
    f -5 -4 -3
    g second
    usemtl matte
    f 1//1 3//1 4//1
    f 2/2 3/1 5/2
    """)
EDGE_MTL = "newmtl matte\nKd 0.1 0.2 0.3\nnewmtl glow\nKd 0 0 0\nKe 2 3.5 1e1\n"


@pytest.mark.parametrize("newline", ["\n", "\r\n"])
def test_loader_edge_cases_product_equals_oracle(pa, tmp_path, newline):
    import obj_oracle as oo
    p = tmp_path / "edge.obj"
    p.write_bytes(EDGE_OBJ.replace("\n", newline).encode())
    (tmp_path / "edge.mtl").write_bytes(EDGE_MTL.replace("\n", newline).encode())
    got, want = pa.load_obj(str(p)), oo.load_model(str(p))
    for k in ("vertices", "indices", "materials", "mat_indices"):
        assert np.array_equal(getattr(got, k).view(np.uint32), want[k].view(np.uint32)), k
    # pentagon -> 3 triangles (fan), unknown material -> grey fallback index (= #materials in the .mtl)
    assert got.indices.size == 3 * (3 + 1 + 2)
    assert list(got.mat_indices) == [1, 1, 1, 2, 0, 0]
    assert got.vertices[0, 9] == 0.25 and got.vertices[0, 10] == 0.25  # V flipped: 1 - 0.75


def test_loader_map_kd_product_equals_oracle(pa, tmp_path):
    """tinyobj's material_t::diffuse_texname (ParseTextureNameAndOption, tiny_obj_loader.h:757-850): options skipped with
    their arities, relative names resolved against the OBJ's directory, absolute names kept."""
    import obj_oracle as oo
    mtl = ("newmtl a\nKd 1 1 1\nmap_Kd wood.png\n"
           "newmtl b\nmap_Kd -clamp on -s 2 2 1 -o 0.5 0 0 -mm 0 1 -imfchan r tex/brick.png\nKd 0.5 0.5 0.5\n"
           "newmtl c\nKd 0 1 0\nmap_Ka notthis.png\n"
           "newmtl d\nmap_Kd\t-blendu off /abs/olute.png \nmap_Bump -bm 2 bump.png\n"
           "newmtl e\nmap_Kd -s 4 tile.png\n")
    p = tmp_path / "maps.obj"
    p.write_text("mtllib maps.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nusemtl b\nf 1//1 2//1 3//1\n")
    (tmp_path / "maps.mtl").write_text(mtl)
    got, want = pa.load_obj(str(p)), oo.load_model(str(p))
    d = str(tmp_path) + "/"
    assert got.diffuse_maps == (d + "wood.png", d + "tex/brick.png", "", "/abs/olute.png", d + "tile.png", "")
    assert got.diffuse_maps == want["diffuse_maps"]
    for k in ("vertices", "indices", "materials", "mat_indices"):
        assert np.array_equal(getattr(got, k).view(np.uint32), want[k].view(np.uint32)), k
    assert list(got.materials[1]) == [0.5, 0.5, 0.5, 0, 0, 0]  # statements after a map_Kd still parse


@pytest.mark.skipif(not HAVE_MODELS, reason="/root/reference/models only exists in the build container")
def test_reference_models_name_no_diffuse_map(pa):
    """None of the reference's four MTL files has a map_Kd (viking_room.png ships beside viking_room.mtl without being
    named in it): the texture of BASELINE configs[2] is set explicitly (pt_render -tex, Renderer.setTexture)."""
    for model in ("cornell", "plane", "teapot", "viking_room"):
        sc = pa.load_obj(os.path.join(REFERENCE_MODELS, model + ".obj"))
        assert set(sc.diffuse_maps) == {""}, model


def test_loader_missing_file_and_missing_mtl(pa, tmp_path):
    with pytest.raises(pa.PathTracerError):
        pa.load_obj(str(tmp_path / "does_not_exist.obj"))
    p = tmp_path / "nomtl.obj"
    p.write_text("mtllib missing.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    sc = pa.load_obj(str(p))
    assert "not found" in sc.warnings
    assert sc.materials.shape == (1, 6) and list(sc.mat_indices) == [0]  # grey fallback only


def test_loader_empty_obj(pa, tmp_path):
    p = tmp_path / "empty.obj"
    p.write_text("# nothing\n")
    sc = pa.load_obj(str(p))
    assert sc.indices.size == 0 and sc.vertices.shape[0] == 0 and sc.materials.shape == (1, 6)


def _decode_png(path):
    """Minimal PNG reader for 8-bit RGBA, filter type 0 (what pt_write_png emits); checks every CRC."""
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == (zlib.crc32(typ + body) & 0xFFFFFFFF), typ
        chunks.append((typ, body))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"sRGB", b"IDAT", b"IEND"]
    w, h, depth, ctype, comp, flt, inter = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, flt, inter) == (8, 6, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[2][1]), dtype=np.uint8).reshape(h, 1 + 4 * w)
    assert np.all(raw[:, 0] == 0)
    return raw[:, 1:].reshape(h, w, 4)


@pytest.mark.parametrize("shape", [(1, 1), (5, 7), (300, 257)])  # the last one needs several stored blocks
def test_png_writer_round_trip(pa, tmp_path, shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, size=(shape[0], shape[1], 4), dtype=np.uint8)
    path = tmp_path / "a.png"
    pa.write_png(str(path), img)
    assert np.array_equal(_decode_png(path), img)
    with pytest.raises(pa.PathTracerError):
        pa.write_png(str(tmp_path / "no_such_dir" / "a.png"), img)


def test_ppm_and_pfm_writers(pa, tmp_path):
    rng = np.random.default_rng(1)
    img8 = rng.integers(0, 256, size=(4, 6, 4), dtype=np.uint8)
    pa.write_ppm(str(tmp_path / "a.ppm"), img8)
    blob = open(tmp_path / "a.ppm", "rb").read()
    assert blob.startswith(b"P6\n6 4\n255\n") and blob[len(b"P6\n6 4\n255\n"):] == img8[..., :3].tobytes()
    imgf = rng.normal(size=(4, 6, 4)).astype(np.float32)
    pa.write_pfm(str(tmp_path / "a.pfm"), imgf)
    blob = open(tmp_path / "a.pfm", "rb").read()
    head = b"PF\n6 4\n-1.0\n"
    assert blob.startswith(head)
    got = np.frombuffer(blob[len(head):], dtype="<f4").reshape(4, 6, 3)
    assert np.array_equal(got, imgf[::-1, :, :3])  # PFM stores the bottom row first
