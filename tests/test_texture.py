"""Diffuse map (BASELINE config 3 "textured").  The reference carries Vertex::texCoord to
raytrace.rchit but never samples its textures/viking_room.png, so the lookup is this build's definition (parity
unpinned by the reference): the CPU tests pin the oracle's lookup against a pure-Python restatement and the PNG
decoder against zlib; the GPU tests compare the HIP path with the oracle bit for bit."""
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import bits

REF_PNG = "/root/reference/textures/viking_room.png"  # only in the build container


def make_png(pix, colour, filters, palette=None, trns=None, level=6, split=1):
    """pix: (H, W, C) uint8 samples as stored; applies the given PNG filter type per row (cycled)."""
    h, w, ch = pix.shape
    rows = []
    prev = np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        cur = pix[y].reshape(-1).astype(np.int32)
        f = filters[y % len(filters)]
        a = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        c = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if f == 0: pred = np.zeros_like(cur)
        elif f == 1: pred = a
        elif f == 2: pred = prev
        elif f == 3: pred = (a + prev) >> 1
        else:
            p = a + prev - c
            pa_, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa_ <= pb) & (pa_ <= pc), a, np.where(pb <= pc, prev, c))
        rows.append(bytes([f]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
        prev = cur
    z = zlib.compress(b"".join(rows), level)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, colour, 0, 0, 0))
    if palette is not None: out += chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    if trns is not None: out += chunk(b"tRNS", trns.astype(np.uint8).tobytes())
    step = (len(z) + split - 1) // split
    for i in range(0, len(z), step): out += chunk(b"IDAT", z[i:i + step])
    return out + chunk(b"IEND", b"")


@pytest.mark.parametrize("colour,ch", [(0, 1), (2, 3), (4, 2), (6, 4), (3, 1)])
@pytest.mark.parametrize("level", [0, 1, 9])
def test_png_reader_against_zlib(pa, tmp_path, colour, ch, level):
    """All five filter types, every supported colour type, stored / fixed / dynamic Huffman blocks, split IDAT."""
    rng = np.random.default_rng(colour * 10 + level)
    h, w = 41, 67
    smooth = (np.add.outer(np.arange(h), np.arange(w))[..., None] * np.arange(1, ch + 1) + rng.integers(0, 9, (h, w, ch))) % 256
    pix = smooth.astype(np.uint8)
    palette = trns = None
    if colour == 3:
        palette, trns = rng.integers(0, 256, (256, 3)), rng.integers(0, 256, 100)
    path = tmp_path / "t.png"
    path.write_bytes(make_png(pix, colour, [0, 1, 2, 3, 4], palette, trns, level, split=3))
    got = pa.load_png(str(path))
    want = np.full((h, w, 4), 255, dtype=np.uint8)
    if colour == 0: want[..., :3] = pix
    elif colour == 2: want[..., :3] = pix
    elif colour == 4: want[..., :3], want[..., 3] = pix[..., :1], pix[..., 1]
    elif colour == 6: want = pix
    else:
        want[..., :3] = palette[pix[..., 0]]
        want[..., 3] = np.where(pix[..., 0] < 100, trns[np.minimum(pix[..., 0], 99)], 255)
    assert np.array_equal(got, want)


def test_png_reader_rejects_damage(pa, tmp_path):
    data = bytearray(make_png(np.zeros((8, 8, 3), np.uint8) + 7, 2, [1]))
    for name, mutate in [("crc", lambda d: d.__setitem__(40, d[40] ^ 1)), ("sig", lambda d: d.__setitem__(1, 0x51)),
                         ("cut", lambda d: d.__delitem__(slice(len(d) - 20, len(d))))]:
        d = bytearray(data)
        mutate(d)
        p = tmp_path / (name + ".png")
        p.write_bytes(bytes(d))
        with pytest.raises(pa.PathTracerError):
            pa.load_png(str(p))
    with pytest.raises(pa.PathTracerError):
        pa.load_png(str(tmp_path / "missing.png"))


def test_png_writer_reader_round_trip(pa, tmp_path):
    img = np.random.default_rng(3).integers(0, 256, (19, 31, 4), dtype=np.uint8)
    pa.write_png(str(tmp_path / "w.png"), img)
    assert np.array_equal(pa.load_png(str(tmp_path / "w.png")), img)


@pytest.mark.skipif(not os.path.exists(REF_PNG), reason="reference texture only exists in the build container")
def test_reference_texture_decodes_like_zlib(pa):
    d = open(REF_PNG, "rb").read()
    pos, idat = 8, b""
    while pos < len(d):
        n, ty = struct.unpack(">I", d[pos:pos + 4])[0], d[pos + 4:pos + 8]
        if ty == b"IDAT": idat += d[pos + 8:pos + 8 + n]
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(1024, 1024 * 3 + 1)
    got = pa.load_png(REF_PNG)
    assert got.shape == (1024, 1024, 4) and np.all(got[..., 3] == 255)
    # rows with filter "up" (or none) can be checked without a full unfilter: row = stored + decoded row above
    flat = got[..., :3].reshape(1024, -1).astype(np.int32)
    for f, above in ((0, 0), (2, 1)):
        rows = np.nonzero(raw[:, 0] == f)[0]
        rows = rows[rows > 0]
        if len(rows):
            assert np.array_equal(flat[rows], (raw[rows, 1:].astype(np.int32) + above * flat[rows - 1]) & 255)
    # filter "sub": running sum along the row per channel
    rows = np.nonzero(raw[:, 0] == 1)[0]
    if len(rows):
        want = np.cumsum(raw[rows, 1:].reshape(len(rows), 1024, 3).astype(np.int64), axis=1) & 255
        assert np.array_equal(got[rows, :, :3], want)
    assert len(np.unique(raw[:, 0])) >= 2
    # the committed fixture the GPU box renders configs[2] with (decoded by Pillow: tests/golden/make_viking_texture.py)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "viking_room_texels.npz"))["rgba8"]
    assert np.array_equal(got, gold)


def test_srgb_table_matches_oracle(pa, O):
    lut = pa.srgb8_to_linear(np.arange(256))
    assert np.array_equal(bits(lut), bits(O.srgb8_to_linear_lut()))
    assert lut[0] == 0.0 and lut[255] == 1.0 and np.all(np.diff(lut) > 0)


def test_oracle_lookup_equals_python_restatement(O):
    """pto_sample_texture against the same expression tree written in numpy float32 scalars, including
    coordinates that wrap, sit exactly on texel centres / edges, and the u - floor(u) == 1.0 corner."""
    f = np.float32
    rng = np.random.default_rng(11)
    tex = rng.random((5, 7, 4), dtype=np.float32)
    h, w = 5, 7
    pts = [(0.0, 0.0), (0.5 / w, 0.5 / h), (1.0, 1.0), (-1e-9, -1e-9), (3.25, -2.75), (0.999999, 0.5), (-0.3, 7.7)]
    pts += [tuple(p) for p in rng.uniform(-3, 3, (200, 2))]
    for u, v in pts:
        u, v = f(u), f(v)
        su, sv = f(u - np.floor(u)), f(v - np.floor(v))
        x, y = f(f(su * f(w)) - f(0.5)), f(f(sv * f(h)) - f(0.5))
        x0, y0 = np.floor(x), np.floor(y)
        fx, fy = f(x - x0), f(y - y0)
        ix0, iy0 = int(x0), int(y0)
        ix1, iy1 = ix0 + 1, iy0 + 1
        if ix0 < 0: ix0 += w
        if iy0 < 0: iy0 += h
        if ix1 >= w: ix1 -= w
        if iy1 >= h: iy1 -= h
        gx, gy = f(f(1) - fx), f(f(1) - fy)
        want = [f(f(f(f(tex[iy0, ix0, k] * gx) + f(tex[iy0, ix1, k] * fx)) * gy) +
                  f(f(f(tex[iy1, ix0, k] * gx) + f(tex[iy1, ix1, k] * fx)) * fy)) for k in range(3)]
        assert np.array_equal(bits(O.sample_texture(tex, u, v)), bits(np.array(want, dtype=np.float32))), (u, v)


def test_oracle_textured_render_modulates_diffuse_only(O, pa):
    """A white texture changes nothing (x * 1.0 is exact); a black one leaves only directly visible emission."""
    sc = pa.scenes.textured_cornell()
    view, proj = pa.reference_camera(48, 40)
    args = dict(spp_per_frame=3, max_depth=6, seed=5)
    plain, rays = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices).render(view, proj, 48, 40, 2, **args)
    white = np.full((4, 4, 4), 255, dtype=np.uint8)
    img, r2 = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices, texture=white).render(view, proj, 48, 40, 2, **args)
    assert np.array_equal(bits(img), bits(plain)) and r2 == rays
    black = np.zeros((4, 4, 4), dtype=np.uint8)
    img, r3 = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices, texture=black).render(view, proj, 48, 40, 2, **args)
    assert r3 == rays  # paths are not terminated by a black albedo (as in the reference: only emission terminates)
    # every sample is 0 or exactly Ke = 15 (seen directly); 3 samples, 2 frames -> multiples of 15/6
    assert set(np.unique(img[..., :3]).tolist()) <= {2.5 * k for k in range(7)}


# ------------------------------------------------------------------------------------------------ GPU ----
PIPELINES = {"fused": 0, "split": 1, "persistent": 2}


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", list(PIPELINES))
def test_textured_image_equals_oracle(rt, pa, O, pipeline):
    sc, tex = pa.scenes.textured_cornell(), pa.scenes.random_texture()
    W, H = 160, 90
    view, proj = pa.reference_camera(W, H)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices, texture=tex)
    want, rays = osc.render(view, proj, W, H, 3, spp_per_frame=5, max_depth=8, seed=3)
    rt.setPipeline(PIPELINES[pipeline])
    rt.buildBlas(sc)
    rt.setTexture(tex)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=5, max_depth=8, seed=3)
    rt.raytrace(3)
    assert np.array_equal(bits(rt.readImage()), bits(want))
    assert rt.stats().rays_total == rays
    # removing the map restores the untextured image
    rt.setTexture(None)
    rt.resetFrame()
    rt.raytrace(3)
    plain, _ = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices).render(view, proj, W, H, 3, spp_per_frame=5,
                                                                                          max_depth=8, seed=3)
    assert np.array_equal(bits(rt.readImage()), bits(plain))


@pytest.mark.gpu
def test_textured_instances_and_large_map(rt, pa, O):
    """Texture + two-level scene (texCoord interpolation is in object space, unaffected by the transform), and a
    1024 x 512 map (the reference texture's order of magnitude)."""
    sc = pa.scenes.textured_cornell()
    tex = np.random.default_rng(5).integers(0, 256, (512, 1024, 4), dtype=np.uint8)
    xf = np.array([[1, 0, 0, -1.2, 0, 1, 0, 0, 0, 0, 1, 0], [0.8, 0, 0.2, 1.3, 0, 1.1, 0, 0.1, -0.2, 0, 0.8, -0.5]], dtype=np.float32)
    W, H = 128, 72
    view = pa.look_at((0.0, 1.0, 6.5), (0.0, 1.0, 0.0), (0, 1, 0))
    proj = pa.perspective(np.radians(45.0), W / H, 0.1, 100.0)
    proj[5] *= -1
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices, instances=xf, texture=tex)
    want, rays = osc.render(view, proj, W, H, 2, spp_per_frame=4, max_depth=6, seed=1)
    rt.uploadScene(sc)
    rt.setInstances(xf)
    rt.buildBlas()
    rt.setTexture(tex)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=4, max_depth=6, seed=1)
    rt.raytrace(2)
    assert np.array_equal(bits(rt.readImage()), bits(want))
    assert rt.stats().rays_total == rays
