"""Writes valid PNGs of every supported colour type / deflate block type plus damaged copies with valid chunk CRCs."""
import os, struct, sys, zlib
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
from test_texture import make_png

out = sys.argv[1]
rng = np.random.default_rng(1)


def chunks(d):
    pos, res = 8, []
    while pos < len(d):
        n = struct.unpack(">I", d[pos:pos + 4])[0]
        res.append((d[pos + 4:pos + 8], bytearray(d[pos + 8:pos + 8 + n])))
        pos += 12 + n
    return res


def assemble(ch):
    o = b"\x89PNG\r\n\x1a\n"
    for t, dd in ch:
        o += struct.pack(">I", len(dd)) + t + bytes(dd) + struct.pack(">I", zlib.crc32(t + bytes(dd)) & 0xFFFFFFFF)
    return o


n = 0
for colour, chn in [(0, 1), (2, 3), (3, 1), (4, 2), (6, 4)]:
    for level in (0, 1, 6, 9):
        pix = rng.integers(0, 256, (13, 17, chn), dtype=np.uint8) if level != 9 else (np.arange(13 * 17 * chn).reshape(13, 17, chn) % 7).astype(np.uint8)
        pal = rng.integers(0, 256, (256, 3)) if colour == 3 else None
        good = make_png(pix, colour, [0, 1, 2, 3, 4], pal, None, level)
        open(os.path.join(out, f"g{n}.png"), "wb").write(good); n += 1
        for k in range(120):
            ch = chunks(good)
            ti = rng.integers(0, len(ch)); t, dd = ch[ti]
            if len(dd) == 0: continue
            for _ in range(rng.integers(1, 4)): dd[rng.integers(0, len(dd))] = rng.integers(0, 256)
            if rng.random() < 0.2: ch[ti] = (t, dd[:rng.integers(0, len(dd))])
            open(os.path.join(out, f"f{n}.png"), "wb").write(assemble(ch)); n += 1
print("corpus", n)
