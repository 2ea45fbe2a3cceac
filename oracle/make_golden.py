"""Pack the reference-derived dumps of oracle/_ref (made by `make -C oracle _ref`, which drives
the reference's own tinyobjloader + glm headers under /root/reference) into the committed
fixture tests/golden/ref_seam.npz.  TEST INFRASTRUCTURE; run in the build container only."""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
ref = os.path.join(here, "_ref")
out = os.path.join(here, "..", "tests", "golden", "ref_seam.npz")


def rd(name, dt):
    return np.fromfile(os.path.join(ref, name), dtype=dt)


d = {}
for m in ("cornell", "plane", "teapot", "viking_room"):
    d[m + "_v"] = rd(m + ".v.f32", np.float32)
    d[m + "_vn"] = rd(m + ".vn.f32", np.float32)
    d[m + "_vt"] = rd(m + ".vt.f32", np.float32)
    d[m + "_idx"] = rd(m + ".idx.i32", np.int32).reshape(-1, 3)  # (v, vn, vt) per face-vertex
    d[m + "_mid"] = rd(m + ".mid.i32", np.int32)
    d[m + "_shapes"] = rd(m + ".shapes.i32", np.int32).reshape(-1, 2)
    d[m + "_mat"] = rd(m + ".mat.f32", np.float32).reshape(-1, 6)
d["cameras"] = rd("glm.cameras.f32", np.float32).reshape(-1, 2 + 64)
d["camkat"] = rd("glm.camkat.f32", np.float32).reshape(-1, 13 + 64)
np.savez_compressed(out, **d)
print("wrote", out, os.path.getsize(out), "bytes")
