"""Generates tests/golden/oracle_kat.npz from the CPU oracle (oracle/pt_oracle.c).

The reference holds no known-answer vectors for its shader path, so these
vectors pin the ORACLE against silent change; the reference-pinned fixture is ref_seam.npz.
Run from the repo root: python tests/golden/make_oracle_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
import pathtracer_amd as pa  # noqa: E402  (host seam only: scene table + camera, no GPU)

d = {}
v0 = np.array([0, 1, 2, 3, 12345, 0xFFFFFFFF, 262143, 2073599, 8294399, 0xDEADBEEF], dtype=np.uint32)
v1 = np.array([0, 0, 1, 7, 999, 0xFFFFFFFF, 5, 63, 1023, 0x5EED], dtype=np.uint32)
d["tea_v0"], d["tea_v1"] = v0, v1
d["tea_out"] = np.array([O.tea(int(a), int(b)) for a, b in zip(v0, v1)], dtype=np.uint32)
d["lcg_state0"] = d["tea_out"]
d["lcg_out"] = np.array([O.lcg_stream(int(s), 16)[0] for s in d["tea_out"]], dtype=np.uint32)
us = np.array([0, 1, 2 ** 22 - 1, 2 ** 22, 2 ** 22 + 1, 2 ** 23, 3 * 2 ** 22, 2 ** 24 - 1, 1234567, 9876543, 2097152, 6291456,
               14680064, 16000000], dtype=np.uint32)
d["sincos_k"] = us
d["sincos_out"] = np.array([O.sincos_2pi(np.float32(k) / np.float32(16777216.0)) for k in us], dtype=np.float32)
d["unit_out"] = np.array([O.rnd_unit_vector(int(s))[0] for s in d["tea_out"]], dtype=np.float32)

scene = pa.scenes.cornell_box()
osc = O.OracleScene(scene.vertices, scene.indices, scene.materials, scene.mat_indices, use_bvh=False)
W, H = 64, 48
view, proj = O.reference_camera(W, H)
hits = osc.primary_hits(view, proj, W, H)
d["cornell_primary_64x48_prim"] = hits["prim"]
d["cornell_primary_64x48_tuv"] = np.stack([hits["t"], hits["u"], hits["v"]], axis=1)
img, rays = osc.render(view, proj, W, H, frames=3, spp_per_frame=4, max_depth=8, seed=0)
d["cornell_image_64x48_f3_s4_d8"] = img
d["cornell_image_rays"] = np.array([rays], dtype=np.uint64)
img2, rays2 = osc.render(view, proj, W, H, frames=2, spp_per_frame=10, max_depth=20, seed=7)
d["cornell_image_64x48_f2_s10_d20_seed7"] = img2
codes, order, cmin, cmax = osc.morton()
d["cornell_morton_codes"], d["cornell_morton_order"] = codes, order
d["cornell_cmin"], d["cornell_cmax"] = cmin, cmax
out = os.path.join(ROOT, "tests", "golden", "oracle_kat.npz")
np.savez_compressed(out, **d)
print("wrote", out, os.path.getsize(out))
