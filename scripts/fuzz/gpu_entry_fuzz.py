"""One-off differential campaign: N random small scenes (tests/test_gpu_entry.py::_random_scene) rendered through the default
persistent pipeline (entry node: flat or tree), PT_NO_FLAT and GPU brute force, each against the CPU oracle, bit for bit.
usage: python scripts/fuzz/gpu_entry_fuzz.py [first_seed] [count]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as O
import pathtracer_amd as pa
from pathtracer_amd.api import PT_ACCEL_BRUTE_FORCE, PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC, PT_PIPELINE_PERSISTENT, PT_PIPELINE_SPLIT
from test_gpu_entry import _random_scene, _render

O.build()
first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 200
rt = pa.HipRaytracing(0)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
bad = 0
for seed in range(first, first + count):
    sc = _random_scene(pa, seed)
    osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
    W, H = 40, 30
    rng = np.random.default_rng(5000 + seed)
    view = pa.look_at(tuple(rng.uniform(-2.5, 2.5, 3)), tuple(rng.uniform(-0.3, 0.3, 3)), (0, 1, 0))
    proj = pa.perspective(np.radians(float(rng.uniform(30, 90))), W / H, 0.1, 100.0); proj[5] *= -1
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=2, max_depth=5, seed=seed)
    for name, flags, env, pipe in (("default", [PT_ACCEL_BVH4_PLOC, PT_ACCEL_BVH4][seed & 1] | ((seed % 3) << 8), None, PT_PIPELINE_PERSISTENT),
                                   ("tree", PT_ACCEL_BVH4_PLOC | ((seed % 3) << 8), {"PT_NO_FLAT": "1"}, PT_PIPELINE_PERSISTENT),
                                   ("brute", PT_ACCEL_BRUTE_FORCE, None, PT_PIPELINE_SPLIT)):
        rt.setPipeline(pipe)
        img, st = _render(rt, pa, sc, view, proj, W, H, flags, env, frames=2, spp=2, depth=5, seed=seed, profile=False)[0]
        if st.rays_total != rays or not np.array_equal(bits(img), bits(want)):
            bad += 1
            print("MISMATCH seed", seed, name, sc.n_tris, st.rays_total, rays, int((bits(img) != bits(want)).sum()), flush=True)
    if (seed - first) % 50 == 49:
        print("seeds", first, "..", seed, "done, mismatches", bad, flush=True)
print("campaign done:", count, "scenes, mismatches", bad)
sys.exit(1 if bad else 0)
