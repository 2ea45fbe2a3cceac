"""One-off differential campaign for two-level scenes: N random scenes of 1 .. 4 Models (tests/test_gpu_entry.py::_random_scene,
with random texture coordinates) shown by 1 .. 12 random instances (rotations, non-uniform and mirroring scales, overlapping
boxes, several instances of one Model), with or without per-Model diffuse maps, both top-level builders -- every image and
ray count against the CPU oracle, bit for bit.
usage: python scripts/fuzz/gpu_instance_fuzz.py [first_seed] [count]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as O
import pathtracer_amd as pa
from pathtracer_amd.api import PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC
from test_gpu_entry import _random_scene
from test_instancing import xform

O.build()
first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 200
rt = pa.HipRaytracing(0)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    n_models = int(rng.integers(1, 5))
    models = []
    for m in range(n_models):
        sc = _random_scene(pa, 100 * seed + m)
        v = sc.vertices.copy()
        v[:, 9:11] = rng.uniform(-2.0, 3.0, (v.shape[0], 2)).astype(np.float32)
        models.append(pa.Scene(v, sc.indices, sc.materials, sc.mat_indices))
    n_inst = int(rng.integers(1, 13))
    custom = rng.integers(0, n_models, n_inst).astype(np.uint32)
    xf = np.stack([xform(rot_y_deg=float(rng.uniform(-180, 180)), rot_x_deg=float(rng.uniform(-40, 40)),
                         scale=tuple(float(s) * (-1.0 if rng.uniform() < 0.15 else 1.0) for s in rng.uniform(0.4, 1.6, 3)),
                         translate=tuple(float(t) for t in rng.uniform(-2.5, 2.5, 3))) for _ in range(n_inst)])
    textures = None
    if seed % 3:
        textures = [pa.scenes.random_texture(int(rng.integers(1, 40)), int(rng.integers(1, 40)), seed=int(rng.integers(0, 1 << 30)))
                    if rng.uniform() < 0.7 else None for _ in range(n_models)]
        if all(t is None for t in textures):
            textures = None
    W, H = 40, 30
    view = pa.look_at(tuple(rng.uniform(-5.0, 5.0, 3)), tuple(rng.uniform(-0.5, 0.5, 3)), (0, 1, 0))
    proj = pa.perspective(np.radians(float(rng.uniform(30, 90))), W / H, 0.1, 100.0); proj[5] *= -1
    kw = {"model_textures": textures} if textures is not None else {}
    osc = O.OracleScene.from_models(models, xf, custom, **kw)
    want, rays = osc.render(view, proj, W, H, frames=2, spp_per_frame=2, max_depth=5, seed=seed)
    for name, flags, tlas in (("ploc", PT_ACCEL_BVH4_PLOC, "ploc"), ("lbvh", PT_ACCEL_BVH4 | ((seed % 3) << 8), "lbvh"), ("auto", PT_ACCEL_BVH4_PLOC, None)):
        if tlas:
            os.environ["PT_TLAS_BUILDER"] = tlas
        else:
            os.environ.pop("PT_TLAS_BUILDER", None)
        rt.uploadModels(models)
        rt.setInstances(xf, custom)
        rt.buildBlas(None, flags)
        rt.setTexture(None)
        if textures is not None:
            for m, t in enumerate(textures):
                if t is not None:
                    rt.setModelTexture(m, t)
        rt.resize(W, H); rt.setCamera(view, proj); rt.setParams(spp_per_frame=2, max_depth=5, seed=seed)
        rt.resetFrame(); rt.resetStats(); rt.raytrace(2)
        img, st = rt.readImage(), rt.stats()
        if st.rays_total != rays or not np.array_equal(bits(img), bits(want)):
            bad += 1
            print("MISMATCH seed", seed, name, n_models, n_inst, st.rays_total, rays, int((bits(img) != bits(want)).sum()), flush=True)
    if (seed - first) % 25 == 24:
        print("seeds", first, "..", seed, "done, mismatches", bad, flush=True)
os.environ.pop("PT_TLAS_BUILDER", None)
print("campaign done:", count, "scenes, mismatches", bad)
sys.exit(1 if bad else 0)
