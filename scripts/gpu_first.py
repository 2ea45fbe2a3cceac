"""First end-to-end GPU check: RNG, Morton, BVH vs brute force, image parity vs the CPU oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle as O
import pathtracer_amd as pa
from pathtracer_amd.api import PT_ACCEL_BRUTE_FORCE, PT_ACCEL_BVH4

sc = pa.scenes.cornell_box()
osc = O.OracleScene(sc.vertices, sc.indices, sc.materials, sc.mat_indices)
rt = pa.HipRaytracing(0)
print("ctx ok", flush=True)
v0 = np.arange(1000, dtype=np.uint32) * 7919
v1 = np.arange(1000, dtype=np.uint32) % 13
tea, lcg, unit = rt.debugRng(v0, v1, 8)
ok = all(int(tea[i]) == O.tea(int(v0[i]), int(v1[i])) for i in range(1000))
print("tea parity", ok, flush=True)
u0, _ = O.rnd_unit_vector(int(tea[5]))
print("unit", unit[5], u0, np.array_equal(unit[5].view(np.uint32), u0.view(np.uint32)), flush=True)

W = H = 256
view, proj = pa.reference_camera(W, H)
for flags, name in ((PT_ACCEL_BRUTE_FORCE, "brute"), (PT_ACCEL_BVH4, "bvh4")):
    rt.buildBlas(sc, flags)
    st = rt.stats()
    print(name, "nodes", st.bvh_nodes, "depth", st.bvh_depth, "tris", st.bvh_tris, "build ms", st.build_ms, flush=True)
    rt.resize(W, H)
    rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=8, max_depth=8, seed=0, max_frames=1000)
    hits = rt.debugPrimaryHits()
    oh = osc.primary_hits(view, proj, W, H)
    print(name, "primary prim eq", np.array_equal(hits["prim"], oh["prim"]), "t eq",
          np.array_equal(hits["t"].view(np.uint32), oh["t"].view(np.uint32)),
          "uv eq", np.array_equal(hits["u"].view(np.uint32), oh["u"].view(np.uint32)), flush=True)
    rt.resetFrame()
    t = time.time()
    rt.raytrace(3)
    img = rt.readImage()
    dt = time.time() - t
    oimg, orays = osc.render(view, proj, W, H, frames=3, spp_per_frame=8, max_depth=8, seed=0)
    st = rt.stats()
    print(name, "rays gpu", st.rays_total, "oracle", orays, "sec", dt, flush=True)
    print(name, "image bit-exact", np.array_equal(img.view(np.uint32), oimg.view(np.uint32)), "max abs diff",
          np.abs(img - oimg).max(), "mean", img[..., :3].mean(), oimg[..., :3].mean(), flush=True)
    rt.resetStats()
codes, order, cmin, cmax = rt.debugMorton()
oc, oo, omin, omax = osc.morton()
print("morton eq", np.array_equal(codes, oc), "order eq", np.array_equal(order, oo), cmin, omin, flush=True)
nodes, tris = rt.debugReadBvh()
print("nodes", len(nodes), "tris", len(tris), sorted(tris["prim"].tolist()) == list(range(sc.n_tris)))
print(nodes[0])
