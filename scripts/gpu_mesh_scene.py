"""Mesh-like scaling probe: wavy height field (shared vertices) of 2*n*n triangles under the reference camera pose
scaled to the mesh; both builders; build time, tree/traversal statistics, Mrays/s at 1920x1080, depth 8."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import pathtracer_amd as pa
from pathtracer_amd.api import PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC, PT_PROFILE_TRAVERSAL

W, H, S, D = 1920, 1080, int(os.environ.get("SPP", "4")), 8
rt = pa.HipRaytracing(0)
for n in [int(x) for x in os.environ.get("SIZES", "100,400,1000").split(",")]:
    sc = pa.scenes.grid_mesh(n, n, size=2.0, amplitude=0.15)
    view = pa.look_at((0.0, 1.1, 2.2), (0.0, 0.1, 0.0), (0, 1, 0))
    proj = pa.perspective(math.radians(45.0), W / H, 0.1, 10.0)
    proj[5] *= -1
    rt.uploadScene(sc)
    for mode, name in ((PT_ACCEL_BVH4, "lbvh"), (PT_ACCEL_BVH4_PLOC, "ploc")):
        rt.buildBlas(None, mode); st = rt.stats()
        rt.resize(W, H); rt.setCamera(view, proj); rt.setParams(spp_per_frame=S, max_depth=D, max_frames=0)
        rt.setProfiling(PT_PROFILE_TRAVERSAL); rt.resetStats(); rt.raytrace(1); ts = rt.stats()
        rays = max(1, ts.rays_total)
        rt.setProfiling(0); rt.resetFrame(); rt.raytrace(1); rt.synchronize(); rt.resetStats(); rt.resetFrame()
        t = time.perf_counter(); rt.raytrace(4); rt.synchronize(); dt = time.perf_counter() - t
        s2 = rt.stats()
        print(f"grid {n}x{n} ({sc.n_tris} tris) {name}: build {st.build_ms:.1f} ms, nodes {st.bvh_nodes} depth {st.bvh_depth} | "
              f"nodes/ray {ts.trav_nodes/rays:.1f} tris/ray {ts.trav_tris/rays:.1f} p_hit {ts.trav_hits/rays:.3f} "
              f"rays/frame {rays/1e6:.1f}M | {s2.rays_total/dt/1e6:.0f} Mrays/s", flush=True)
