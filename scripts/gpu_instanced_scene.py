"""Config-4 shape (BASELINE.json configs[3]): one mesh of ~6.3 k triangles (stand-in for teapot.obj, which lives in
the reference tree and cannot travel) instanced on an n x n grid, spacing 8 units on XZ, camera pulled back to
frame the grid; build time, traversal statistics, Mrays/s at 1920x1080, depth 8."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import pathtracer_amd as pa
from pathtracer_amd.api import PT_PROFILE_TRAVERSAL

W, H, S, D = 1920, 1080, int(os.environ.get("SPP", "4")), 8
rt = pa.HipRaytracing(0)
sc = pa.scenes.grid_mesh(56, 56, size=6.0, amplitude=0.8)
for n in [int(x) for x in os.environ.get("GRIDS", "8,32,64").split(",")]:
    rng = np.random.default_rng(1)
    xf = np.zeros((n * n, 12), dtype=np.float32)
    k = 0
    for i in range(n):
        for j in range(n):
            a = rng.uniform(0, 2 * math.pi)
            xf[k] = [math.cos(a), 0, math.sin(a), 8.0 * (i - (n - 1) / 2), 0, 1, 0, 0, -math.sin(a), 0, math.cos(a), 8.0 * (j - (n - 1) / 2)]
            k += 1
    ext = 8.0 * n
    view = pa.look_at((0.0, 0.45 * ext, 0.8 * ext), (0.0, 0.0, 0.0), (0, 1, 0))
    proj = pa.perspective(math.radians(45.0), W / H, 0.1, 10.0 * ext)
    proj[5] *= -1
    rt.setInstances(xf)
    t = time.perf_counter(); rt.buildBlas(sc); rt.synchronize(); tb = time.perf_counter() - t
    st = rt.stats()
    rt.resize(W, H); rt.setCamera(view, proj); rt.setParams(spp_per_frame=S, max_depth=D, tmax=1e6, max_frames=0)
    rt.setProfiling(PT_PROFILE_TRAVERSAL); rt.resetStats(); rt.raytrace(1); ts = rt.stats()
    rays = max(1, ts.rays_total)
    rt.setProfiling(0); rt.resetFrame(); rt.raytrace(1); rt.synchronize(); rt.resetStats(); rt.resetFrame()
    t = time.perf_counter(); rt.raytrace(4); rt.synchronize(); dt = time.perf_counter() - t
    s2 = rt.stats()
    print(f"grid {n}x{n} x {sc.n_tris} tris = {n*n*sc.n_tris/1e6:.2f} M instanced tris: build {tb*1e3:.1f} ms wall, nodes {st.bvh_nodes} | "
          f"nodes/ray {ts.trav_nodes/rays:.1f} tris/ray {ts.trav_tris/rays:.1f} leaf+instance visits/ray {ts.trav_leaf_visits/rays:.2f} "
          f"p_hit {ts.trav_hits/rays:.3f} rays/frame {rays/1e6:.1f}M | {s2.rays_total/dt/1e6:.0f} Mrays/s", flush=True)
