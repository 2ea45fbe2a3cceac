"""Config-5-style scaling probe: N random triangles (SURVEY 8d config 5 generator), both builders.
Prints build time / Mtris/s, tree statistics, traversal statistics and Mrays/s at 1920x1080, depth 8."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import pathtracer_amd as pa
from pathtracer_amd.api import PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC, PT_PROFILE_KERNELS, PT_PROFILE_TRAVERSAL

W, H, S, D = 1920, 1080, int(os.environ.get("SPP", "4")), 8
rt = pa.HipRaytracing(0)
for n in [int(float(x)) for x in os.environ.get("SIZES", "1e5,1e6,4e6").split(",")]:
    extent = 1000.0 * (n / 1e8) ** (1.0 / 3.0)
    t = time.perf_counter()
    sc = pa.scenes.random_triangles(n, extent=extent, edge=1.0)
    tgen = time.perf_counter() - t
    c = extent * 0.5
    view = pa.look_at((c, c * 1.2, c + extent * 1.6), (c, c * 0.9, c), (0, 1, 0))
    proj = pa.perspective(math.radians(45.0), W / H, 0.1, extent * 10)
    proj[5] *= -1  # main.cpp:1475 flips Y for Vulkan clip space
    t = time.perf_counter(); rt.uploadScene(sc); tup = time.perf_counter() - t
    for mode, name in ((PT_ACCEL_BVH4, "lbvh"), (PT_ACCEL_BVH4_PLOC, "ploc")):
        if name not in os.environ.get("BUILDERS", "lbvh,ploc"): continue
        t = time.perf_counter(); rt.buildBlas(None, mode); rt.synchronize(); twall = time.perf_counter() - t
        st = rt.stats()
        rt.resize(W, H); rt.setCamera(view, proj); rt.setParams(spp_per_frame=S, max_depth=D, tmax=1e6, max_frames=0)
        rt.setProfiling(PT_PROFILE_TRAVERSAL); rt.resetStats(); rt.raytrace(1); ts = rt.stats()
        rays = max(1, ts.rays_total)
        rt.setProfiling(0); rt.resetFrame(); rt.raytrace(1); rt.synchronize(); rt.resetStats(); rt.resetFrame()
        t = time.perf_counter(); rt.raytrace(4); rt.synchronize(); dt = time.perf_counter() - t
        s2 = rt.stats()
        print(f"n {n:>9d} {name}: gen {tgen:.1f}s upload {tup*1e3:.0f} ms | build {st.build_ms:.1f} ms dev ({twall*1e3:.0f} wall) = "
              f"{n/st.build_ms/1e3:.1f} Mtris/s, nodes {st.bvh_nodes} depth {st.bvh_depth} | nodes/ray {ts.trav_nodes/rays:.1f} "
              f"tris/ray {ts.trav_tris/rays:.1f} p_hit {ts.trav_hits/rays:.3f} rays/frame {rays/1e6:.1f}M | {s2.rays_total/dt/1e6:.0f} Mrays/s", flush=True)
