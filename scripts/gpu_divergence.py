"""Wave-level divergence statistics of k_paths per scene (instrumented kernel): lane utilisation of node / leaf steps."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import pathtracer_amd as pa
from pathtracer_amd.api import PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC, PT_PROFILE_TRAVERSAL
W, H = 1920, 1080
rt = pa.HipRaytracing(0)
def probe(name, sc, view, proj, builder, spp=4, tmax=1e6):
    rt.buildBlas(sc, builder); rt.resize(W, H); rt.setCamera(view, proj)
    rt.setParams(spp_per_frame=spp, max_depth=8, tmax=tmax, max_frames=0)
    rt.setProfiling(PT_PROFILE_TRAVERSAL); rt.resetStats(); rt.resetFrame(); rt.raytrace(2); ts = rt.stats(); rt.setProfiling(0)
    r = max(1, ts.rays_total)
    print(f"{name:28s} nodes/ray {ts.trav_nodes/r:6.2f} leafvisits/ray {ts.trav_leaf_visits/r:5.2f} tris/ray {ts.trav_tris/r:5.2f} | "
          f"node steps/64rays {64*ts.trav_node_steps/r:6.2f} util {ts.trav_nodes/(64*ts.trav_node_steps):.3f} | "
          f"leaf steps/64rays {64*ts.trav_leaf_steps/r:6.2f} util {ts.trav_leaf_visits/(64*ts.trav_leaf_steps):.3f}", flush=True)
v, p = pa.reference_camera(W, H)
probe("cornell ploc", pa.scenes.cornell_box(), v, p, PT_ACCEL_BVH4_PLOC, spp=16, tmax=1e4)
g = 1000
view = pa.look_at((0.0, 1.1, 2.6), (0.0, 0.1, 0.0), (0, 1, 0)); proj = pa.perspective(math.radians(45.0), W / H, 0.1, 100.0); proj[5] *= -1
sc = pa.scenes.grid_mesh(g, g, size=2.0, amplitude=0.15)
probe("grid 2M lbvh", sc, view, proj, PT_ACCEL_BVH4); probe("grid 2M ploc", sc, view, proj, PT_ACCEL_BVH4_PLOC)
n = int(4e6); extent = 1000.0 * (n / 1e8) ** (1.0 / 3.0); c = extent * 0.5
view = pa.look_at((c, c * 1.2, c + extent * 1.6), (c, c * 0.9, c), (0, 1, 0)); proj = pa.perspective(math.radians(45.0), W / H, 0.1, extent * 10); proj[5] *= -1
sc = pa.scenes.random_triangles(n, extent=extent, edge=1.0)
probe("soup 4M lbvh", sc, view, proj, PT_ACCEL_BVH4); probe("soup 4M ploc", sc, view, proj, PT_ACCEL_BVH4_PLOC)
