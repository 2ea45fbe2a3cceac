"""Config-5-style traversal probe for profilers: N random triangles, LBVH, a few frames at 1920x1080 depth 8.
Prints the ray count and the time so that PMC bytes can be turned into bytes per ray.  env: N, SPP, FRAMES, BUILDER."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import pathtracer_amd as pa
from pathtracer_amd.api import PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC

n = int(float(os.environ.get("N", "2e7")))
W, H, S, D = 1920, 1080, int(os.environ.get("SPP", "4")), 8
frames = int(os.environ.get("FRAMES", "4"))
extent = 1000.0 * (n / 1e8) ** (1.0 / 3.0)
rt = pa.HipRaytracing(0)
kind = os.environ.get("SCENE", "soup")
if kind == "soup":
    sc = pa.scenes.random_triangles(n, extent=extent, edge=1.0)
    c = extent * 0.5
    view = pa.look_at((c, c * 1.2, c + extent * 1.6), (c, c * 0.9, c), (0, 1, 0))
    proj = pa.perspective(math.radians(45.0), W / H, 0.1, extent * 10)
else:
    g = int(math.sqrt(n / 2))
    sc = pa.scenes.grid_mesh(g, g, size=2.0, amplitude=0.15)
    view = pa.look_at((0.0, 1.1, 2.6), (0.0, 0.1, 0.0), (0, 1, 0))
    proj = pa.perspective(math.radians(45.0), W / H, 0.1, 100.0)
proj[5] *= -1
rt.uploadScene(sc)
rt.buildBlas(None, (PT_ACCEL_BVH4_PLOC if os.environ.get("BUILDER") == "ploc" else PT_ACCEL_BVH4) | (int(os.environ.get("LEAF", "0")) << 8))
rt.resize(W, H); rt.setCamera(view, proj); rt.setParams(spp_per_frame=S, max_depth=D, tmax=1e6, max_frames=0)
rt.raytrace(1); rt.synchronize(); rt.resetStats(); rt.resetFrame()
t = time.perf_counter(); rt.raytrace(frames); rt.synchronize(); dt = time.perf_counter() - t
st = rt.stats()
print(f"scene {kind} tris {sc.n_tris} nodes {st.bvh_nodes} frames {frames} rays {st.rays_total} sec {dt:.4f} Mrays/s {st.rays_total/dt/1e6:.0f}", flush=True)
