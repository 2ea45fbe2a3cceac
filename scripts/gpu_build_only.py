"""Build-only probe for rocprofv3: N random triangles (config-5 generator), one LBVH and one PLOC build."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import pathtracer_amd as pa
from pathtracer_amd.api import PT_ACCEL_BVH4, PT_ACCEL_BVH4_PLOC

n = int(float(os.environ.get("N", "4e6")))
extent = 1000.0 * (n / 1e8) ** (1.0 / 3.0)
rt = pa.HipRaytracing(0)
sc = pa.scenes.random_triangles(n, extent=extent, edge=1.0)
rt.uploadScene(sc)
for mode, name in ((PT_ACCEL_BVH4, "lbvh"), (PT_ACCEL_BVH4_PLOC, "ploc")):
    if name not in os.environ.get("BUILDERS", "lbvh,ploc"): continue
    for rep in range(int(os.environ.get("REPS", "2"))):
        t = time.perf_counter(); rt.buildBlas(None, mode); rt.synchronize(); tw = time.perf_counter() - t
        st = rt.stats()
        print(f"n {n} {name} rep {rep}: build {st.build_ms:.2f} ms dev, {tw*1e3:.1f} ms wall = {n/st.build_ms/1e3:.1f} Mtris/s, nodes {st.bvh_nodes} depth {st.bvh_depth}", flush=True)
