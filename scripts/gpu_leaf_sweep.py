"""Leaf-size sweep on the headline workload: traversal statistics + per-kernel time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import pathtracer_amd as pa
from pathtracer_amd.api import PT_PROFILE_KERNELS, PT_PROFILE_TRAVERSAL
W, H, S, D = 1920, 1080, 16, 8
sc = pa.scenes.cornell_box()
view, proj = pa.reference_camera(W, H)
rt = pa.HipRaytracing(0)
for pipe in [int(x) for x in os.environ.get('PIPES', '0,1').split(',')]:
  rt.setPipeline(pipe)
  for mode, leaf in [(m, l) for m in [int(x) for x in os.environ.get("MODES", "0,2").split(",")]
                     for l in [int(x) for x in os.environ.get("LEAVES", "1,2,4").split(",")]]:
      rt.buildBlas(sc, mode | (leaf << 8))
      rt.resize(W, H); rt.setCamera(view, proj); rt.setParams(spp_per_frame=S, max_depth=D, max_frames=0)
      rt.setProfiling(PT_PROFILE_TRAVERSAL); rt.resetStats(); rt.raytrace(2); st = rt.stats()
      nodes, tris = st.trav_nodes / st.rays_total, st.trav_tris / st.rays_total
      rt.setProfiling(PT_PROFILE_KERNELS); rt.resetFrame(); rt.resetStats()
      rt.raytrace(2); rt.synchronize(); rt.resetStats(); rt.resetFrame()
      t = time.perf_counter(); rt.raytrace(16); rt.synchronize(); dt = time.perf_counter() - t
      st = rt.stats()
      print(f"pipe {pipe} mode {mode} leaf {leaf}: bvh nodes {st.bvh_nodes} depth {st.bvh_depth} | nodes/ray {nodes:.2f} tris/ray {tris:.2f} | "
            f"{st.rays_total/dt/1e6:.0f} Mrays/s  extend {st.extend_ms_total/st.extend_launches:.3f} ms shade {st.shade_ms_total/st.shade_launches:.3f} ms", flush=True)
