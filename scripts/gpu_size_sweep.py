"""k_bounce time vs queue size (fixed-cost hunt)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import pathtracer_amd as pa
from pathtracer_amd.api import PT_PROFILE_KERNELS
sc = pa.scenes.cornell_box()
rt = pa.HipRaytracing(0)
rt.buildBlas(sc)
for (W, H, S, D) in [(64, 64, 1, 8), (256, 256, 1, 8), (512, 512, 4, 8), (1024, 1024, 4, 8), (1920, 1080, 4, 8), (1920, 1080, 16, 1), (1920, 1080, 16, 2), (1920, 1080, 16, 8), (1920, 1080, 32, 8)]:
    view, proj = pa.reference_camera(W, H)
    rt.resize(W, H); rt.setCamera(view, proj); rt.setParams(spp_per_frame=S, max_depth=D, max_frames=0)
    rt.setProfiling(PT_PROFILE_KERNELS); rt.raytrace(2); rt.synchronize(); rt.resetStats(); rt.resetFrame()
    t = time.perf_counter(); rt.raytrace(8); rt.synchronize(); dt = time.perf_counter() - t
    st = rt.stats()
    print(f"{W}x{H} spp {S} depth {D}: paths {W*H*S/1e6:.2f} M rays/frame {st.rays_total/8/1e6:.2f} M | frame {dt/8*1e3:.3f} ms | "
          f"bounce mean {st.extend_ms_total/st.extend_launches*1e3:.1f} us | {st.rays_total/dt/1e6:.0f} Mrays/s", flush=True)
