"""Config-3-style throughput probe: Cornell box with a 1024x1024 diffuse map (and a 4 k-triangle height field with one),
3840x2160, 16 spp per frame, depth 8.  env PT_NO_SHADE_RECORDS=1 takes the gather path of raytrace.rchit instead of the
per-primitive shading records (same image; compare the Mrays/s)."""
import math, os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import pathtracer_amd as pa

W, H, S, D = 3840, 2160, 16, 8
frames = int(os.environ.get("FRAMES", "8"))
tex = pa.scenes.random_texture(1024, 1024, seed=5)
for name in ("cornell", "grid"):
    if name == "cornell":
        sc = pa.scenes.textured_cornell()
        view, proj = pa.reference_camera(W, H)
    else:
        sc = pa.scenes.grid_mesh(45, 45, size=2.0, amplitude=0.15)  # ~4 k triangles, the size of viking_room.obj
        v = sc.vertices.copy()
        v[:, 9] = v[:, 0] * np.float32(0.5) + np.float32(0.5)
        v[:, 10] = v[:, 2] * np.float32(0.5) + np.float32(0.5)
        sc = pa.Scene(v, sc.indices, sc.materials, sc.mat_indices)
        view = pa.look_at((0.0, 1.1, 2.6), (0.0, 0.1, 0.0), (0, 1, 0))
        proj = pa.perspective(math.radians(45.0), W / H, 0.1, 100.0)
        proj[5] *= -1
    for textured in (False, True):
        rt = pa.HipRaytracing(0)
        rt.buildBlas(sc)
        if textured:
            rt.setTexture(tex)
        rt.resize(W, H); rt.setCamera(view, proj); rt.setParams(spp_per_frame=S, max_depth=D, max_frames=0)
        rt.raytrace(8); rt.synchronize(); rt.resetStats(); rt.resetFrame()
        t = time.perf_counter(); rt.raytrace(frames); rt.synchronize(); dt = time.perf_counter() - t
        st = rt.stats()
        crc = zlib.crc32(rt.readImage().tobytes())
        print(f"{name} tris {sc.n_tris} textured {int(textured)} records {int(not os.environ.get('PT_NO_SHADE_RECORDS'))} "
              f"rays {st.rays_total} ms/frame {dt/frames*1e3:.3f} Mrays/s {st.rays_total/dt/1e6:.0f} crc {crc:08x}", flush=True)
        rt.cleanup()
