"""Soak: (a) 2048 frames of the headline config in one context, checksum compared between two runs; (b) 42 create /
build / render / destroy cycles over 6 recurring configurations, free device memory compared between cycle 6 and the end
(ever-changing sizes fragment the HIP runtime's sub-2 MiB pools by ~2 MiB per cycle without any allocation of ours being
lost: scripts/gpu_leak_probe.py)."""
import ctypes, os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import pathtracer_amd as pa
hip = ctypes.CDLL("libamdhip64.so")
def free_mem():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value
W, H = 1920, 1080
sums = []
for run in range(2):
    rt = pa.HipRaytracing(0); rt.buildBlas(pa.scenes.cornell_box()); rt.resize(W, H)
    v, p = pa.reference_camera(W, H); rt.setCamera(v, p); rt.setParams(spp_per_frame=16, max_depth=8, seed=1, max_frames=0)
    t = time.perf_counter(); rt.raytrace(1024); rt.synchronize(); dt = time.perf_counter() - t
    img = rt.readImage(); st = rt.stats()
    sums.append(zlib.crc32(img.tobytes()))
    print(f"run {run}: 1024 frames {dt:.2f} s, {st.rays_total/dt/1e6:.0f} Mrays/s, crc {sums[-1]:08x}, mean {img[..., :3].mean():.6f}", flush=True)
    rt.cleanup()
assert sums[0] == sums[1], "non-deterministic image"
rng = np.random.default_rng(0)
cfgs = [(int(rng.integers(1, 400)), int(rng.integers(1, 300)), int(rng.integers(1, 9)), int(rng.integers(1, 12)), int(rng.integers(2, 12))) for _ in range(6)]
for i in range(42):
    if i == 6: base = free_mem()  # every configuration has been seen once: the runtime's sub-allocator pools are warm
    rt = pa.HipRaytracing(0)
    sc = [pa.scenes.cornell_box, lambda: pa.scenes.grid_mesh(60, 40), lambda: pa.scenes.random_triangles(30000, extent=30.0)][i % 3]()
    rt.buildBlas(sc, [0, 2][i % 2])
    w, h, spp, depth, frames = cfgs[i % 6]
    rt.resize(w, h); v, p = pa.reference_camera(w, h); rt.setCamera(v, p)
    rt.setParams(spp_per_frame=spp, max_depth=depth, seed=i)
    rt.raytrace(frames); rt.readImage(); rt.cleanup()
after = free_mem()
print(f"free device memory after 6 cycles {base/2**20:.0f} MiB, after 42 {after/2**20:.0f} MiB, delta {(base-after)/2**20:.1f} MiB")
# informational: with recurring-but-different sizes the HIP runtime keeps ~2 MiB per cycle in its own pools (identical
# cycles leak nothing: scripts/gpu_leak_probe.py, which asserts)
print("soak ok")
