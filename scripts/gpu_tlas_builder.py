"""Top-level builder A/B (PT_TLAS_BUILDER=lbvh|ploc) on instance layouts that are NOT a regular grid: teapot.obj x N instances at
random positions in a square (density of the 64 x 64 grid), random rotation about a random axis, random scale 0.5 .. 2.
usage: gpu_tlas_builder.py [N=4096]   (run once per setting of PT_TLAS_BUILDER; prints Mrays/s, rays and an image CRC)"""
import math, os, sys, time, zlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import bench
import pathtracer_amd as pa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "random"
sc = pa.scenes.with_light_quad(bench.reference_mesh(pa, "teapot"), [[-3.0, 5.0, -3.0], [3.0, 5.0, -3.0], [3.0, 5.0, 3.0], [-3.0, 5.0, 3.0]], (0, -1, 0))
rng = np.random.default_rng(7)
ext = 8.0 * math.sqrt(n)
xf = np.zeros((n, 12), dtype=np.float32)
for k in range(n):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    a = rng.uniform(0, 2 * math.pi); s = rng.uniform(0.5, 2.0)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = (np.eye(3) + math.sin(a) * K + (1 - math.cos(a)) * K @ K) * s
    if mode == "clustered":  # a few dense clusters and empty space between them
        c = rng.integers(0, 12); centre = np.array([math.cos(c) * ext * 0.35, 0.0, math.sin(c * 1.7) * ext * 0.35])
        t = centre + rng.normal(size=3) * np.array([ext * 0.04, 6.0, ext * 0.04])
    else:
        t = np.array([rng.uniform(-ext / 2, ext / 2), rng.uniform(0.0, 30.0), rng.uniform(-ext / 2, ext / 2)])
    xf[k] = np.concatenate([R, t[:, None]], axis=1).reshape(12)
W, H = 1920, 1080
view = pa.look_at((0.0, 0.45 * ext, 0.8 * ext), (0.0, 0.0, 0.0), (0, 1, 0))
proj = pa.perspective(math.radians(45.0), W / H, 0.1, 10.0 * ext); proj[5] *= -1
rt = pa.HipRaytracing(0)
rt.setInstances(xf); rt.buildBlas(sc); rt.buildTlas()
rt.resize(W, H); rt.setCamera(view, proj); rt.setParams(spp_per_frame=16, max_depth=8, tmin=1e-3, tmax=1e6, seed=0, max_frames=0)
rt.raytrace(2); rt.synchronize(); crc = zlib.crc32(rt.readImage().tobytes()); r2 = rt.stats().rays_total
best = 1e9
for rep in range(3):
    rt.resetStats(); rt.resetFrame()
    t0 = time.perf_counter(); rt.raytrace(8); rt.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"{mode:9s} n {n} tlas {os.environ.get('PT_TLAS_BUILDER', 'same as blas'):12s} {rt.stats().rays_total / best / 1e6:9.1f} Mrays/s  rays(2) {r2}  crc {crc:08x}", flush=True)
