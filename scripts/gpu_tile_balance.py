"""Load balance of the image-tile split (BASELINE configs[3] / configs[4]): renders each of the G column strips of
multigpu.tile_schedule ALONE on this GPU (what rank r of G would do) and prints max / mean of the strip times -- the factor by
which the slowest rank holds the others up.  With INTERLEAVE=B the same for the interleaved schedule (blocks of B columns dealt
round-robin).  usage: gpu_tile_balance.py teapot|soup:2e7|viking|cornell [G=8]   env: SPP (16), FRAMES (2)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
import pathtracer_amd as pa
from pathtracer_amd import multigpu

scene = sys.argv[1]
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
argv = ["--scene", scene.split(":")[0]]
if scene.startswith("viking"): argv += ["--width", "3840", "--height", "2160"]
if ":" in scene: argv += ["--tris", scene.split(":")[1]]
a = bench.parse(argv)
a.spp_per_frame = int(os.environ.get("SPP", "16"))
F = int(os.environ.get("FRAMES", "2"))
B = int(os.environ.get("INTERLEAVE", "0"))
sc, view, proj, tmax, name = bench.build_scene(a, pa)
rt = pa.HipRaytracing(0)
if a.instances is not None: rt.setInstances(a.instances)
rt.buildBlas(sc, pa.api.PT_ACCEL_BVH4 if scene.startswith("soup") else pa.api.PT_ACCEL_BVH4_PLOC); rt.buildTlas()
if a.texture is not None: rt.setTexture(a.texture)
W, H = a.width, a.height
rt.resize(W, H); rt.setCamera(view, proj)
rt.setParams(spp_per_frame=a.spp_per_frame, max_depth=a.depth, tmin=1e-3, tmax=tmax, seed=0, max_frames=0)


def timed():
    rt.resetFrame(); rt.raytrace(1); rt.synchronize()
    rt.resetFrame(); rt.resetStats()
    t = time.perf_counter(); rt.raytrace(F); rt.synchronize(); dt = time.perf_counter() - t
    return dt / F * 1e3, rt.stats().rays_total / F

whole_ms, whole_rays = timed()
ms, rays = [], []
for r in range(G):
    if B:
        rt.setTileInterleaved(r, G, B)
    else:
        rt.setTile(*multigpu.tile_schedule(r, G, W, H))
    m, n = timed()
    ms.append(m); rays.append(n)
mean = sum(ms) / G
print(f"{name} {W}x{H} x {a.spp_per_frame} spp, {G} {'interleaved %d-column blocks' % B if B else 'column strips'}: whole frame {whole_ms:.2f} ms ({whole_rays/1e6:.1f} M rays)")
print("  strip ms   : " + " ".join(f"{m:7.2f}" for m in ms))
print("  strip Mrays: " + " ".join(f"{n/1e6:7.1f}" for n in rays))
print(f"  max/mean = {max(ms)/mean:.3f}   sum of strips / whole = {sum(ms)/whole_ms:.3f}   ideal speed-up {G}, with this split {whole_ms/max(ms):.2f}")
