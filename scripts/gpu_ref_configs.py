"""Throughput of BASELINE configs[2] and [3] on the reference's own meshes (tests/golden/ref_seam.npz: tinyobj's output for
models/viking_room.obj and models/teapot.obj, through pt_scene_from_tinyobj): viking_room 3840x2160 with a 1024^2 diffuse
map and one added light quad; teapot x 64x64 instances 1920x1080.  Depth 8.  env: SPP (4), FRAMES (4)."""
import math, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import pathtracer_amd as pa
from pathtracer_amd.api import PT_PROFILE_TRAVERSAL

S, F, D = int(os.environ.get("SPP", "4")), int(os.environ.get("FRAMES", "4")), 8
z = np.load(os.path.join(ROOT, "tests", "golden", "ref_seam.npz"))
mesh = lambda name: pa.scene_from_tinyobj(*(z[f"{name}_{k}"] for k in ("v", "vn", "vt", "idx", "mid", "mat")))


def run(tag, rt, W, H, view, proj, tmax=10000.0):
    rt.resize(W, H); rt.setCamera(view, proj); rt.setParams(spp_per_frame=S, max_depth=D, tmax=tmax, max_frames=0)
    rt.setProfiling(PT_PROFILE_TRAVERSAL); rt.resetStats(); rt.raytrace(1); ts = rt.stats(); rays = max(1, ts.rays_total)
    rt.setProfiling(0); rt.resetFrame(); rt.raytrace(1); rt.synchronize(); rt.resetStats(); rt.resetFrame()
    t = time.perf_counter(); rt.raytrace(F); rt.synchronize(); dt = time.perf_counter() - t
    s2 = rt.stats()
    print(f"{tag}: {W}x{H} x {S} spp, nodes {s2.bvh_nodes} build {s2.build_ms:.2f} ms | nodes/ray {ts.trav_nodes/rays:.2f} "
          f"tris/ray {ts.trav_tris/rays:.2f} p_hit {ts.trav_hits/rays:.3f} rays/frame {rays/1e6:.1f}M | "
          f"{s2.rays_total/dt/1e6:.0f} Mrays/s, {dt/F*1e3:.2f} ms/frame", flush=True)


rt = pa.HipRaytracing(0)
sc = pa.scenes.with_light_quad(mesh("viking_room"), [[-2.0, -2.0, 2.0], [2.0, -2.0, 2.0], [2.0, 2.0, 2.0], [-2.0, 2.0, 2.0]], (0, 0, -1))
W, H = 3840, 2160
view = pa.look_at((1.3, 1.3, 1.2), (0.05, 0.0, 0.3), (0.0, 0.0, 1.0)); proj = pa.perspective(math.radians(45.0), W / H, 0.1, 10.0); proj[5] *= -1
for textured in (False, True):
    rt.buildBlas(sc)
    rt.setTexture(np.random.default_rng(3).integers(0, 256, (1024, 1024, 4), dtype=np.uint8) if textured else None)
    run(f"configs[2] viking_room ({sc.n_tris} tris){' + 1024^2 map' if textured else ''}", rt, W, H, view, proj)
rt.setTexture(None)
sc = pa.scenes.with_light_quad(mesh("teapot"), [[-3.0, 5.0, -3.0], [3.0, 5.0, -3.0], [3.0, 5.0, 3.0], [-3.0, 5.0, 3.0]], (0, -1, 0))
n = 64; ext = 8.0 * n; W, H = 1920, 1080
view = pa.look_at((0.0, 0.45 * ext, 0.8 * ext), (0.0, 0.0, 0.0), (0, 1, 0)); proj = pa.perspective(math.radians(45.0), W / H, 0.1, 10.0 * ext); proj[5] *= -1
rt.uploadScene(sc); rt.setInstances(pa.scenes.instance_grid(n, 8.0)); rt.buildBlas()
run(f"configs[3] teapot ({sc.n_tris} tris) x {n}x{n} instances", rt, W, H, view, proj, tmax=1e6)
