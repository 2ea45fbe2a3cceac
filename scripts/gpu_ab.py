"""A/B of library variants on bench.py's scenes: every (variant, scene) in a process of its own, two interleaved rounds;
prints Mrays/s (best of 3 timed calls), rays per frame and a CRC of the image after 2 frames -- a variant that changes the
CRC or the ray count has changed the result, not just the speed.
usage: gpu_ab.py cornell,viking,teapot[,soup:2e6] default varA varB@ENV=1 ...     (variants: scripts/build_variant.sh)"""
import os, subprocess, sys
root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
scenes = sys.argv[1].split(",")
variants = sys.argv[2:]
code = r'''
import os, sys, time, zlib
sys.path.insert(0, %r)
import bench
import pathtracer_amd as pa
scene = os.environ["AB_SCENE"]
argv = ["--scene", scene.split(":")[0]]
if scene.startswith("viking"): argv += ["--width", "3840", "--height", "2160"]
if ":" in scene: argv += ["--tris", scene.split(":")[1]]
a = bench.parse(argv)
if os.environ.get("AB_SPP"): a.spp_per_frame = int(os.environ["AB_SPP"])
sc, view, proj, tmax, name = bench.build_scene(a, pa)
rt = pa.HipRaytracing(0)
if a.instances is not None: rt.setInstances(a.instances)
rt.buildBlas(sc, (pa.api.PT_ACCEL_BVH4 if scene.startswith("soup") else pa.api.PT_ACCEL_BVH4_PLOC) | int(os.environ.get("AB_ACCEL_FLAGS", "0"), 0)); rt.buildTlas()
if a.texture is not None: rt.setTexture(a.texture)
rt.resize(a.width, a.height); rt.setCamera(view, proj)
rt.setParams(spp_per_frame=a.spp_per_frame, max_depth=a.depth, tmin=1e-3, tmax=tmax, seed=0, max_frames=0)
rt.raytrace(2); rt.synchronize(); crc = zlib.crc32(rt.readImage().tobytes()); r2 = rt.stats().rays_total
F = int(os.environ.get("AB_FRAMES", "8"))
rt.resetStats(); rt.resetFrame(); rt.raytrace(F); rt.synchronize()
best = 1e9
for rep in range(3):
    rt.resetStats(); rt.resetFrame()
    t = time.perf_counter(); rt.raytrace(F); rt.synchronize(); best = min(best, time.perf_counter() - t)
st = rt.stats()
print("%%-10s %%-26s %%9.1f Mrays/s  %%7.3f ms/frame  rays(2 frames) %%d  crc %%08x" %% (scene, os.environ.get("AB_NAME"), st.rays_total / best / 1e6, best / F * 1e3, r2, crc), flush=True)
''' % root
for rnd in range(2):
    for s in scenes:
        for v in variants:
            env = dict(os.environ, AB_SCENE=s, AB_NAME=v)
            v, *sets = v.split("@")
            for kv in sets:
                k, _, val = kv.partition("=")
                env[k] = val
            if v != "default":
                env["PT_HIP_LIB"] = os.path.join(root, "pathtracer_amd", "variants", f"libpt_hip_{v}.so")
            r = subprocess.run([sys.executable, "-c", code], env=env, timeout=300)
            if r.returncode:
                sys.exit(r.returncode)
