"""A/B of library variants in ONE process each but interleaved rounds: prints frame ms per variant."""
import os, subprocess, sys
root = os.path.join(os.path.dirname(__file__), "..")
variants = sys.argv[1:]
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import pathtracer_amd as pa
W, H, S, D = 1920, 1080, int(os.environ.get("SPP", "16")), 8
sc = pa.scenes.cornell_box(); view, proj = pa.reference_camera(W, H)
rt = pa.HipRaytracing(0); rt.buildBlas(sc); rt.resize(W, H); rt.setCamera(view, proj)
rt.setParams(spp_per_frame=S, max_depth=D, max_frames=0)
rt.raytrace(3); rt.synchronize(); rt.resetStats(); rt.resetFrame()
best = 1e9
for rep in range(3):
    t = time.perf_counter(); rt.raytrace(8); rt.synchronize(); best = min(best, (time.perf_counter() - t) / 8)
st = rt.stats()
rt.setProfiling(pa.api.PT_PROFILE_TRAVERSAL); rt.resetStats(); rt.resetFrame(); rt.raytrace(1); rt.synchronize(); s2 = rt.stats()
print("%%-28s frame %%.3f ms  %%.0f Mrays/s   nodes/ray %%.3f tris/ray %%.3f" %% (os.environ.get("PT_HIP_LIB", "default")[-28:], best * 1e3, st.rays_total / 24 / best / 1e6, s2.trav_nodes / s2.rays_total, s2.trav_tris / s2.rays_total), flush=True)
''' % root
for rnd in range(2):
    for v in variants:
        env = dict(os.environ)
        v, *sets = v.split("@")  # name@VAR=value@VAR2=value: environment for this variant only
        for kv in sets:
            k, _, val = kv.partition("=")
            env[k] = val
        if v != "default":
            env["PT_HIP_LIB"] = os.path.abspath(os.path.join(root, "pathtracer_amd", "variants", f"libpt_hip_{v}.so"))
        subprocess.run([sys.executable, "-c", code], env=env, check=True)
