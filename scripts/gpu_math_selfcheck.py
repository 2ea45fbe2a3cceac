"""pt_debug_math over 2^33 operand sets (8 seeds x 2^30): mismatch counts must all be zero."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import pathtracer_amd as pa
rt = pa.HipRaytracing(0)
tot = [0, 0, 0, 0]
t = time.perf_counter()
for seed in range(8):
    bad = rt.debugMath(1 << 30, 100 + seed)
    tot = [a + int(b) for a, b in zip(tot, bad)]
    print("seed", seed, "mismatches (quotients, roots, reciprocal roots), sets checked", bad.tolist(), flush=True)
print("total", tot, "over", 8 << 30, "operand sets in %.1f s" % (time.perf_counter() - t))
assert tot == [0, 0, 0, 8 << 30]
