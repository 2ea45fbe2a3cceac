"""profiles/traffic_bounce.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
HBM bytes per k_bounce launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE/WRITE_SIZE are in KiB, and on
gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, section HBM)."""
import json
import sys

sys.path.insert(0, "scripts")
from pmc_summary import load

fetch_dir, write_dir, out, kernel = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
f = load(fetch_dir)[kernel]["FETCH_SIZE"]
w = load(write_dir)[kernel]["WRITE_SIZE"]
fm, wm = sum(f) / len(f), sum(w) / len(w)
d = {"kernel": kernel, "launches_sampled": len(f), "FETCH_SIZE_KiB_mean": fm, "WRITE_SIZE_KiB_mean": wm,
     "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), WRITE_SIZE x1",
     "hbm_bytes_per_launch": (2.0 * fm + wm) * 1024.0,
     "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 16 --warmup 16 --no-cpu-baseline (mean over all launches of the kernel; one launch = 16 frames)"}
json.dump(d, open(out, "w"), indent=1)
print(d)
