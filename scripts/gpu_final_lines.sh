#!/bin/bash
# The bench lines of a round, on the final sources and with the round's PMC summaries in place:
#   bash scripts/gpu_final_lines.sh <tag>      (through gpurun; every line -> gpurun_out/<tag>/)
set -o pipefail
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
run() { name=$1; shift; timeout -k 10 600 python -u bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 1; }; python3 - $OUT/$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r, c = d["roofline"], d.get("cpu_baseline") or {}
print(f"{sys.argv[1].split('/')[-1]:28s} {d['value']:10.1f} Mrays/s  {d['ms_per_step']:8.3f} ms/step  roofline {r['bound']} {r['frac']:.3f} (traffic {r.get('traffic')})  cpu {c.get('value')}", flush=True)
PY
}
run bench_default
run bench_driver_style --gpus 1 --steps 20 --warmup 5
run bench_viking --scene viking --width 3840 --height 2160
run bench_teapot --scene teapot
run bench_soup_1m --scene soup --builder lbvh
run bench_strong_forcedist --force-dist --scaling strong --steps 64 --warmup 8 --no-cpu-baseline
run bench_tiles_forcedist --force-dist --split tiles --scene soup --builder lbvh --no-cpu-baseline
run bench_soup_100m --scene soup --tris 1e8 --builder lbvh --steps 8 --warmup 2
timeout -k 10 400 python -u scripts/gpu_soak.py > $OUT/soak.log 2>&1 || { tail -5 $OUT/soak.log; exit 1; }
head -3 $OUT/soak.log
