#!/bin/bash
# HBM / L2 counters of k_paths on a big scene (each pass its own run).  usage: bash scripts/gpu_big_pmc.sh <tag>   env: N SCENE SPP FRAMES
set -o pipefail
TAG=${1:-bigpmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 200 python3 $R/scripts/gpu_big_render.py | tee $OUT/plain.log || exit 1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $R/scripts/gpu_big_render.py > $OUT/pass$i.log 2>&1 || { tail -5 $OUT/pass$i.log; exit 1; }
  grep "^scene" $OUT/pass$i.log
  python3 $R/scripts/pmc_summary.py $OUT/pass$i | grep -E "k_paths"
done
