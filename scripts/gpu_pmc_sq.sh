#!/bin/bash
# SQ counter passes over bench.py (each pass its own run).  usage: bash scripts/gpu_pmc_sq.sh <tag> [bench args]
set -o pipefail
TAG=${1:-sq}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --ramp-ms 0 "$@" > $OUT/pass$i.log 2>&1 || { tail -5 $OUT/pass$i.log; exit 1; }
  python3 $R/scripts/pmc_summary.py $OUT/pass$i | grep -E "k_paths|k_bounce" 
done
