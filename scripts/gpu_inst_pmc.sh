#!/bin/bash
# Counters of k_paths<_, true, _, _> on the instanced grid (BASELINE configs[3] shape), each pass its own run.
# usage: bash scripts/gpu_inst_pmc.sh <tag>
set -o pipefail
TAG=${1:-instpmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export GRIDS=64
timeout -k 10 200 python3 $R/scripts/gpu_instanced_scene.py | tee $OUT/plain.log || exit 1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $R/scripts/gpu_instanced_scene.py > $OUT/pass$i.log 2>&1 || { tail -5 $OUT/pass$i.log; exit 1; }
  python3 $R/scripts/pmc_summary.py $OUT/pass$i | grep -E "k_paths"
done
