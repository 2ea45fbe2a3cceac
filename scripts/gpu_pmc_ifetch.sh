#!/bin/bash
# Instruction-fetch counters of k_paths on the headline workload (is the 50 KB kernel body waiting on the instruction cache?)
# usage (through gpurun): bash scripts/gpu_pmc_ifetch.sh <tag>
set -o pipefail
TAG=${1:-ifetch}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --ramp-ms 0"
timeout -s KILL 90 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc_if1 -- $B > $OUT/pmc_if1.log 2>&1 || { tail $OUT/pmc_if1.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_if2 -- $B > $OUT/pmc_if2.log 2>&1 || { tail $OUT/pmc_if2.log; exit 1; }
cd $R
for p in pmc_if1 pmc_if2; do python scripts/pmc_summary.py $OUT/$p | grep -E "k_paths" ; done > $OUT/pmc_summary.log
cat $OUT/pmc_summary.log | cut -c1-600
