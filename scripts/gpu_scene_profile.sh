#!/bin/bash
# Bench line + rocprofv3 kernel stats + PMC passes (each its own run) for ONE bench.py workload other than the default:
#   bash scripts/gpu_scene_profile.sh <tag> <steps per PMC pass> <bench args...>     (through gpurun)
# e.g. ... r03a_viking 4 --scene viking --width 3840 --height 2160
# Collected afterwards with scripts/collect_profile.sh <tag> auto <suffix>  (-> profiles/valu_paths<suffix>.json, traffic_paths<suffix>.json)
set -o pipefail
TAG=$1; STEPS=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 400 python -u bench.py "$@" > $OUT/bench_default_line.json 2> $OUT/bench_default.err || { tail $OUT/bench_default.err; exit 1; }
cut -c1-600 $OUT/bench_default_line.json
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps $STEPS --warmup $STEPS --no-cpu-baseline --ramp-ms 0 $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/bench_steps32_line.json 2> $OUT/stats.err || { tail $OUT/stats.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || { tail $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || { tail $OUT/pmc_write.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1 -- $B > $OUT/pmc_sq1.log 2>&1 || { tail $OUT/pmc_sq1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || { tail $OUT/pmc_sq2.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -- $B > $OUT/pmc_tcc.log 2>&1 || { tail $OUT/pmc_tcc.log; exit 1; }
cd $R
for p in pmc_fetch pmc_write pmc_sq1 pmc_sq2 pmc_tcc; do python scripts/pmc_summary.py $OUT/$p | grep -E "k_paths" ; done > $OUT/pmc_summary.log
cut -c1-500 $OUT/pmc_summary.log
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160 | head -8
