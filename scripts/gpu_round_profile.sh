#!/bin/bash
# One GPU call: tests, un-profiled bench line, rocprofv3 kernel stats of the same command, PMC passes (each its own run):
# FETCH_SIZE, WRITE_SIZE, two SQ sets (the second with GRBM_GUI_ACTIVE for the clock the chip held).
# usage (through gpurun): bash scripts/gpu_round_profile.sh <tag> [notests]
set -o pipefail
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
if [ "$2" != "notests" ]; then
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 600 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -20 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
fi
timeout -k 10 300 python -u bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err || { tail $OUT/bench_default.err; exit 1; }
cat $OUT/bench_default_line.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline --ramp-ms 0 > $OUT/bench_steps32_line.json 2> $OUT/stats.err || { tail $OUT/stats.err; exit 1; }
cat $OUT/bench_steps32_line.json
B="python3 $R/bench.py --steps 16 --warmup 16 --no-cpu-baseline --ramp-ms 0"  # (one launch = 16 frames: every launch of the run is a full one)
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || { tail $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || { tail $OUT/pmc_write.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1 -- $B > $OUT/pmc_sq1.log 2>&1 || { tail $OUT/pmc_sq1.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || { tail $OUT/pmc_sq2.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -- $B > $OUT/pmc_tcc.log 2>&1 || { tail $OUT/pmc_tcc.log; exit 1; }
cd $R
for p in pmc_fetch pmc_write pmc_sq1 pmc_sq2 pmc_tcc; do python scripts/pmc_summary.py $OUT/$p | grep -E "k_paths" ; done > $OUT/pmc_summary.log
cat $OUT/pmc_summary.log | cut -c1-400
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-140
