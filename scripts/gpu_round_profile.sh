#!/bin/bash
# One GPU call: tests, un-profiled bench line, rocprofv3 kernel stats of the same command, FETCH/WRITE PMC passes.
# usage (through gpurun): bash scripts/gpu_round_profile.sh <tag>
set -o pipefail
TAG=${1:-r01c}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -20 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 300 python -u bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err || { tail $OUT/bench_default.err; exit 1; }
cat $OUT/bench_default_line.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline > $OUT/bench_steps16_line.json 2> $OUT/stats.err || { tail $OUT/stats.err; exit 1; }
cat $OUT/bench_steps16_line.json
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 16 --warmup 16 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || { tail $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 16 --warmup 16 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || { tail $OUT/pmc_write.log; exit 1; }
cd $R
python scripts/pmc_summary.py $OUT/pmc_fetch; python scripts/pmc_summary.py $OUT/pmc_write
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-120
