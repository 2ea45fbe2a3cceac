#!/bin/bash
# HBM traffic of the BVH build, per kernel: one rocprofv3 --pmc FETCH_SIZE and one --pmc WRITE_SIZE pass (each its own run) over
# scripts/gpu_build_only.py, plus the kernel-trace timing of the same command.   usage: bash scripts/gpu_build_pmc.sh <tag> [N=1e8] [builders=lbvh]
set -o pipefail
TAG=${1:-buildpmc}
export N=${2:-1e8} BUILDERS=${3:-lbvh} REPS=2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/scripts/gpu_build_only.py > $OUT/stats.log 2>&1 || { tail $OUT/stats.log; exit 1; }
grep "build" $OUT/stats.log
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/gpu_build_only.py > $OUT/pmc_fetch.log 2>&1 || { tail $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/gpu_build_only.py > $OUT/pmc_write.log 2>&1 || { tail $OUT/pmc_write.log; exit 1; }
cd $R
python3 scripts/build_traffic.py $OUT $N > $OUT/build_traffic.txt && cat $OUT/build_traffic.txt
