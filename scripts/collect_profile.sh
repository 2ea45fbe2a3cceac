#!/bin/bash
# Copies the summaries of one scripts/gpu_round_profile.sh (headline workload) or scripts/gpu_scene_profile.sh (any other
# workload) run from gpurun_out/<tag>/ into profiles/ and regenerates profiles/traffic_paths<suffix>.json +
# profiles/valu_paths<suffix>.json.
# usage: scripts/collect_profile.sh <tag> ['<kernel name>'|auto] [suffix]     suffix: "" (cornell), _viking, _teapot, _soup
set -e
T=$1
K=${2:-"k_paths<false, false, 0, 2>"}
S=${3:-}
O=gpurun_out/$T
cp $O/bench_default_line.json profiles/${T}_bench_default_line.json
cp $O/bench_steps32_line.json profiles/${T}_bench_stats_line.json
cp "$(find $O/stats -name '*kernel_stats.csv' | head -1)" profiles/${T}_bench_kernel_stats.csv
for k in fetch write; do
  f=$(find $O/pmc_$k -name '*counter_collection.csv')
  (head -1 $f; grep -E "k_paths|k_accumulate_items|k_frame" $f) > profiles/${T}_pmc_${k}_size.csv
done
cp $O/pmc_summary.log profiles/${T}_pmc_sq.log
python scripts/make_pmc_json.py $O "$K" "$(git rev-parse --short HEAD)" "$S"
