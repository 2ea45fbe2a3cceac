#!/bin/bash
# Copies the summaries of one scripts/gpu_round_profile.sh run from gpurun_out/<tag>/ into profiles/ and regenerates
# profiles/traffic_paths.json.  usage: scripts/collect_profile.sh <tag> '<kernel name as rocprofv3 prints it>'
set -e
T=$1
K=${2:-"k_paths<false, false, 2>"}
O=gpurun_out/$T
cp $O/bench_default_line.json profiles/${T}_bench_default_line.json
cp $O/bench_steps16_line.json profiles/${T}_bench_steps16_line.json
cp "$(find $O/stats -name '*kernel_stats.csv' | head -1)" profiles/${T}_bench_steps16_kernel_stats.csv
for k in fetch write; do
  f=$(find $O/pmc_$k -name '*counter_collection.csv')
  (head -1 $f; grep -E "k_paths|k_accumulate_items|k_frame" $f) > profiles/${T}_pmc_${k}_size.csv
done
python scripts/make_traffic_json.py $O/pmc_fetch $O/pmc_write profiles/traffic_paths.json "$K" | cut -c1-160
