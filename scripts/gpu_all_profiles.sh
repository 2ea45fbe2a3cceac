#!/bin/bash
# Every workload's bench line + kernel stats + PMC passes on the sources in the tree (the per-workload counter summaries that
# bench.py prices its roofline with are only valid for the kernel sources they were taken on):
#   bash scripts/gpu_all_profiles.sh <prefix> [small|big|all]      (through gpurun; `big` = the 100 M-triangle soup, a call of its own)
# then, on the host:  for the tags it prints, scripts/collect_profile.sh <tag> auto <suffix>
set -o pipefail
P=${1:-r03z}; WHAT=${2:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "$WHAT" != "big" ]; then
bash scripts/gpu_round_profile.sh ${P} notests || exit 1
bash scripts/gpu_scene_profile.sh ${P}_viking 4 --scene viking --width 3840 --height 2160 || exit 1
bash scripts/gpu_scene_profile.sh ${P}_teapot 8 --scene teapot || exit 1
bash scripts/gpu_scene_profile.sh ${P}_soup1m 8 --scene soup --builder lbvh || exit 1
fi
if [ "$WHAT" != "small" ]; then
bash scripts/gpu_scene_profile.sh ${P}_soup100m 4 --scene soup --tris 1e8 --builder lbvh || exit 1
fi
echo "collect: scripts/collect_profile.sh ${P}; ... ${P}_viking auto _viking; ${P}_teapot auto _teapot; ${P}_soup1m auto _soup_1000000; ${P}_soup100m auto _soup"
