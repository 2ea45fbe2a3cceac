#!/bin/bash
# Texture-addresser / vector-L1 counter passes over bench.py (each pass its own run): is k_paths bound by the rate at which
# the CU's TA/TCP works through divergent 16-byte loads?   usage: bash scripts/gpu_pmc_ta.sh <tag> [bench args]
set -o pipefail
TAG=${1:-ta}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 60 rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
grep -oE "\b(TA|TCP|TD)_[A-Z0-9_a-z]+" $OUT/avail.txt | sort -u > $OUT/avail_ta_tcp.txt
if [ "$1" = "list" ]; then wc -l $OUT/avail_ta_tcp.txt; exit 0; fi
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE" \
           "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --ramp-ms 0 "$@" > $OUT/pass$i.log 2>&1 || { tail -5 $OUT/pass$i.log; echo "pass $i failed"; exit 1; }
  python3 $R/scripts/pmc_summary.py $OUT/pass$i | grep -E "k_paths" | cut -c1-700
done
