#!/bin/bash
# bench.py's N > 1 code path on ONE GPU: N ranks under torch.distributed.run, all rendering on GPU 0, gloo instead of RCCL
# (RCCL refuses two ranks on one device).  What it rehearses: the self-launch, rank -> frame schedule / tile schedule, sum
# accumulation, the reduce, the max-over-ranks timing and the one JSON line; the image of every mode is compared with a
# one-process render of the same global frames.  Not a scaling measurement (the ranks share one GPU).
#   bash scripts/gpu_multirank_rehearsal.sh <tag>      (through gpurun)
set -o pipefail
TAG=${1:-rehearsal}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
W=480; H=270
COMMON="--width $W --height $H --warmup 2 --no-cpu-baseline --dist-backend gloo"
run() { name=$1; shift; timeout -k 10 300 python -u bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 1; }; }
run one_16 --width $W --height $H --warmup 2 --no-cpu-baseline --steps 16 --out-image $OUT/one_16.npy
run weak2 $COMMON --gpus 2 --steps 8 --out-image $OUT/weak2.npy
run weak4 $COMMON --gpus 4 --steps 4 --out-image $OUT/weak4.npy
run strong3 $COMMON --gpus 3 --steps 16 --scaling strong --out-image $OUT/strong3.npy
run tiles4 $COMMON --gpus 4 --steps 16 --split tiles --out-image $OUT/tiles4.npy
python3 - $OUT <<'PY'
import json, sys
import numpy as np
out = sys.argv[1]
ref = np.load(out + "/one_16.npy")
for name in ("weak2", "weak4", "strong3", "tiles4"):
    d = json.loads(open(f"{out}/{name}.json").read())
    img = np.load(f"{out}/{name}.npy")
    err = float(np.abs(img[..., :3] - ref[..., :3]).max())
    same = bool(np.array_equal(img.view(np.uint32), ref.view(np.uint32)))
    print(f"{name:8s} n_gpus {d['n_gpus']} scaling {d['scaling']:6s} total_spp {d['config']['total_spp']:4d} value {d['value']:9.1f} Mrays/s  "
          f"max |image - one-process image of the same 16 frames| = {err:.3e}{' (bit-identical)' if same else ''}", flush=True)
    # the same 16 global frames, summed (N ranks, then / 16) instead of folded into a running mean (one process): fp32 rounding
    assert d["config"]["total_spp"] == 256 and err < 2e-5, name
print("rehearsal ok")
PY
