#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DPT_REFILL_LANES=16 ..."   -> pathtracer_amd/variants/libpt_hip_NAME.so
set -e
cd "$(dirname "$0")/../pathtracer_amd/csrc"
mkdir -p ../variants /tmp/ptv_$1
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-value -Wno-unused-result $2"
for f in pt_api pt_trace pt_bounce pt_paths pt_bvh_build; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/ptv_$1/$f.o & done
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -c pt_host_seam.cpp -o /tmp/ptv_$1/pt_host_seam.o &
wait
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o ../variants/libpt_hip_$1.so /tmp/ptv_$1/*.o
echo built ../variants/libpt_hip_$1.so
