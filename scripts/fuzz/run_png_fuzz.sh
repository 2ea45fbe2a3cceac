#!/bin/bash
# Host-only robustness check of pt_load_png (own inflate + unfilter): ~1600 PNGs with random damage and VALID chunk
# CRCs (so the damage reaches the inflate / unfilter code) through an ASan + UBSan build of the host seam.
# usage: bash scripts/fuzz/run_png_fuzz.sh     (CPU only; prints "ok N rejected M", any sanitizer report is a failure)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -I$R/include $R/scripts/fuzz/png_driver.cpp \
    $R/pathtracer_amd/csrc/pt_host_seam.cpp -o $T/drv
python3 $R/scripts/fuzz/make_png_corpus.py $T
ls $T/*.png | xargs $T/drv
rm -rf $T
