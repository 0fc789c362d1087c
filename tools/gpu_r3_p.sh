#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2; do
run "8-phase variant 0 (DMA in load sections)" "X=1"
run "8-phase variant 1 (DMA inside MFMA clusters)" "GAVIKO_HIP_LIB=$PWD/gaviko_amd/libgaviko_hip_g7.so"
done
