#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "default" "X=1"
run "K768: 64x128" "GAVIKO_HIP_GEMM_N768_K768=64"
run "K768: 128x128 2-stage" "GAVIKO_HIP_GEMM_N768_K768=128"
run "default" "X=1"
