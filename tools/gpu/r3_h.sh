#!/bin/bash
# round 3, call h: 96-row GEMM tiles -- tests, cfg5 A/B (row tile, wide-tile threshold)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3h
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -k "gemm" > gpurun_out/r3h/test_gemm.log 2>&1; echo "gemm tests rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/r3h/test_gemm.log | tail -5
run() { echo -n "$1: "; env $2 timeout -k 10 300 python bench.py --backbone vit-l16 --batch 2 --steps 25 --warmup 6 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2; do
run "cfg5 default (BM rule)" "X=1"
run "cfg5 BM=128" "GAVIKO_HIP_GEMM_BM=128"
run "cfg5 BM=96 + wide>=100" "GAVIKO_HIP_GEMM_WIDE_LO=100"
run "cfg5 BM=96 + wide>=140" "GAVIKO_HIP_GEMM_WIDE_LO=140"
done
