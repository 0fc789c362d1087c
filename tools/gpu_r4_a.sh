#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "split_k_eight_wave" 2>&1 | tail -3
ONLY=out,fc2,fc1_dgrad,out_dgrad,qkv_dgrad TILES=4128128,9128128,3128128 python3 tools/bench_gemm.py 2>/dev/null
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2; do
run "k4 tiles" "GAVIKO_HIP_GEMM_K4=1"
run "four-wave tiles" "X=1"
done
