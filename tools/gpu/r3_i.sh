#!/bin/bash
# round 3, call i: after the diag split -- product GPU suite, diag-only tests on the measurement build, attention variant A/B in the step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3i
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r3i/test_all.log 2>&1; echo "gpu tests rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r3i/test_all.log | tail -8
GAVIKO_HIP_DIAG=1 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "split_k or implicit_gemm or key_tiles" > gpurun_out/r3i/test_diag.log 2>&1; echo "diag tests rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r3i/test_diag.log | tail -5
run() { echo -n "$1: "; env GAVIKO_HIP_DIAG=1 $2 timeout -k 10 300 python bench.py --allow-diag --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2 3; do
run "VAR=0 (VALU sums)" "GAVIKO_HIP_ATTN_VAR=0"
run "VAR=2 (spread DMA)" "GAVIKO_HIP_ATTN_VAR=2"
run "VAR=1 (ones-MFMA sums)" "GAVIKO_HIP_ATTN_VAR=1"
done
echo -n "product: "; timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*'
