#!/bin/bash
# round 3, call e: pipelined attention backward -- kernel tests, isolated timing, dropout tests, step rate
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3e
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_sidepath_kernels_gpu.py tests/test_dropout_gpu.py tests/test_model_dropout_gpu.py -q -k "attention or dropout" > gpurun_out/r3e/test_attn.log 2>&1; echo "attention tests rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/r3e/test_attn.log | tail -10
timeout -k 10 300 python tools/bench_attn.py > gpurun_out/r3e/bench_attn.log 2>&1; echo "bench_attn rc=$?"; grep -v amdgpu gpurun_out/r3e/bench_attn.log
for k in 1 2 3; do echo -n "step: "; timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; done
