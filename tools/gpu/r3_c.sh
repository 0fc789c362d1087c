#!/bin/bash
# round 3, call c: new attention forward + backward -- kernel tests, isolated timing, full GPU suite, step rate
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3c
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_sidepath_kernels_gpu.py tests/test_dropout_gpu.py -q -s -k "attention or column_scale" > gpurun_out/r3c/test_attn.log 2>&1; echo "attention tests rc=$?"; grep -E "passed|failed|^FAILED|^attention_fwd B|^attention_bwd B=2 T=1033|^attention_bwd B=1 T=1001" gpurun_out/r3c/test_attn.log | tail -40
timeout -k 10 300 python tools/bench_attn.py > gpurun_out/r3c/bench_attn.log 2>&1; echo "bench_attn rc=$?"; cat gpurun_out/r3c/bench_attn.log
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r3c/test_all.log 2>&1; echo "gpu tests rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r3c/test_all.log | tail -15
for k in 1 2; do echo -n "step: "; timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; done
