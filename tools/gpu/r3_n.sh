#!/bin/bash
# round 3, call n: fused attention backward (delta from the out-projection dgrad) -- tests, isolated timing, step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3n
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_sidepath_kernels_gpu.py tests/test_dropout_gpu.py -q -k "attention or delta_side" > gpurun_out/r3n/test_attn.log 2>&1; echo "attention tests rc=$?"; grep -E "passed|failed|^FAILED|Error" gpurun_out/r3n/test_attn.log | tail -6
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_model_dropout_gpu.py -q > gpurun_out/r3n/test_model.log 2>&1; echo "model tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3n/test_model.log | tail -6
for k in 1 2 3; do echo -n "step: "; timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; done
timeout -k 10 300 python tools/plan_marks.py 4 2>&1 | grep -v amdgpu | sed -n 6,16p
