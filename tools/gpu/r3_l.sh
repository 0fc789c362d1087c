#!/bin/bash
# round 3, call l: LayerNorm fold -- kernel test, gaviko goldens, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3l
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -s -k "layernorm_folded or gemm_epilogues or column_scale" > gpurun_out/r3l/test_fold.log 2>&1; echo "fold tests rc=$?"; grep -E "passed|failed|^FAILED|^LN fold|Error" gpurun_out/r3l/test_fold.log | tail -10
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_model_dropout_gpu.py tests/test_distributed_gpu.py -q -k "gaviko or dropout or two_process or bench" > gpurun_out/r3l/test_model.log 2>&1; echo "model tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3l/test_model.log | tail -10
for k in 1 2 3; do echo -n "step: "; timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; done
