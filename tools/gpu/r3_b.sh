#!/bin/bash
# round 3, call b: full GPU test suite with the new attention forward (parity report -> gpurun_out/parity_report.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3b
timeout -k 10 1000 python -m pytest tests -q -m gpu -x --deselect "tests/test_kernels_gpu.py::test_attention_fwd" > gpurun_out/r3b/test_all.log 2>&1; echo "gpu tests rc=$?"; tail -8 gpurun_out/r3b/test_all.log
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -k "test_attention_fwd" > gpurun_out/r3b/test_attn.log 2>&1; echo "attention tests rc=$?"; grep -E "passed|failed|^FAILED|assert 0\." gpurun_out/r3b/test_attn.log | head -20
