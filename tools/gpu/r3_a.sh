#!/bin/bash
# round 3, call a: new attention forward -- parity tests, variant race, whole-step bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3a
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > gpurun_out/r3a/test_attn.log 2>&1; echo "attention tests rc=$?"; tail -5 gpurun_out/r3a/test_attn.log
timeout -k 10 300 python tools/bench_attn.py --fwd-variants > gpurun_out/r3a/bench_attn.log 2>&1; echo "bench_attn rc=$?"; cat gpurun_out/r3a/bench_attn.log
for v in 0 1 2 3; do echo -n "step VAR=$v: "; GAVIKO_HIP_ATTN_VAR=$v timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; done
