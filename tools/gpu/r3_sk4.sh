#!/bin/bash
O=gpurun_out/r3sk; mkdir -p $O
GAVIKO_HIP_DIAG=1 timeout -k 10 120 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "stream_k" 2>&1 | tail -1
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -q -m gpu 2>&1 | tail -1
for i in 1 2; do python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'; done
