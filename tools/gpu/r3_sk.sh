#!/bin/bash
O=gpurun_out/r3sk; mkdir -p $O
timeout -k 10 120 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "stream_k" > $O/sk.log 2>&1; echo "rc=$?"
tail -25 $O/sk.log | cut -c1-250
