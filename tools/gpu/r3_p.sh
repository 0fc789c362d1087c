#!/bin/bash
# full GPU suite on the product build, then the diag-only cases
set -e
mkdir -p gpurun_out/r3p
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r3p/suite.log 2>&1 || { tail -40 gpurun_out/r3p/suite.log; exit 1; }
tail -3 gpurun_out/r3p/suite.log
cp gpurun_out/parity_report.txt gpurun_out/r3p/parity_report.txt
GAVIKO_HIP_DIAG=1 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "key_tiles or patch or diag or stream_k" > gpurun_out/r3p/diag.log 2>&1 || { tail -30 gpurun_out/r3p/diag.log; exit 1; }
tail -2 gpurun_out/r3p/diag.log
for i in 1 2 3; do python bench.py --steps 30 --warmup 10 2>/dev/null | tail -1 | cut -c1-200; done
