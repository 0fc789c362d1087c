#!/bin/bash
O=gpurun_out/r3ah; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -q -m gpu -s -k "unfrozen or fft_t16 or bitfit or (gaviko_t16_b2 and not lat16 and not k366)" > $O/uf.log 2>&1; echo "rc=$?"
grep -E "passed|failed|^FAILED|Error|assert |gradnorm rel" $O/uf.log | cut -c1-330 | tail -30
