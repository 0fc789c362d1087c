#!/bin/bash
O=gpurun_out/r3aj; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_model_dropout_gpu.py -q -m gpu -s -k "ssf" > $O/dvpt.log 2>&1; echo "rc=$?"
grep -E "passed|failed|^FAILED|Error|assert |gradnorm rel" $O/dvpt.log | cut -c1-330 | tail -30
