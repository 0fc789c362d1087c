#!/bin/bash
O=gpurun_out/r3ag; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_model_dropout_gpu.py -q -m gpu -k "adaptformer" -s > $O/af.log 2>&1; echo "rc=$?"
grep -E "passed|failed|^FAILED|Error|assert |lowest cos" $O/af.log | cut -c1-300 | tail -30
