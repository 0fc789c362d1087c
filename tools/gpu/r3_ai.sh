#!/bin/bash
O=gpurun_out/r3ai; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_model_dropout_gpu.py -q -m gpu -s > $O/drop.log 2>&1; echo "rc=$?"
grep -E "passed|failed|^FAILED|Error|assert |lowest cos" $O/drop.log | cut -c1-330 | tail -30
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "unfrozen" > $O/uf.log 2>&1; echo "rc=$?"; tail -2 $O/uf.log
