#!/bin/bash
O=gpurun_out/r3host; mkdir -p $O
python tools/host_split.py 4 2>&1 | tail -2
python tools/host_split.py 2 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests/test_train_loop_gpu.py tests/test_distributed_gpu.py tests/test_optim.py -q -m gpu > $O/t.log 2>&1; echo "rc=$?"; tail -2 $O/t.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "gaviko_t16 or fft_t16 or accumul or eval" > $O/m.log 2>&1; echo "rc=$?"; tail -2 $O/m.log
for b in 4 2; do python bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'; done
