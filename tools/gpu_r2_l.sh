#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2l; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_dropout_gpu.py tests/test_model_dropout_gpu.py -x -q -k "attention or attn or dropout" > $O/t.log 2>&1; tail -3 $O/t.log
python3 tools/probe/probe_attn.py 2>&1 | tee $O/attn_stamps2.txt
python3 tools/bench_attn.py 2>/dev/null
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c80-150
