#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2w; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_sidepath_kernels_gpu.py -x -q -k "window" > $O/t.log 2>&1; tail -3 $O/t.log
python3 tools/bench_side.py 2>/dev/null | grep -i "win" 
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "mfma windows" "X=1"
run "row windows" "GAVIKO_HIP_WIN_MFMA=0"
run "mfma windows" "X=1"
