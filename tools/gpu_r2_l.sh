#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2l; mkdir -p $O
python -m pytest tests/test_sidepath_kernels_gpu.py -x -q > $O/t.log 2>&1; tail -3 $O/t.log
python3 tools/bench_side.py 2>/dev/null | grep outer
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --allow-ablate 2>/dev/null | cut -c80-150; }
run "default" "X=1"
run "nowin" "GAVIKO_HIP_ABLATE=nowin"
run "noparams (no GPA param-gradient kernels)" "GAVIKO_HIP_ABLATE=noparams"
run "noside" "GAVIKO_HIP_ABLATE=noside"
run "default" "X=1"
