#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2u; mkdir -p $O
python -m pytest tests -x -q -m gpu -k "gaviko or distributed" > $O/t.log 2>&1; tail -3 $O/t.log
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "shift" "X=1"
run "no shift" "GAVIKO_HIP_LOC_SHIFT=0"
run "shift" "X=1"
run "no shift" "GAVIKO_HIP_LOC_SHIFT=0"
python3 tools/plan_marks.py 4 2>/dev/null | tail -9
