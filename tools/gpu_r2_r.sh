#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2r; mkdir -p $O
python -m pytest tests -x -q -m gpu -k "gpa or gaviko or sidepath or abi" > $O/t.log 2>&1; tail -5 $O/t.log
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "fuse_up" "X=1"
run "fuse_up off" "GAVIKO_HIP_FUSE_UP=0"
run "fuse_up" "X=1"
python3 tools/plan_marks.py 4 2>/dev/null | head -6
