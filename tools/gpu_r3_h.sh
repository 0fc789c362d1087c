#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3h; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "gpa or gaviko" > $O/t.log 2>&1; tail -2 $O/t.log
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2; do
run "separate LN2-bwd + scatter" "X=1"
run "fused scatter" "GAVIKO_HIP_FUSE_SCATTER=1"
done
python3 tools/plan_marks.py 4 2>/dev/null | tail -9
