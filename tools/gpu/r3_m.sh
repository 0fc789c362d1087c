#!/bin/bash
# round 3, call m: LayerNorm fold A/B on the measurement build (same binary, switch only), plan marks
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$1: "; env GAVIKO_HIP_DIAG=1 $2 timeout -k 10 300 python bench.py --allow-diag --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2 3 4; do
run "fold LN1 on " "GAVIKO_HIP_FOLD_LN1=1"
run "fold LN1 off" "GAVIKO_HIP_FOLD_LN1=0"
done
timeout -k 10 300 python tools/plan_marks.py 4 2>&1 | grep -v amdgpu | head -8
