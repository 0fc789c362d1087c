#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 $3 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2; do
run "8-wave prompt-side bwd" "X=1"
run "2-wave prompt-side bwd" "GAVIKO_HIP_GPA_BWD_WAVES=2"
done
run "B=2 8-wave" "X=1" "--batch 2"
run "B=2 2-wave" "GAVIKO_HIP_GPA_BWD_WAVES=2" "--batch 2"
