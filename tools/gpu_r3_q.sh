#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "gaviko or abi or train_loop" 2>&1 | tail -2
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2; do
run "fix in LN1" "X=1"
run "separate fix kernel" "GAVIKO_HIP_FIX_IN_LN=0"
done
