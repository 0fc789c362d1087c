#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2; do
run "row LN kernels (0)" "GAVIKO_HIP_SIDE_LN=0"
run "tile fwd only (f)" "GAVIKO_HIP_SIDE_LN=f"
run "tile bwd only (b)" "GAVIKO_HIP_SIDE_LN=b"
run "tile both (1)" "GAVIKO_HIP_SIDE_LN=1"
done
