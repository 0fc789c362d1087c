#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2; do
run "LN 4 rows/WG" "X=1"
run "LN 2 rows/WG" "GAVIKO_HIP_LIB=$PWD/gaviko_amd/libgaviko_hip_ln2.so"
run "LN 8 rows/WG" "GAVIKO_HIP_LIB=$PWD/gaviko_amd/libgaviko_hip_ln8.so"
done
