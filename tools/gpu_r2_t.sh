#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "nt" "X=1"
run "no nt" "GAVIKO_HIP_LIB=$PWD/gaviko_amd/libgaviko_hip_nont.so"
run "nt" "X=1"
run "no nt" "GAVIKO_HIP_LIB=$PWD/gaviko_amd/libgaviko_hip_nont.so"
python3 tools/plan_marks.py 4 2>/dev/null
