#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 $3 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2; do
run "windows 8 waves" "X=1"
run "windows 4 waves" "GAVIKO_HIP_LIB=$PWD/gaviko_amd/libgaviko_hip_wm4.so"
run "windows 2 waves" "GAVIKO_HIP_LIB=$PWD/gaviko_amd/libgaviko_hip_wm2.so"
done
run "B=2 8 waves" "X=1" "--batch 2"
run "B=2 4 waves" "GAVIKO_HIP_LIB=$PWD/gaviko_amd/libgaviko_hip_wm4.so" "--batch 2"
run "B=2 2 waves" "GAVIKO_HIP_LIB=$PWD/gaviko_amd/libgaviko_hip_wm2.so" "--batch 2"
