#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 $3 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "default" "X=1"
run "gpa stream high priority" "GAVIKO_HIP_GPA_PRIORITY=-1"
run "loc stream high priority" "GAVIKO_HIP_LOC_PRIORITY=-1"
run "default" "X=1"
run "gpa stream high priority" "GAVIKO_HIP_GPA_PRIORITY=-1"
run "B=2 default" "X=1" "--batch 2"
run "B=2 gpa high" "GAVIKO_HIP_GPA_PRIORITY=-1" "--batch 2"
