#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*\|gaviko_hip:.*' | sort | uniq -c | head -5; }
run "default" "X=1"
run "pad loc 32K" "GAVIKO_HIP_LDS_PAD_LOC=32768"
run "pad loc 60K" "GAVIKO_HIP_LDS_PAD_LOC=61440"
run "pad loc 64K" "GAVIKO_HIP_LDS_PAD_LOC=65536"
run "pad loc 96K" "GAVIKO_HIP_LDS_PAD_LOC=98304"
