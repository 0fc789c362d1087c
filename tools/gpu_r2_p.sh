#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|Error.*' ; }
run "default" "X=1"
run "lo:64" "GAVIKO_HIP_SIDE_CUMASK=lo:64"
run "lo:128" "GAVIKO_HIP_SIDE_CUMASK=lo:128"
run "hi:64" "GAVIKO_HIP_SIDE_CUMASK=hi:64"
run "mod:4:0" "GAVIKO_HIP_SIDE_CUMASK=mod:4:0"
run "mod:2:0" "GAVIKO_HIP_SIDE_CUMASK=mod:2:0"
run "modlt:32:8 (8 of each 32)" "GAVIKO_HIP_SIDE_CUMASK=modlt:32:8"
run "modlt:8:2" "GAVIKO_HIP_SIDE_CUMASK=modlt:8:2"
run "lo:32" "GAVIKO_HIP_SIDE_CUMASK=lo:32"
