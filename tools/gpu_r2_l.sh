#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2l; mkdir -p $O
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>$O/err.log | cut -c80-150; tail -2 $O/err.log | grep -i error; }
run "default" "X=1"
run "main prio -1" "GAVIKO_BENCH_MAIN_PRIORITY=-1"
run "main prio 0 (own stream)" "GAVIKO_BENCH_MAIN_PRIORITY=0"
run "main prio -1, side prio 0 explicit" "GAVIKO_BENCH_MAIN_PRIORITY=-1 GAVIKO_HIP_SIDE_PRIORITY=0"
run "default again" "X=1"
