#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "sidepath or gaviko or distributed or train_loop or abi" > $O/t.log 2>&1; tail -5 $O/t.log
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "next fused" "X=1"
run "next off" "GAVIKO_HIP_FUSE_NEXT=0"
run "next fused" "X=1"
run "next off" "GAVIKO_HIP_FUSE_NEXT=0"
