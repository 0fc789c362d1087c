#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$1: "; env $2 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2 3; do
run "next fused" "X=1"
run "next off  " "GAVIKO_HIP_FUSE_NEXT=0"
done
run "both off  " "GAVIKO_HIP_FUSE_NEXT=0 GAVIKO_HIP_FUSE_BOUNDARY=0"
