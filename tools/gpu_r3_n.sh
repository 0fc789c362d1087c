#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2; do
run "default" "X=1"
run "FUSE_PROJ=0" "GAVIKO_HIP_FUSE_PROJ=0"
done
