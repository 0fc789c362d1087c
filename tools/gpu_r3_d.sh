#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "chunks 1" "X=1"
run "chunks 2" "GAVIKO_HIP_SIDE_CHUNKS=2"
run "chunks 4" "GAVIKO_HIP_SIDE_CHUNKS=4"
run "chunks 1" "X=1"
run "chunks 3" "GAVIKO_HIP_SIDE_CHUNKS=3"
