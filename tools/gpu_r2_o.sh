#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --allow-ablate 2>/dev/null | cut -c70-150; }
run "default" "X=1"
run "sidenop" "GAVIKO_HIP_ABLATE=sidenop"
run "noside" "GAVIKO_HIP_ABLATE=noside"
GAVIKO_HIP_ABLATE=sidenop python3 tools/plan_marks.py 4 2>/dev/null
