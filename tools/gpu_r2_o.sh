#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --allow-ablate 2>/dev/null | grep -o '"value": [0-9.]*'; }
run "default" "X=1"
run "sidenop" "GAVIKO_HIP_ABLATE=sidenop"
run "locnop" "GAVIKO_HIP_ABLATE=locnop"
run "gpanop" "GAVIKO_HIP_ABLATE=gpanop"
echo "--- locnop marks"; GAVIKO_HIP_ABLATE=locnop python3 tools/plan_marks.py 4 2>/dev/null
echo "--- gpanop marks"; GAVIKO_HIP_ABLATE=gpanop python3 tools/plan_marks.py 4 2>/dev/null
