#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --allow-ablate 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "default" "X=1"
run "nowait (main stream never waits)" "GAVIKO_HIP_ABLATE=nowait"
run "noevents" "GAVIKO_HIP_ABLATE=noevents"
GAVIKO_HIP_ABLATE=nowait python3 tools/plan_marks.py 4 2>/dev/null
