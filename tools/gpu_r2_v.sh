#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --allow-ablate 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "default" "X=1"
run "nowin" "GAVIKO_HIP_ABLATE=nowin"
run "loc_noupdown" "GAVIKO_HIP_ABLATE=loc_noupdown"
run "loc_noouter" "GAVIKO_HIP_ABLATE=loc_noouter"
run "locnop" "GAVIKO_HIP_ABLATE=locnop"
run "gpanop" "GAVIKO_HIP_ABLATE=gpanop"
run "sidenop" "GAVIKO_HIP_ABLATE=sidenop"
echo "B=2"; python bench.py --steps 30 --warmup 10 --batch 2 --no-cpu-baseline --no-roofline 2>&1 | grep -o "\"value\": [0-9.]*"
