#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "$1"; env $2 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --allow-ablate 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "default" "X=1"
run "loc_noupdown" "GAVIKO_HIP_ABLATE=loc_noupdown"
run "loc_noouter" "GAVIKO_HIP_ABLATE=loc_noouter"
run "loc_nosmall" "GAVIKO_HIP_ABLATE=loc_nosmall"
run "nowin" "GAVIKO_HIP_ABLATE=nowin"
run "nowin,loc_noouter" "GAVIKO_HIP_ABLATE=nowin,loc_noouter"
run "nowin,loc_noouter,loc_noupdown" "GAVIKO_HIP_ABLATE=nowin,loc_noouter,loc_noupdown"
run "locnop" "GAVIKO_HIP_ABLATE=locnop"
