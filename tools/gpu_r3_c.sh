#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --allow-ablate 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
run "default" "X=1"
run "nowin" "GAVIKO_HIP_ABLATE=nowin"
run "loc_noupdown" "GAVIKO_HIP_ABLATE=loc_noupdown"
run "loc_noouter" "GAVIKO_HIP_ABLATE=loc_noouter"
run "loc_nosmall" "GAVIKO_HIP_ABLATE=loc_nosmall"
run "locnop" "GAVIKO_HIP_ABLATE=locnop"
run "gpanop" "GAVIKO_HIP_ABLATE=gpanop"
run "noparams" "GAVIKO_HIP_ABLATE=noparams"
run "sidenop" "GAVIKO_HIP_ABLATE=sidenop"
run "nowait" "GAVIKO_HIP_ABLATE=nowait"
run "default" "X=1"
