#!/bin/bash
run() { echo -n "$1: "; shift; env GAVIKO_HIP_DIAG=1 "$@" python bench.py --allow-diag --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'; }
for i in 1 2; do
  run "default" X=1
  run "noevents" GAVIKO_HIP_ABLATE=noevents
  run "nowait" GAVIKO_HIP_ABLATE=nowait
  run "noside" GAVIKO_HIP_ABLATE=noside
  run "noside,noevents" GAVIKO_HIP_ABLATE=noside,noevents
done
