#!/bin/bash
# timing ablations on the round-3 build (diag library; results are garbage, only the step time is read)
run() { echo -n "$1: "; shift; env GAVIKO_HIP_DIAG=1 "$@" python bench.py --allow-diag --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'; }
for i in 1 2; do
  run "default" X=1
  for a in nowait noside sidenop locnop gpanop noparams nowin loc_noupdown; do run "$a" GAVIKO_HIP_ABLATE=$a; done
done
echo "== round-2 tree"
for a in "" noside locnop; do echo -n "r2 ${a:-default}: "; (cd _r2 && env GAVIKO_HIP_ABLATE=$a python bench.py --allow-ablate --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'); done
