#!/bin/bash
set -e
mkdir -p gpurun_out/r3s
export GAVIKO_HIP_DIAG=1
for i in 1 2 3; do
  for v in 0 1; do
    echo -n "FOLD_LN2=$v: "; GAVIKO_HIP_FOLD_LN2=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --allow-diag 2>/dev/null | grep -o '"value": [0-9.]*'
  done
done
for v in 0 1; do echo "== marks FOLD_LN2=$v"; GAVIKO_HIP_FOLD_LN2=$v timeout -k 10 200 python tools/plan_marks.py 4 vit-b16 2>/dev/null | grep -A5 "fwd"; done
for v in 0 1; do echo -n "cfg5 FOLD_LN2=$v: "; GAVIKO_HIP_FOLD_LN2=$v python bench.py --backbone vit-l16 --batch 2 --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --allow-diag 2>/dev/null | grep -o '"value": [0-9.]*'; done
