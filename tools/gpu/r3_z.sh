#!/bin/bash
# one box, interleaved: where the round's wall-clock gain is (round-2 tree | product | diag default | diag with single round-3 changes undone)
run() { echo -n "$1: "; shift; "$@" python bench.py --allow-diag --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'; }
for i in 1 2 3; do
  echo -n "round-2 tree: "; (cd _r2 && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*')
  run "product" env
  run "diag default" env GAVIKO_HIP_DIAG=1
  run "diag row sums on matrix pipe" env GAVIKO_HIP_DIAG=1 GAVIKO_HIP_ATTN_VAR=1
  run "diag no LN1 fold" env GAVIKO_HIP_DIAG=1 GAVIKO_HIP_FOLD_LN1=0
  run "diag group_m 8" env GAVIKO_HIP_DIAG=1 GAVIKO_HIP_GEMM_GROUP_M=8
  run "diag key tile 128" env GAVIKO_HIP_DIAG=1 GAVIKO_HIP_ATTN_KB=128
done
