#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3k; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "patch_embed" > $O/t.log 2>&1; tail -2 $O/t.log
run() { echo -n "$1: "; env $2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['patch_embed']['avg_us'], d['patch_embed']['frac'])"; }
run "implicit patch GEMM" "GAVIKO_HIP_PATCH_IMPLICIT=1"
run "im2col + GEMM" "X=1"
run "implicit patch GEMM" "GAVIKO_HIP_PATCH_IMPLICIT=1"
