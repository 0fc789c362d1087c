#!/bin/bash
O=gpurun_out/r3sk; mkdir -p $O
ONLY=out,fc2,fc1_dgrad,out_dgrad,qkv_dgrad TILES=3128128,5128128 timeout -k 10 120 python tools/bench_gemm.py 4132 2>/dev/null
run() { echo -n "$1: "; shift; env GAVIKO_HIP_DIAG=1 "$@" timeout -k 10 200 python bench.py --allow-diag --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'; }
for i in 1 2 3; do
  run "stream-K off" GAVIKO_HIP_GEMM_SK=0
  run "stream-K on " GAVIKO_HIP_GEMM_SK=1
done
for v in 0 1 2; do echo -n "cfg5 SK=$v: "; env GAVIKO_HIP_DIAG=1 GAVIKO_HIP_GEMM_SK=$v timeout -k 10 200 python bench.py --allow-diag --backbone vit-l16 --batch 2 --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'; done
for v in 0 1 2; do echo -n "B=2 SK=$v: "; env GAVIKO_HIP_DIAG=1 GAVIKO_HIP_GEMM_SK=$v timeout -k 10 200 python bench.py --allow-diag --batch 2 --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'; done
