#!/bin/bash
# LayerNorm 2 folded into fc1: kernel test, gaviko goldens, interleaved step A/B (diag library: GAVIKO_HIP_FOLD_LN2=0 / 1)
set -e
mkdir -p gpurun_out/r3r
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -s -m gpu -k "layernorm_folded or gemm_epilogues or eight_phase" > gpurun_out/r3r/kern.log 2>&1 || { tail -30 gpurun_out/r3r/kern.log; exit 1; }
grep -h "LN fold\|passed\|failed" gpurun_out/r3r/kern.log | cut -c1-200
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_model_dropout_gpu.py -q -m gpu -k "gaviko or dropout" > gpurun_out/r3r/model.log 2>&1 || { tail -40 gpurun_out/r3r/model.log; exit 1; }
tail -2 gpurun_out/r3r/model.log
export GAVIKO_HIP_DIAG=1
for i in 1 2 3; do
  for v in 0 1; do
    echo -n "FOLD_LN2=$v: "; GAVIKO_HIP_FOLD_LN2=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --allow-diag 2>/dev/null | grep -o '"value": [0-9.]*'
  done
done
for v in 0 1; do echo "== marks FOLD_LN2=$v"; GAVIKO_HIP_FOLD_LN2=$v timeout -k 10 200 python tools/plan_marks.py 4 vit-b16 2>/dev/null | grep -A5 "fwd"; done
for v in 0 1; do echo -n "cfg5 FOLD_LN2=$v: "; GAVIKO_HIP_FOLD_LN2=$v python bench.py --backbone vit-l16 --batch 2 --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --allow-diag 2>/dev/null | grep -o '"value": [0-9.]*'; done
