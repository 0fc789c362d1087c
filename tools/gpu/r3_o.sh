#!/bin/bash
# A/B of the forward attention's row-sum variant on the parity cases that moved: VAR=0 (VALU sums of unrounded P) vs VAR=1 (matrix-pipe sums of bf16 P)
set -e
mkdir -p gpurun_out/r3o
export GAVIKO_HIP_DIAG=1
K='melo_t16_b2_layers or cfg4_adaptformer or cfg3_deep_vpt_data or cfg4_melo or adaptformer_t16'
for v in 0 1; do
  GAVIKO_HIP_ATTN_VAR=$v timeout -k 10 500 python -m pytest tests/test_model_gpu.py -q -m gpu -s -k "$K" > gpurun_out/r3o/model_var$v.log 2>&1 || true
  GAVIKO_HIP_ATTN_VAR=$v timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -s -k "test_attention_fwd and not key_tiles" > gpurun_out/r3o/attn_var$v.log 2>&1 || true
done
grep -h "PARITY\|passed\|failed\|assert\|Error" gpurun_out/r3o/*.log | cut -c1-220 | tail -60
