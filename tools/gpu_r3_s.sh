#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for sw in "GAVIKO_HIP_FUSE_SCATTER=1" "GAVIKO_HIP_SIDE_LN=1" "GAVIKO_HIP_SIDE_LN=0" "GAVIKO_HIP_FIX_IN_LN=1" "GAVIKO_HIP_PATCH_IMPLICIT=1" "GAVIKO_HIP_FUSE_NEXT=0" "GAVIKO_HIP_FUSE_BOUNDARY=0" "GAVIKO_HIP_FUSE_UP=0" "GAVIKO_HIP_WIN_MFMA=0" "GAVIKO_HIP_SIDE=0" "GAVIKO_HIP_GEMM_WIDE=0" "GAVIKO_HIP_ATTN8=1" "GAVIKO_HIP_ATTN_RS=1" "GAVIKO_HIP_GPA_BWD_WAVES=2" "GAVIKO_HIP_LOC_SHIFT=1"; do
  echo -n "$sw: "; env $sw timeout -k 10 300 python -m pytest tests/test_model_gpu.py -x -q -k "gaviko_forward_backward_vs_golden or eval_forward" 2>&1 | tail -1
done
