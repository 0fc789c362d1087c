#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "eight_phase" > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -3 $O/test.log
ONLY=qkv,fc1,fc2_dgrad,fc1_plain16,fc1_plain32,big,fc2 TILES=8256256,7256256,256256,128128 python3 tools/bench_gemm.py > $O/bench_gemm.log 2>&1 || { tail -20 $O/bench_gemm.log; exit 1; }
cat $O/bench_gemm.log
export PMC_GEMM_ORDER=$O/order.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $O/pmc_a -- python3 tools/pmc_gemm.py run 8256256,7256256 > $O/pmc_a.log 2>&1 || { tail -20 $O/pmc_a.log; exit 1; }
python3 tools/pmc_gemm.py sum $O/pmc_a $O/pmc_gemm_a.json | tee $O/pmc_a_summary.txt
