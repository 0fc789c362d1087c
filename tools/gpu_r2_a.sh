#!/bin/bash
# round-2 GPU session A: new 8-phase GEMM -- parity, isolated timing, SQ counters of old and new kernels
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -3 $O/test.log
TILES=8256256,256256,3128128,128128 python3 tools/bench_gemm.py > $O/bench_gemm.log 2>&1 || { tail -20 $O/bench_gemm.log; exit 1; }
cat $O/bench_gemm.log
rocprofv3 -L > $O/counters_list.txt 2>&1
export PMC_GEMM_ORDER=$O/order.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $O/pmc_a -- python3 tools/pmc_gemm.py run > $O/pmc_a.log 2>&1 || { tail -20 $O/pmc_a.log; exit 1; }
python3 tools/pmc_gemm.py sum $O/pmc_a $O/pmc_gemm_a.json | tee $O/pmc_a_summary.txt
