#!/bin/bash
# The GPU-box command sequences behind profiles/ and DESIGN.md, as ONE script:  gpurun -- 'bash tools/gpu/run.sh <what> [args]'
#   ab [names..]      headline bench + timing ablations (diag build; results of ablated runs are garbage, only the rate is read)
#   final <tag>       the measurement set of a round: PMC traffic -> bench lines -> kernel trace (stats + per shape) -> cfg5 -> plan marks
#   libs a.so b.so..  interleaved bench of several builds of the library (GAVIKO_HIP_LIB), three rounds
#   trace [bench args] rocprofv3 --kernel-trace --stats of the bench command -> per-kernel and per-shape tables
#   tests [expr]      pytest -m gpu (optionally -k expr), output under gpurun_out/
#   pmc <tag>         SQ counters of the GEMM and attention kernels (rocprofv3 --pmc, kernel trace only) + isolated GEMMs beside the vendor library
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
what=$1; shift || true
case "$what" in
ab)
  O=$R/gpurun_out/ab; mkdir -p $O
  run() { echo -n "$1: "; shift; env GAVIKO_HIP_DIAG=1 "$@" python bench.py --allow-diag --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'; }
  abl=${*:-noside sidenop locnop gpanop noparams nowin loc_noupdown nowait}
  for i in 1 2; do
    run default X=1
    for a in $abl; do run "$a" GAVIKO_HIP_ABLATE=$a; done
  done | tee $O/abl.txt
  ;;
final)
  tag=${1:-r05}; O=$R/gpurun_out/$tag; mkdir -p $O
  cd /tmp && export TMPDIR=/tmp
  echo "[1] PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, kernel trace only)"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
  cd $R
  python bench.py --steps 2 --warmup 2 --no-cpu-baseline --dump-gemm-shapes $O/gemm_shapes.json > /dev/null 2> $O/shapes.err
  python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/${tag}_pmc_traffic.json $O/gemm_shapes.json > $O/pmc_traffic.log 2>&1
  cp $O/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json        # bench.py reads it from profiles/ in step [2]
  rm -rf $O/pmc_fetch $O/pmc_write
  echo "[2] bench (default flags, then 30/10)"
  python bench.py > $O/bench_default.json 2> $O/bench_default.err
  python bench.py --steps 30 --warmup 10 > $O/${tag}_bench.json 2> $O/bench.err
  cut -c1-300 $O/${tag}_bench.json
  echo "[3] kernel trace"
  cd /tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace.log 2>&1
  cd $R
  cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/${tag}_bench_kernel_stats.csv
  python tools/kernel_stats_by_shape.py $O/trace --out $O/${tag}_kernel_stats_by_shape.csv 2> $O/by_shape.err
  rm -rf $O/trace
  echo "[4] cfg5"
  python bench.py --backbone vit-l16 --batch 2 --steps 30 --warmup 10 > $O/${tag}_bench_cfg5.json 2> $O/bench_cfg5.err
  cut -c1-300 $O/${tag}_bench_cfg5.json
  cd /tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace5 -- python3 $R/bench.py --backbone vit-l16 --batch 2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace5.log 2>&1
  cd $R
  python tools/kernel_stats_by_shape.py $O/trace5 --out $O/${tag}_kernel_stats_by_shape_cfg5.csv --M 2066 --C 1024 --mlp 4096 2>> $O/by_shape.err
  rm -rf $O/trace5
  echo "[5] marks + other batches"
  timeout -k 10 200 python tools/plan_marks.py 4 vit-b16 > $O/marks_cfg2.txt 2>&1
  for b in 2 8; do python bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200; done
  ;;
libs)
  # interleaved A/B of library builds on one box: run.sh libs gaviko_amd/libgaviko_hip.so gaviko_amd/libgaviko_hip_b.so ...
  O=$R/gpurun_out/libs; mkdir -p $O
  for i in 1 2 3; do
    for l in "$@"; do echo -n "$l: "; env GAVIKO_HIP_LIB=$R/$l python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'; done
  done | tee $O/libs.txt
  ;;
trace)
  # kernel trace of the bench command -> per-kernel stats and the per-shape table (no PMC): gpurun_out/trace/
  O=$R/gpurun_out/trace; mkdir -p $O
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $O/trace.log 2>&1
  cd $R
  cp $(find $O/raw -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
  python tools/kernel_stats_by_shape.py $O/raw --out $O/by_shape.csv 2> $O/by_shape.err
  rm -rf $O/raw
  head -40 $O/kernel_stats.csv | cut -c1-160
  ;;
tests)
  O=$R/gpurun_out/tests; mkdir -p $O
  if [ -n "$1" ]; then python -m pytest tests -m gpu -x -q -k "$1" > $O/pytest.log 2>&1; else python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; fi
  tail -5 $O/pytest.log
  ;;
pmc)
  tag=${1:-r04}; O=$R/gpurun_out/$tag; mkdir -p $O
  C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY"
  A1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
  A2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  export PMC_GEMM_ORDER=$O/pmc_gemm_order.json
  cd /tmp && export TMPDIR=/tmp
  echo "[1] GEMM SQ counters (product tile choice, tile=0)"
  timeout -k 10 300 rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $O/pg_a -- python3 $R/tools/pmc_gemm.py run 0 > $O/pg_a.log 2>&1
  python3 $R/tools/pmc_gemm.py sum $O/pg_a $O/${tag}_pmc_gemm_sq_counters.json > $O/pg_sum.log 2>&1
  echo "[2] attention SQ counters"
  timeout -k 10 300 rocprofv3 --pmc $A1 --kernel-trace --output-format csv -d $O/pa_a -- python3 $R/tools/pmc_attn.py run > $O/pa_a.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc $A2 --kernel-trace --output-format csv -d $O/pa_b -- python3 $R/tools/pmc_attn.py run > $O/pa_b.log 2>&1
  python3 $R/tools/pmc_attn.py sum $O/pa_a $O/pa_b $O/${tag}_pmc_attention.json > $O/pa_sum.log 2>&1
  rm -rf $O/pg_a $O/pa_a $O/pa_b
  cd $R
  echo "[3] isolated GEMMs vs the vendor library"
  python3 tools/gemm_vs_lib.py $O/${tag}_gemm_isolated_vs_lib.csv
  ;;
*) echo "usage: run.sh ab|final|libs|trace|tests|pmc" >&2; exit 2 ;;
esac
