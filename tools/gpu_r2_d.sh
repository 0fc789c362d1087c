#!/bin/bash
# full GPU test-suite + bench with the new default GEMM dispatch
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/test.log 2>&1; rc=$?
tail -5 $O/test.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 30 --warmup 10 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json
GAVIKO_HIP_GEMM_WIDE=256 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_old.json 2> $O/bench_old.err
cat $O/bench_old.json
