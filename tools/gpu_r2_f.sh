#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
python -m pytest tests/test_sidepath_kernels_gpu.py tests/test_kernels_gpu.py -x -q > $O/test_side.log 2>&1; rc=$?
tail -5 $O/test_side.log
[ $rc -ne 0 ] && exit $rc
python -m pytest tests/test_sidepath_kernels_gpu.py -x -q -k "bwd_up or proj" > $O/t_new.log 2>&1; tail -3 $O/t_new.log
echo "== new side kernels"; python3 tools/bench_side.py > $O/side_new.log 2>&1; cat $O/side_new.log
echo "== old (GAVIKO_HIP_SIDE=0)"; GAVIKO_HIP_SIDE=0 python3 tools/bench_side.py > $O/side_old.log 2>&1; head -12 $O/side_old.log
python -m pytest tests -x -q -m gpu > $O/test.log 2>&1; rc=$?
tail -5 $O/test.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-400 $O/bench.json
GAVIKO_HIP_SIDE=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench_oldside.json 2> $O/bench_oldside.err
cut -c1-300 $O/bench_oldside.json
