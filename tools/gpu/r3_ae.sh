#!/bin/bash
set -e
O=gpurun_out/r3ae; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_sidepath_kernels_gpu.py -q -m gpu -x -k "outer or reduce or mwsa or gpa" > $O/side.log 2>&1 || { tail -30 $O/side.log; exit 1; }
tail -1 $O/side.log
python tools/bench_side.py 2>/dev/null | grep -i "outer"
(cd _r2 && python tools/bench_side.py 2>/dev/null | grep -i "outer" | sed 's/^/r2: /')
for i in 1 2 3; do
  echo -n "round-2 tree: "; (cd _r2 && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*')
  echo -n "this tree:    "; python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'
done
