#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/test.log 2>&1; rc=$?
tail -3 $O/test.log
[ $rc -ne 0 ] && exit $rc
cp gpurun_out/parity_report.txt $O/parity_report.txt 2>/dev/null
for cfg in "vit-b16 2" "vit-b16 8" "vit-l16 2" "vit-t16 4"; do set -- $cfg
  echo -n "$1 B=$2: "; python bench.py --steps 30 --warmup 10 --backbone $1 --batch $2 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 10, "ms_per_step": [0-9.]*'
done
python3 tools/bench_side.py > $O/bench_side.txt 2>/dev/null; grep -i "win\|side\|skinny\|outer" $O/bench_side.txt | head -30
python3 tools/bench_reducer.py > $O/bench_reducer.txt 2>&1; tail -6 $O/bench_reducer.txt
