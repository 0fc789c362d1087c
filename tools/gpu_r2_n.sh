#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2n; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > $O/prof_bench.log 2>&1
grep '^{' $O/prof_bench.log | cut -c1-200
mkdir -p $O/kt/x; find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kt/x/kernel_trace.csv \;
rm -rf $O/prof
python3 tools/timeline.py $O/kt 30 5 > $O/timeline.txt 2>&1 || true
