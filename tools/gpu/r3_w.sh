#!/bin/bash
# same-box comparison of the round-2 and round-3 trees: main-stream marks, per-shape kernel times, step time with the host-side split
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w; mkdir -p $O
(cd _r2 && timeout -k 10 200 python tools/plan_marks.py 4 > $O/marks_r2.txt 2>&1)
timeout -k 10 200 python tools/plan_marks.py 4 vit-b16 > $O/marks_r3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
(cd $R/_r2 && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_r2 -- python3 $R/_r2/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace_r2.log 2>&1)
cd $R
python tools/kernel_stats_by_shape.py $O/trace_r2 --out $O/by_shape_r2.csv 2> $O/by_shape.err
rm -rf $O/trace_r2
for t in _r2 .; do echo "== $t"; (cd $t && python tools/host_time.py 2>/dev/null | tail -8); done
grep -A 14 "plan 0" $O/marks_r2.txt; grep -A 14 "plan 0" $O/marks_r3.txt
