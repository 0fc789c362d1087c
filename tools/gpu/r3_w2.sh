#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R/_r2 && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_r2 -- python3 $R/_r2/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace_r2.log 2>&1)
(cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_r3 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace_r3.log 2>&1)
cd $R
python tools/kernel_stats_by_shape.py $O/trace_r2 --out $O/by_shape_r2.csv 2> $O/by_shape.err
python tools/kernel_stats_by_shape.py $O/trace_r3 --out $O/by_shape_r3.csv 2>> $O/by_shape.err
rm -rf $O/trace_r2 $O/trace_r3
