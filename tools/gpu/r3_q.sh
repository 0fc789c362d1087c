#!/bin/bash
# cfg5 (ViT-L/16, B = 2): where the main stream spends the step, per-shape kernel stats
set -e
mkdir -p gpurun_out/r3q
timeout -k 10 300 python tools/plan_marks.py 2 vit-l16 > gpurun_out/r3q/marks_cfg5.txt 2>&1
timeout -k 10 300 python tools/plan_marks.py 4 vit-b16 > gpurun_out/r3q/marks_cfg2.txt 2>&1
for i in 1 2; do python bench.py --backbone vit-l16 --batch 2 --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3q/trace_cfg5 -- python3 $GRAFT_REPO_ROOT/bench.py --backbone vit-l16 --batch 2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r3q/prof_cfg5.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r3q/trace_cfg5 -name "*kernel_trace.csv" | head -1)
python tools/kernel_stats_by_shape.py $f > gpurun_out/r3q/cfg5_by_shape.csv 2> gpurun_out/r3q/by_shape.err || true
rm -rf gpurun_out/r3q/trace_cfg5
head -40 gpurun_out/r3q/cfg5_by_shape.csv | cut -c1-170
