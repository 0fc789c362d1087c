#!/bin/bash
# round 3 final measurements (product build): PMC traffic -> bench line -> kernel trace (stats + per-shape) -> cfg5 line + per-shape -> marks
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[1] PMC passes"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
cd $R
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write profiles/r03_pmc_traffic.json > $O/pmc_traffic.log 2>&1
cp profiles/r03_pmc_traffic.json $O/r03_pmc_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
echo "[2] bench (default flags, then 30/10)"
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 30 --warmup 10 > $O/r03_bench.json 2> $O/bench.err
cut -c1-300 $O/r03_bench.json
echo "[3] kernel trace"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace.log 2>&1
cd $R
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats.csv
python tools/kernel_stats_by_shape.py $O/trace --out $O/r03_kernel_stats_by_shape.csv 2> $O/by_shape.err
rm -rf $O/trace
echo "[4] cfg5"
python bench.py --backbone vit-l16 --batch 2 --steps 30 --warmup 10 > $O/r03_bench_cfg5.json 2> $O/bench_cfg5.err
cut -c1-300 $O/r03_bench_cfg5.json
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace5 -- python3 $R/bench.py --backbone vit-l16 --batch 2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/trace5.log 2>&1
cd $R
python tools/kernel_stats_by_shape.py $O/trace5 --out $O/r03_kernel_stats_by_shape_cfg5.csv --M 2066 --C 1024 --mlp 4096 2>> $O/by_shape.err
rm -rf $O/trace5
echo "[5] marks + other batches"
timeout -k 10 200 python tools/plan_marks.py 4 vit-b16 > $O/marks_cfg2.txt 2>&1
for b in 2 8; do python bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200; done
