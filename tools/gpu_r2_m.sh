#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2m; mkdir -p $O
export TMPDIR=/tmp
for cfg in "vit-b16 2" "vit-b16 8" "vit-l16 2" "vit-t16 4"; do set -- $cfg
  echo "== $1 B=$2"; python bench.py --steps 30 --warmup 10 --backbone $1 --batch $2 --no-cpu-baseline --no-roofline 2>/dev/null | tee $O/bench_$1_$2.json | cut -c1-160
done
rocprofv3 --kernel-trace --stats -d $O/prof -o r02 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_bench.log 2>&1
tail -1 $O/prof_bench.log | cut -c1-300
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python3 tools/timeline.py $O/prof 30 > $O/timeline.txt 2>&1 || true
rm -rf $O/prof
head -60 $O/kernel_stats.csv | cut -c1-150
