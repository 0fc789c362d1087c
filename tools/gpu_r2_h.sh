#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
tail -2 $O/kt.log | cut -c1-300
