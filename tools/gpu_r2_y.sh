#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2y; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/test.log 2>&1; rc=$?
tail -4 $O/test.log
[ $rc -ne 0 ] && exit $rc
cp gpurun_out/parity_report.txt $O/parity_report.txt 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/r02_pmc_traffic.json && cp $O/r02_pmc_traffic.json profiles/r02_pmc_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
python bench.py --steps 30 --warmup 10 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
tail -3 $O/bench.err
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(d.get('cpu_baseline')); print(d.get('roofline')); print(d.get('patch_embed'))"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
grep '^{' $O/kt.log | cut -c1-200
rm -rf $O/kt
