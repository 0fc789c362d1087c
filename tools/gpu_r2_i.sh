#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2i; mkdir -p $O
python -m pytest tests/test_model_gpu.py -q -k "not fp32" > $O/test.log 2>&1; tail -4 $O/test.log
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
python bench.py --steps 30 --warmup 10 > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
cut -c1-300 $O/bench.json; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d.get('cpu_baseline')); print(d.get('roofline'))"
