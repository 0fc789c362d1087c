#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3j; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "attention or attn" > $O/t.log 2>&1; tail -2 $O/t.log
python3 tools/bench_attn.py 2>/dev/null | tail -8
