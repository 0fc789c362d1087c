#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c; mkdir -p $O
python3 tools/probe/probe_gemm8p.py > $O/probe.log 2>&1 || { tail -30 $O/probe.log; exit 1; }
cat $O/probe.log
