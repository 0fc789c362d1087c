#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2g; mkdir -p $O
python3 tools/probe/probe_determinism.py > $O/det.log 2>&1; tail -25 $O/det.log
