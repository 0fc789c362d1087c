#!/bin/bash
# round 3, call f: full GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3f
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r3f/test_all.log 2>&1; echo "gpu tests rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r3f/test_all.log | tail -15
cp gpurun_out/parity_report.txt gpurun_out/r3f/parity_report.txt
