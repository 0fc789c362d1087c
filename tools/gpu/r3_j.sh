#!/bin/bash
# round 3, call j: GEMM rasterisation groups sized to an XCD's run -- tests, isolated GEMM timing, step A/B (diag build: group_m 8 = round 2), PMC traffic
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r3j
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -q -k "gemm" > gpurun_out/r3j/test_gemm.log 2>&1; echo "gemm tests rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/r3j/test_gemm.log | tail -3
run() { echo -n "$1: "; env GAVIKO_HIP_DIAG=1 $2 timeout -k 10 300 python bench.py --allow-diag --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*' ; }
for k in 1 2 3; do
run "group_m auto" "X=1"
run "group_m 8 (round 2)" "GAVIKO_HIP_GEMM_GROUP_M=8"
done
# PMC traffic of the product build (two passes, counters only)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r3j/pmc_fetch -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r3j/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r3j/pmc_write -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r3j/pmc_write.log 2>&1; echo "pmc write rc=$?"
python3 tools/pmc_traffic.py gpurun_out/r3j/pmc_fetch gpurun_out/r3j/pmc_write gpurun_out/r3j/pmc_traffic.json > gpurun_out/r3j/pmc_traffic.log 2>&1; echo "pmc_traffic rc=$?"; tail -3 gpurun_out/r3j/pmc_traffic.log
rm -rf gpurun_out/r3j/pmc_fetch gpurun_out/r3j/pmc_write
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r3j/pmc_traffic.json"))
for k,v in d["kernels"].items():
    if "gemm" in k: print(k[:60], v.get("clusters") or v["total_bytes"])
PY
