#!/bin/bash
# round 3, call d: kernel split of the attention pair, step-level kernel trace (+ per call-site stats), bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r3d
timeout -k 10 300 python tools/bench_attn.py > gpurun_out/r3d/bench_attn.log 2>&1; echo "bench_attn rc=$?"; cat gpurun_out/r3d/bench_attn.log | grep -v amdgpu
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d/attn_prof -- python3 tools/bench_attn.py > gpurun_out/r3d/attn_prof.log 2>&1; echo "rocprof attn rc=$?"
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r3d/attn_prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d/step_prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r3d/step_prof.log 2>&1; echo "rocprof step rc=$?"; tail -1 gpurun_out/r3d/step_prof.log | cut -c1-300
python3 tools/kernel_stats_by_shape.py gpurun_out/r3d/step_prof --skip-first 3 --out gpurun_out/r3d/kernel_stats_by_shape.csv
cp $(ls gpurun_out/r3d/step_prof/*/*kernel_stats.csv | head -1) gpurun_out/r3d/bench_kernel_stats.csv
rm -rf gpurun_out/r3d/step_prof/*/*kernel_trace.csv gpurun_out/r3d/attn_prof/*/*kernel_trace.csv
timeout -k 10 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err; echo "bench rc=$?"; cat gpurun_out/r3d/bench.json | cut -c1-1500
