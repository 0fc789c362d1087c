#!/bin/bash
# round 3, call g: other configurations (cfg5 = ViT-L B=2, ViT-B B=2 / B=8) with GEMM classes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3g
timeout -k 10 400 python bench.py --backbone vit-l16 --batch 2 --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/r3g/cfg5.json 2> gpurun_out/r3g/cfg5.err; echo "cfg5 rc=$?"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r3g/cfg5.json"))
print("cfg5", d["value"], d["ms_per_step"], d.get("mfma_roofline_frac_whole_step"))
for k,v in d["gemm_classes"].items(): print("  ",k,v)
PY
for cfg in "vit-b16 2" "vit-b16 8"; do set -- $cfg; echo -n "$1 B=$2: "; timeout -k 10 300 python bench.py --backbone $1 --batch $2 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|rror.*'; done
