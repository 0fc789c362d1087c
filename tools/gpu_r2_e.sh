#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json | cut -c1-600
GAVIKO_HIP_GEMM_WIDE=256 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_old.json 2> $O/bench_old.err
cat $O/bench_old.json | cut -c1-300
GAVIKO_HIP_ABLATE=noside python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench_noside.json 2> $O/bench_noside.err
cat $O/bench_noside.json | cut -c1-300
python - <<'PY' > $O/bitfit.log 2>&1
import sys, os
sys.path.insert(0, "examples"); sys.path.insert(0, ".")
import train_synthetic as ts
for ep, lr in ((4, 2e-3), (6, 5e-3)):
    for seed in (0, 1, 2):
        r = ts.run(method="bitfit", backbone="vit-t16", epochs=ep, samples=8, out=f"/tmp/bf{seed}_{ep}", batch_size=4, lr=lr, seed=seed, log=lambda *a: None)
        print(ep, lr, seed, [round(e["train_loss"], 4) for e in r["history"]], flush=True)
PY
cat $O/bitfit.log | tail -8
