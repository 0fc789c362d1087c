#!/bin/bash
# side-path clean-up (reduce_batch slabs per job, ln_lowrank_affine loads in flight): tests, then the step against the round-2 tree as the fixed reference
set -e
O=gpurun_out/r3ac; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_sidepath_kernels_gpu.py -q -m gpu -x > $O/side.log 2>&1 || { tail -30 $O/side.log; exit 1; }
tail -1 $O/side.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_model_dropout_gpu.py -q -m gpu -k "gaviko or dropout or dvpt or evp" > $O/model.log 2>&1 || { tail -30 $O/model.log; exit 1; }
tail -1 $O/model.log
for i in 1 2 3; do
  echo -n "round-2 tree: "; (cd _r2 && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*')
  echo -n "this tree:    "; python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'
done
echo -n "cfg5: "; python bench.py --backbone vit-l16 --batch 2 --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'
