#!/bin/bash
# same-box A/B: the round-2 tree (git archive of b6c81bd extracted to _r2/, its own library) against this tree, interleaved
set -e
for i in 1 2 3 4; do
  echo -n "round-2 tree: "; (cd _r2 && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*')
  echo -n "round-3 tree: "; python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'
done
for cfg in "--backbone vit-l16 --batch 2" "--batch 8" "--batch 2"; do
  echo -n "round-2 tree $cfg: "; (cd _r2 && python bench.py $cfg --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*')
  echo -n "round-3 tree $cfg: "; python bench.py $cfg --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'
done
