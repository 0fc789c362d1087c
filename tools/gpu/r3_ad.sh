#!/bin/bash
for i in 1 2 3; do
  echo -n "round-2 tree: "; (cd _r2 && python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*')
  echo -n "this tree:    "; python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"value": [0-9.]*'
done
