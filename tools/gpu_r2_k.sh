#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2k; mkdir -p $O
for q in 1 2 4 8 16; do
  echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c80-200
done
echo "default"; python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c80-200
