#!/bin/bash
O=gpurun_out/r3x; mkdir -p $O
python tools/host_split.py 4 2>&1 | tail -4
python tools/host_split.py 2 2>&1 | tail -4
echo "== graph mode"
GAVIKO_HIP_GRAPHS=graph python tools/host_split.py 4 > $O/graph.txt 2>&1; tail -5 $O/graph.txt | cut -c1-300
nproc; lscpu | grep -i "model name\|MHz" | head -4
