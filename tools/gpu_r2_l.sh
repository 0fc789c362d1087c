#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2l; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_dropout_gpu.py -x -q -k "attention or attn" > $O/t.log 2>&1; tail -3 $O/t.log
echo "== LDS-DMA"; PROBE_LIB=libprobe_attn_false.so python3 tools/probe/probe_attn.py 2>/dev/null
echo "== register staging"; PROBE_LIB=libprobe_attn_true.so python3 tools/probe/probe_attn.py 2>/dev/null | tee $O/attn_stamps_rs.txt
echo "== RS"; python3 tools/bench_attn.py 2>/dev/null
echo "== DMA"; GAVIKO_HIP_ATTN_RS=0 python3 tools/bench_attn.py 2>/dev/null
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c80-150
GAVIKO_HIP_ATTN_RS=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c80-150
