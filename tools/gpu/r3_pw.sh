#!/bin/bash
# clocks / power while the step runs back to back (is the chip power-limited?)
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|power\|temp" | head -12
python bench.py --steps 1500 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/pw_bench.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3 4; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|power\|junction\|edge" | head -8; echo --; sleep 1.5; done
wait $BP
grep -o '"value": [0-9.]*' gpurun_out/pw_bench.json
