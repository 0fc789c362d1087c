#!/usr/bin/env python3
"""GPU box: where does a step's wall time go BETWEEN the two launch plans?  torch events on the main stream around the forward call, the loss
and the backward call of the bench's own step (no sync inside the loop): forward / loss / backward / boundary to the next step."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaviko_amd.utils import synth
from gaviko_amd.losses import CrossEntropyLoss, StepMeter

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model = bench.build(sys.argv[2] if len(sys.argv) > 2 else "vit-b16", dev)
eng = model._engine()
x = eng.input_buffer(B, dev, train=True); x.copy_(torch.from_numpy(synth.volumes(0, B)))
y = torch.from_numpy(synth.labels(0, B)).to(dev)
eng.static_io = True
crit = CrossEntropyLoss().attach_meter(StepMeter(dev))
N = 40
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(N)]
def step(e=None):
    model.zero_grad(set_to_none=False)
    if e: e[0].record()
    out = model(x)
    if e: e[1].record()
    loss = crit(out, y)
    if e: e[2].record()
    loss.backward()
    if e: e[3].record()
for _ in range(10): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(N): step(ev[i])
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
f = sum(e[0].elapsed_time(e[1]) for e in ev[5:]) / (N - 5)
l = sum(e[1].elapsed_time(e[2]) for e in ev[5:]) / (N - 5)
b = sum(e[2].elapsed_time(e[3]) for e in ev[5:]) / (N - 5)
g = sum(ev[i][3].elapsed_time(ev[i + 1][0]) for i in range(5, N - 1)) / (N - 6)
print(f"wall {wall:.3f} ms/step | forward {f:.3f}  loss {l:.3f}  backward {b:.3f}  boundary (backward end -> next forward start, incl. the memset) {g:.3f}  sum {f + l + b + g:.3f}")
