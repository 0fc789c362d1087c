#!/usr/bin/env python3
"""cProfile of the host side of one training step (what the Python interpreter does while the GPU works)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaviko_amd.utils import synth
dev = torch.device("cuda:0")
model = bench.build("vit-b16", dev)
x = torch.from_numpy(synth.volumes(0, 4)).to(dev); y = torch.from_numpy(synth.labels(0, 4)).to(dev)
def step():
    for p in model.parameters(): p.grad = None
    torch.nn.functional.cross_entropy(model(x), y).backward()
for _ in range(6): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
