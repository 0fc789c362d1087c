#!/usr/bin/env python3
"""Step time with the data-parallel gradient reducer attached (world size 1: bucketed / segmented backward, no wire)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaviko_amd.utils import synth
dev = torch.device("cuda:0")
for label, lpb, mode in (("no reducer", None, None), ("events, 4 layers/bucket", 4, "events"), ("segments, 4 layers/bucket", 4, "segments"),
                         ("events, 12 layers/bucket", 12, "events"), ("segments, 12 layers/bucket", 12, "segments"), ("no reducer (again)", None, None),
                         ("events, 4 layers/bucket (again)", 4, "events")):
    model = bench.build("vit-b16", dev)
    if lpb:
        model.make_reducer(layers_per_bucket=lpb, mode=mode)
    x = torch.from_numpy(synth.volumes(0, 4)).to(dev); y = torch.from_numpy(synth.labels(0, 4)).to(dev)
    params = list(model.parameters())
    def step():
        for p in params: p.grad = None
        torch.nn.functional.cross_entropy(model(x), y).backward()
    for _ in range(8): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"{label:34s} {dt * 1e3:6.2f} ms/step  {4 / dt:6.1f} volumes/s")
