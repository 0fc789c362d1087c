#!/usr/bin/env python3
"""Host time of one training step, split: model() call | of which engine.forward | of which plan replay || loss || loss.backward() | of which
the autograd node's backward | engine.backward | plan replay.  No sync inside the timed loop (the host runs ahead of the GPU)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaviko_amd import lib as L
from gaviko_amd.model import vision_transformer as vt
from gaviko_amd.utils import synth

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model = bench.build(sys.argv[2] if len(sys.argv) > 2 else "vit-b16", dev)
x = torch.from_numpy(synth.volumes(0, B)).to(dev); y = torch.from_numpy(synth.labels(0, B)).to(dev)
from gaviko_amd.losses import CrossEntropyLoss, StepMeter
crit = CrossEntropyLoss().attach_meter(StepMeter(dev))
acc = {}
def timed(obj, name, key):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, w)
lib = L.load()
timed(lib, "gvk_plan_replay", "replay")
eng = model._engine()
timed(eng, "forward", "eng.forward")
timed(eng, "backward", "eng.backward")
ob = vt._HotPathFn.backward
def nb(ctx, g):
    t0 = time.perf_counter(); r = ob(ctx, g); acc["node.backward"] = acc.get("node.backward", 0.0) + time.perf_counter() - t0; return r
vt._HotPathFn.backward = staticmethod(nb)
params = list(model.parameters())
def step():
    t0 = time.perf_counter()
    for p in params: p.grad = None
    t1 = time.perf_counter(); out = model(x)
    t2 = time.perf_counter(); loss = crit(out, y)
    t3 = time.perf_counter(); loss.backward()
    t4 = time.perf_counter()
    for k, v in (("zero", t1 - t0), ("model()", t2 - t1), ("loss", t3 - t2), ("loss.backward()", t4 - t3)): acc[k] = acc.get(k, 0.0) + v
for _ in range(8): step()
torch.cuda.synchronize()
for n in (1, 10, 40):
    acc.clear()
    t0 = time.perf_counter()
    for _ in range(n): step()
    ti = time.perf_counter() - t0
    torch.cuda.synchronize()
    ta = time.perf_counter() - t0
    print(f"{n:3d} steps: host {ti / n * 1e3:.2f} ms/step, wall {ta / n * 1e3:.2f} ms/step | " + "  ".join(f"{k} {v / n * 1e3:.2f}" for k, v in acc.items()))
