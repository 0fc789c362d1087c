#!/usr/bin/env python3
"""How much host time does issuing one step take (plan replay calls vs the whole Python step), without any sync in between?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaviko_amd import lib as L
from gaviko_amd.utils import synth

dev = torch.device("cuda:0")
model = bench.build("vit-b16", dev)
x = torch.from_numpy(synth.volumes(0, 4)).to(dev); y = torch.from_numpy(synth.labels(0, 4)).to(dev)
lib = L.load()
orig = lib.gvk_plan_replay
acc = {"t": 0.0, "n": 0}
class Wrap:
    def __call__(self, pid):
        t0 = time.perf_counter(); rc = orig(pid); acc["t"] += time.perf_counter() - t0; acc["n"] += 1; return rc
lib.gvk_plan_replay = Wrap()
def step():
    for p in model.parameters(): p.grad = None
    torch.nn.functional.cross_entropy(model(x), y).backward()
for _ in range(6): step()
torch.cuda.synchronize()
for n in (1, 2, 4, 8, 20):
    acc["t"] = 0.0; acc["n"] = 0
    t0 = time.perf_counter()
    for _ in range(n): step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{n:3d} steps: host issue {t_issue / n * 1e3:6.2f} ms/step (in replay {acc['t'] / n * 1e3:5.2f} ms, {acc['n'] // n} replays), wall {t_all / n * 1e3:6.2f} ms/step")
