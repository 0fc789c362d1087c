#!/usr/bin/env python3
"""Where does the main stream spend a step?  Run with GAVIKO_HIP_PLAN_TIMING=1: the engine drops timestamped events ("marks")
on the main stream into the recorded plans; this prints the average time between consecutive marks, grouped by phase name.
Unlike rocprofv3 this adds no per-launch host overhead, so the stalls it shows are the real ones."""
import collections, ctypes, os, sys
os.environ.setdefault("GAVIKO_HIP_PLAN_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaviko_amd import lib as L
from gaviko_amd.utils import synth

dev = torch.device("cuda:0")
_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(_pos[0]) if len(_pos) > 0 else 4
model = bench.build(_pos[1] if len(_pos) > 1 else "vit-b16", dev)      # plan_marks.py [B [backbone]]
x = torch.from_numpy(synth.volumes(0, B)).to(dev); y = torch.from_numpy(synth.labels(0, B)).to(dev)
def step():
    for p in model.parameters(): p.grad = None
    torch.nn.functional.cross_entropy(model(x), y).backward()
for _ in range(12): step()
torch.cuda.synchronize()
eng, lib = model._engine(), L.load()
for pid, (tag, marks) in eng.plan_marks.items():
    agg, total = collections.OrderedDict(), 0.0
    ms = ctypes.c_float()
    for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
        L.check(lib.gvk_plan_event_elapsed(pid, e0, e1, ctypes.byref(ms)), "elapsed")
        key = n0.split(":")[1] + " -> " + n1.split(":")[1]
        agg.setdefault(key, []).append(ms.value * 1e3)
        total += ms.value * 1e3
    print(f"plan {pid} ({tag}): {total:.0f} us between first and last mark")
    for k, v in agg.items():
        print(f"   {k:22s} n={len(v):3d} avg {sum(v) / len(v):7.1f} us  min {min(v):7.1f} max {max(v):7.1f}")
    if "--per-layer" in sys.argv:                # every interval of the plan in order (layer numbers kept): where a single layer differs
        for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
            L.check(lib.gvk_plan_event_elapsed(pid, e0, e1, ctypes.byref(ms)), "elapsed")
            print(f"      {n0:14s} -> {n1:14s} {ms.value * 1e3:7.1f} us")
