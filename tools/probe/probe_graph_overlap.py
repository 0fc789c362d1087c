#!/usr/bin/env python3
"""Does a captured HIP graph with forked branches overlap them the way the capture's event edges say?
Mimics the backward layer boundary: main chain M (big GEMMs), side chain S (many small kernels) forked from every M_k,
main waits on an EARLY event of the side chain (like dz_ready) while the side chain continues.  Compares graph replay,
eager multi-stream and the two single chains alone."""
import time
import torch

dev = torch.device("cuda:0")
a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
c = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
small = [torch.randn(1 << 23, device=dev) for _ in range(4)]
side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
LAYERS, NS_EARLY, NS_LATE = 12, 2, 5


def big():
    torch.mm(a, b, out=c)


def tiny(k):
    small[k % 4].mul_(1.0001)


def body(mode):
    cur = torch.cuda.current_stream()
    if mode == "main_only":
        for _ in range(LAYERS):
            big(); big()
        return
    if mode == "side_only":
        for _ in range(LAYERS):
            for k in range(NS_EARLY + NS_LATE):
                tiny(k)
        return
    e = torch.cuda.Event(); e.record(cur); side.wait_event(e)
    for _ in range(LAYERS):
        with torch.cuda.stream(side):
            for k in range(NS_EARLY):
                tiny(k)
            early = torch.cuda.Event(); early.record(side)
            for k in range(NS_LATE):
                tiny(k)
        big()
        cur.wait_event(early)
        big()
        e = torch.cuda.Event(); e.record(cur); side.wait_event(e)     # next layer's side chain needs this layer's main result
    e = torch.cuda.Event(); e.record(side); cur.wait_event(e)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for mode in ("main_only", "side_only", "forked"):
    eager = timeit(lambda: body(mode))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body(mode)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            body(mode)
    graph = timeit(g.replay)
    print(f"{mode:10s} eager {eager:8.1f} us   graph {graph:8.1f} us")
