"""GPU box: isolated timing of the bf16 attention kernels at the cfg2 shape (B=4, T=1033, H=12).  Every variant is a launch plan of 50
launches; the plans are replayed in interleaved rounds inside ONE process (min / median over rounds).
usage: python tools/bench_attn.py [B T H] [--fwd-variants]"""
import os, statistics, sys
os.environ.setdefault("GAVIKO_HIP_DIAG", "1")      # kernel variants / A/B switches live in the measurement build (python -m gaviko_amd.build --diag), statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaviko_amd import lib, ops
lib.require_device()
dev = torch.device("cuda:0")
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
B, T, H = (int(a) for a in (pos + ["4", "1033", "12"][len(pos):]))
inner = H * 64
qkv = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev); qkv[:B * T] = (torch.randn(B * T, 3 * inner, device=dev) * 0.7).bfloat16()
ops.qkv_prescale(qkv, B * T, H, 0.125)                  # the operand form the engine hands the kernels
out, dout = ops.act_zeros(B * T, inner, torch.bfloat16, dev), ops.act_zeros(B * T, inner, torch.bfloat16, dev)
dout[:B * T] = torch.randn(B * T, inner, device=dev).bfloat16()
lse, delta = torch.empty(B * H * T, device=dev), torch.empty(B * H * T, device=dev)
dqkv = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev)
l = lib.load()
NL = 50


def plan(fn, env=None):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        for _ in range(3): fn()
        lib.check(l.gvk_plan_begin(), "begin")
        for _ in range(NL): fn()
        pid = l.gvk_plan_end()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    torch.cuda.synchronize()
    return pid


def race(cases, flops, rounds=7):
    """cases: [(name, plan id)] -> interleaved rounds"""
    t = {n: [] for n, _ in cases}
    for _ in range(rounds):
        for n, pid in cases:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            l.gvk_plan_replay(pid)
            e0.record(); l.gvk_plan_replay(pid); e1.record(); torch.cuda.synchronize()
            t[n].append(e0.elapsed_time(e1) * 1e3 / NL)
    for n, _ in cases:
        mn, md = min(t[n]), statistics.median(t[n])
        print(f"{n:36s} min {mn:7.2f} us  median {md:7.2f} us  {flops[n] / md / 1e6:6.0f} TFLOP/s", flush=True)


f = 4.0 * B * H * T * T * 64
fwd = lambda: ops.attention_fwd(qkv, out, lse, B, T, H, 0.125, q_prescaled=True)
bwd = lambda: ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, T, H, 0.125, q_prescaled=True)
cases, flops = [], {}
if "--fwd-variants" in sys.argv:
    for kb in (96, 128):
        for var in (0, 1, 2, 3):
            n = f"fwd KB={kb} VAR={var}"
            cases.append((n, plan(fwd, {"GAVIKO_HIP_ATTN_KB": str(kb), "GAVIKO_HIP_ATTN_VAR": str(var)})))
            flops[n] = f
cases.append(("attention_fwd (default)", plan(fwd))); flops[cases[-1][0]] = f
cases.append(("attention_bwd (5 products)", plan(bwd))); flops[cases[-1][0]] = 2.5 * f
race(cases, flops)
