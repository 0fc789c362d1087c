"""GPU box: isolated timing of the bf16 attention kernels at the cfg2 shape (B=4, T=1033, H=12), 50 launches replayed from a plan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaviko_amd import lib, ops
lib.require_device()
dev = torch.device("cuda:0")
B, T, H = (int(a) for a in (sys.argv[1:4] + ["4", "1033", "12"][len(sys.argv) - 1:]))
inner = H * 64
qkv = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev); qkv[:B * T] = (torch.randn(B * T, 3 * inner, device=dev) * 0.7).bfloat16()
out, dout = ops.act_zeros(B * T, inner, torch.bfloat16, dev), ops.act_zeros(B * T, inner, torch.bfloat16, dev)
dout[:B * T] = torch.randn(B * T, inner, device=dev).bfloat16()
lse, delta = torch.empty(B * H * T, device=dev), torch.empty(B * H * T, device=dev)
dqkv = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev)


def t(name, fn, flops):
    for _ in range(3): fn()
    l = lib.load()
    lib.check(l.gvk_plan_begin(), "begin")
    for _ in range(50): fn()
    pid = l.gvk_plan_end()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    l.gvk_plan_replay(pid)
    e0.record(); l.gvk_plan_replay(pid); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{name:28s} {us:7.1f} us  {flops / us / 1e6:6.0f} TFLOP/s")
    l.gvk_plan_free(pid)


f = 4.0 * B * H * T * T * 64
t("attention_fwd", lambda: ops.attention_fwd(qkv, out, lse, B, T, H, 0.125), f)
t("attention_bwd (5 products)", lambda: ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, T, H, 0.125), 2.5 * f)
