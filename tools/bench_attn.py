"""GPU box: isolated timing of the attention kernels at the cfg2 shape (B=4, T=1033, H=12)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaviko_amd import ops, lib
lib.require_device()
dev = torch.device("cuda:0")
B, T, H = 4, 1033, 12
inner = H * 64
M = B * T
qkv = ops.act_zeros(M, 3 * inner, torch.bfloat16, dev); qkv[:M] = torch.randn(M, 3 * inner, device=dev).bfloat16()
O = ops.act_zeros(M, inner, torch.bfloat16, dev)
dO = ops.act_zeros(M, inner, torch.bfloat16, dev); dO[:M] = torch.randn(M, inner, device=dev).bfloat16()
dq = ops.act_zeros(M, 3 * inner, torch.bfloat16, dev)
lse = torch.zeros(B, H, T, device=dev); delta = torch.zeros(B, H, T, device=dev)
fl = 4.0 * B * H * T * T * 64
def t(name, fn, flops):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{name:20s} {us:7.1f} us  {flops / us / 1e6:6.0f} TF", flush=True)
t("attention_fwd", lambda: ops.attention_fwd(qkv, O, lse, B, T, H, 0.125), fl)
t("attention_bwd", lambda: ops.attention_bwd(qkv, O, dO, lse, delta, dq, B, T, H, 0.125), 2.5 * fl)
