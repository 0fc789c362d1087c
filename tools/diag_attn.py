"""GPU box: attention backward, 96- against 128-row tiles and both against float64 autograd -- where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaviko_amd import lib, ops
lib.require_device()
dev = torch.device("cuda:0")
B, T, H = (int(a) for a in (sys.argv[1:4] + ["2", "1001", "3"][len(sys.argv) - 1:]))
inner = H * 64
g = torch.Generator().manual_seed(5)
qkv = (torch.randn(B, T, 3 * inner, generator=g) * 1.0).bfloat16().double().requires_grad_(True)
dO = torch.randn(B, T, inner, generator=g).bfloat16().double()
q, k, v = (t.reshape(B, T, H, 64).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
s = q @ k.transpose(-1, -2) * 0.125
o = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, T, inner)
o.backward(dO)
want = qkv.grad
Q = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev); Q[: B * T] = qkv.detach().reshape(B * T, -1).to(dev).bfloat16()
DO = ops.act_zeros(B * T, inner, torch.bfloat16, dev); DO[: B * T] = dO.reshape(B * T, -1).to(dev).bfloat16()
res = {}
for kb in (96, 128):
    os.environ["GAVIKO_HIP_ATTN_KB"] = str(kb)
    O = ops.act_zeros(B * T, inner, torch.bfloat16, dev)
    lse = torch.zeros((B, H, T), device=dev); delta = torch.zeros((B, H, T), device=dev)
    DQ = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev)
    ops.attention_fwd(Q, O, lse, B, T, H, 0.125)
    ops.attention_bwd(Q, O, DO, lse, delta, DQ, B, T, H, 0.125)
    torch.cuda.synchronize()
    res[kb] = (DQ[: B * T].view(B, T, 3 * inner).cpu().double(), O[: B * T].view(B, T, inner).cpu().double(), lse.cpu().double(), delta.cpu().double())
for kb in (96, 128):
    got = res[kb][0]
    for name, sl in (("dq", slice(0, inner)), ("dk", slice(inner, 2 * inner)), ("dv", slice(2 * inner, 3 * inner))):
        d = (got[..., sl] - want[..., sl]).abs()
        rowerr = d.amax(-1)                                  # [B, T]
        worst = rowerr.flatten().topk(5)
        print(f"KB={kb} {name}: max err {d.max().item():.3e} (ref max {want[..., sl].abs().max().item():.3e}), mean err {d.mean().item():.3e}; worst rows (b*T+t): {[(int(i) // T, int(i) % T) for i in worst.indices]} {[f'{x:.2e}' for x in worst.values.tolist()]}")
a, b_ = res[96], res[128]
print("96 vs 128: dqkv max diff", (a[0] - b_[0]).abs().max().item(), " O diff", (a[1] - b_[1]).abs().max().item(), " lse diff", (a[2] - b_[2]).abs().max().item(), " delta diff", (a[3] - b_[3]).abs().max().item())
d = (a[0] - b_[0]).abs().amax(-1)
worst = d.flatten().topk(8)
print("rows with the largest 96/128 difference:", [(int(i) // T, int(i) % T, f"{x:.2e}") for i, x in zip(worst.indices, worst.values.tolist())])
