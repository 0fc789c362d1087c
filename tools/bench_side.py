"""GPU box: isolated timing of the rank-L side-path kernels at the cfg2 shapes (M=4132, C=768, L=20)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaviko_amd import ops, lib
lib.require_device()
dev = torch.device("cuda:0")
M, C, L = 4132, 768, 20
f = lambda *s: torch.randn(*s, device=dev)
x, w, wt, b, g, bt = f(M, C), f(L, C) * 0.03, f(C, L) * 0.03, f(L), f(C), f(C)
lat, y, z, y2 = f(M, L), f(M, L), f(M, L), f(M, 3 * L)
mean, rstd = f(M), f(M).abs() + 0.5
w2 = f(3 * L, L)
out = f(M, C); res = f(M, C)
scratch = torch.zeros(ops.outer_scratch_elems(L, C), device=dev)
dW = torch.zeros(L, C, device=dev); dWt = torch.zeros(C, L, device=dev); cs = torch.zeros(C, device=dev)
out16 = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
qkv = f(4000, 60); ctx = f(4000, L); lse = f(4000); dctx = f(4000, L); delta = f(4000); dqkv = f(4000, 60)

def t(name, fn, bytes_):
    # 50 launches recorded into a launch plan and replayed from C: the Python wrappers cost more host time than these kernels run
    for _ in range(3): fn()
    l = lib.load()
    lib.check(l.gvk_plan_begin(), "begin")
    for _ in range(50): fn()
    pid = l.gvk_plan_end()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    l.gvk_plan_replay(pid)
    e0.record()
    l.gvk_plan_replay(pid)
    e1.record(); torch.cuda.synchronize()
    l.gvk_plan_free(pid)
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{name:34s} {us:7.1f} us   {bytes_ / us / 1e6:6.2f} TB/s algorithmic", flush=True)

pass_b = M * C * 4
t("skinny_down LN+qkv", lambda: ops.skinny_down(x=x, w=w, bias=b, ln_gamma=g, ln_beta=bt, mean=mean, rstd=rstd, y=y, w2=w2, y2=y2, M=M, C=C, L=L, L2=60, act=0, w_layout=0, eps=1e-5), pass_b)
t("skinny_down quickgelu", lambda: ops.skinny_down(x=x, w=w, bias=b, z=z, y=y, M=M, C=C, L=L, act=1, w_layout=0), pass_b)
t("skinny_down layout1", lambda: ops.skinny_down(x=x, w=wt, y=y, M=M, C=C, L=L, act=0, w_layout=1), pass_b)
t("skinny_down layout1 dropout", lambda: ops.skinny_down(x=x, w=wt, y=y, M=M, C=C, L=L, act=0, w_layout=1, drop_p=0.2, seed=1), pass_b)
t("skinny_up res", lambda: ops.skinny_up(lat=lat, w=wt, bias=g, res=res, out=out, M=M, C=C, L=L, w_layout=0), 2 * pass_b)
t("skinny_up res dropout", lambda: ops.skinny_up(lat=lat, w=wt, bias=g, res=res, out=out, M=M, C=C, L=L, w_layout=0, drop_p=0.2, seed=1), 2 * pass_b)
t("skinny_up accumulate layout1 +bf16", lambda: ops.skinny_up(lat=lat, w=w, out=out, out_bf16=out16, M=M, C=C, L=L, w_layout=1, accumulate=1), 2.5 * pass_b)
t("skinny_up LN-bwd epilogue", lambda: ops.skinny_up(lat=lat, w=w, res=res, out=out, ln_x=x, ln_mean=mean, ln_rstd=rstd, ln_gamma=g, M=M, C=C, L=L, w_layout=1), 3 * pass_b)
t("outer_reduce plain", lambda: ops.outer_reduce(narrow=lat, wide=x, scratch=scratch, out=dW, M=M, C=C, L=L, transposed=0, accumulate=0), pass_b)
t("outer_reduce LN transposed colsum", lambda: ops.outer_reduce(narrow=lat, wide=x, mean=mean, rstd=rstd, ln_gamma=g, ln_beta=bt, scratch=scratch, out=dWt, colsum=cs, M=M, C=C, L=L, transposed=1, accumulate=0), pass_b)
kw = dict(qkv=qkv, ctx=ctx, lse=lse, B=4, D=10, H=10, W=10, kd=6, kh=6, kw=6, L=L, scale=C ** -0.5)
t("window_attn_fwd", lambda: ops.window_attn_fwd(**kw), 4000 * 80 * 4)
t("window_attn_fwd dropout", lambda: ops.window_attn_fwd(drop_p=0.2, seed=3, **kw), 4000 * 80 * 4)
t("window_attn_bwd", lambda: ops.window_attn_bwd(dctx=dctx, delta=delta, dqkv=dqkv, **kw), 4000 * 160 * 4)
y16 = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
t("layernorm_fwd", lambda: ops.layernorm_fwd(x, g, bt, M, C, y16=y16, mean=mean, rstd=rstd), 1.5 * pass_b)
t("layernorm_bwd", lambda: ops.layernorm_bwd(x, res, mean, rstd, g, M, C, dx=out, dres=res, dx16=y16), 4.5 * pass_b)

# fused LayerNorm + projection kernels vs the plain LayerNorm kernels
y16 = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
dxo = f(M, C)
t("layernorm_fwd (plain)", lambda: ops.layernorm_fwd(x, g, bt, M, C, y16=y16, mean=mean, rstd=rstd), 1.5 * pass_b)
t("layernorm_fwd_proj layout0 gelu", lambda: ops.layernorm_fwd_proj(x, g, bt, M, C, y16=y16, mean=mean, rstd=rstd, w=w, bias=b, z=z, y=y, act=1, w_layout=0), 1.5 * pass_b)
t("layernorm_bwd (plain, dres, dx16)", lambda: ops.layernorm_bwd(out, x, mean, rstd, g, M, C, dx=dxo, dres=res, dx16=y16), 4.5 * pass_b)
t("layernorm_bwd_proj layout1", lambda: ops.layernorm_bwd_proj(out, x, mean, rstd, g, M, C, dx=dxo, dres=res, dx16=y16, w=wt, y=y, w_layout=1), 4.5 * pass_b)
