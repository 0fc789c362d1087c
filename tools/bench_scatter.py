import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gaviko_amd import ops, lib
lib.require_device()
dev = torch.device("cuda:0")
M, C, L = 4132, 768, 20
f = lambda *s: torch.randn(*s, device=dev)
lat, w = f(M, L), f(L, C) * 0.03
out = ops.act_zeros(M, C, torch.float32, dev); out[:M] = f(M, C)
out16 = ops.act_zeros(M, C, torch.bfloat16, dev)
A = ops.act_zeros(M, 64, torch.bfloat16, dev)
W = torch.zeros(C, 64, dtype=torch.bfloat16, device=dev)
wt = w.t().contiguous()
ops.pack_split_bf16(lat, A, 0, M, weight_side=False)
ops.pack_split_bf16(wt, W, 0, C, weight_side=True)
# correctness of the GEMM form
ref = out[:M].double() + lat.double() @ w.double()
o2 = out.clone(); o16 = torch.zeros_like(out16)
ops.gemm_nt(A, W, M, o2, epilogue=ops.EPI_BIAS_RES_F32_BF16, out1=o16, res=o2)
torch.cuda.synchronize()
print("gemm-form max err", (o2[:M].double() - ref).abs().max().item(), "ref scale", ref.abs().max().item())
o3 = out.clone()
ops.skinny_up(lat=lat, w=w, out=o3, out_bf16=out16, M=M, C=C, L=L, w_layout=1, accumulate=1)
torch.cuda.synchronize()
print("side_up max err", (o3[:M].double() - ref).abs().max().item())
def t(name, fn):
    for _ in range(3): fn()
    l = lib.load()
    lib.check(l.gvk_plan_begin(), "begin")
    for _ in range(50): fn()
    pid = l.gvk_plan_end()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    l.gvk_plan_replay(pid); e0.record(); l.gvk_plan_replay(pid); e1.record(); torch.cuda.synchronize()
    l.gvk_plan_free(pid)
    print(f"{name:44s} {e0.elapsed_time(e1) * 1e3 / 50:7.1f} us", flush=True)
t("side_up scatter (+bf16)", lambda: ops.skinny_up(lat=lat, w=w, out=out, out_bf16=out16, M=M, C=C, L=L, w_layout=1, accumulate=1))
for tile in (0, 64128, 128128, 3128128, 3064128):
    try:
        t(f"gemm K=64 epi6 in place, tile {tile}", lambda: ops.gemm_nt(A, W, M, out, epilogue=ops.EPI_BIAS_RES_F32_BF16, out1=out16, res=out, tile=tile))
    except Exception as ex:
        print(tile, "n/a", str(ex)[:80])
t("pack_split activation side (M x 20)", lambda: ops.pack_split_bf16(lat, A, 0, M, weight_side=False))
