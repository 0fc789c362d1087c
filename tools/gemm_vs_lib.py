"""GPU box: the eight hot-path GEMM shapes of a ViT-B B = 4 layer (M = 4132), each timed three ways, back to back on an otherwise idle GPU:
  ours      -- gvk_gemm_nt_bf16 with the tile the product picks (tile=0) and the epilogue the step uses (bias / residual / GELU / GELU')
  ours_plain-- the same launch with the plain bf16 store (what the vendor call below computes)
  lib       -- torch.mm (hipBLASLt / rocBLAS as torch picks it), bf16 in, bf16 out, no epilogue
Writes a CSV (default profiles-style name under gpurun_out/); the in-step durations of the same launches are in r04_kernel_stats_by_shape.csv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaviko_amd import ops, lib

lib.require_device()
dev = torch.device("cuda:0")
M = 4132
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_isolated_vs_lib.csv"
shapes = [("qkv fwd", 2304, 768, ops.EPI_STORE_BF16), ("out-proj fwd", 768, 768, ops.EPI_BIAS_RES_F32), ("fc1 fwd", 3072, 768, ops.EPI_BIAS_GELU_BF16),
          ("fc2 fwd", 768, 3072, ops.EPI_BIAS_RES_F32), ("fc2 dgrad", 3072, 768, ops.EPI_GELU_BWD_BF16), ("fc1 dgrad", 768, 3072, ops.EPI_STORE_F32),
          ("out-proj dgrad", 768, 768, ops.EPI_STORE_BF16), ("qkv dgrad", 768, 2304, ops.EPI_STORE_F32)]
REP = 100


def timed(fn):
    for _ in range(10): fn()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP): fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / REP)
    return min(best)


rows = ["call site,M,N,K,ours_us,ours_tflops,ours_plain_store_us,lib_us,lib_tflops,ours_over_lib"]
for name, N, K, epi in shapes:
    a = ops.act_zeros(M, K, torch.bfloat16, dev); a[:M] = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    f32 = epi in (ops.EPI_BIAS_RES_F32, ops.EPI_STORE_F32)
    out0 = ops.act_zeros(M, N, torch.float32 if f32 else torch.bfloat16, dev)
    out1 = ops.act_zeros(M, N, torch.bfloat16, dev)
    outp = ops.act_zeros(M, N, torch.bfloat16, dev)
    bias = torch.randn(N, device=dev)
    res = ops.act_zeros(M, N, torch.float32, dev)
    aux = ops.act_zeros(M, N, torch.bfloat16, dev); aux.normal_()
    kw = dict(epilogue=epi)
    if epi == ops.EPI_BIAS_RES_F32: kw.update(bias=bias, res=res)
    if epi == ops.EPI_BIAS_GELU_BF16: kw.update(bias=bias, out1=out1)
    if epi == ops.EPI_GELU_BWD_BF16: kw.update(aux=aux)
    ours = timed(lambda: ops.gemm_nt(a, w, M, out0, **kw))
    plain = timed(lambda: ops.gemm_nt(a, w, M, outp, epilogue=ops.EPI_STORE_BF16))
    al = a[:M].contiguous(); wt = w.t(); ol = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    libus = timed(lambda: torch.mm(al, wt, out=ol))
    # same numbers?  (bf16 store of fp32 accumulation on both sides)
    ops.gemm_nt(a, w, M, outp, epilogue=ops.EPI_STORE_BF16); torch.mm(al, wt, out=ol)
    err = (outp[:M].float() - ol.float()).abs().max().item() / ol.float().abs().max().item()
    assert err < 2e-2, (name, err)
    fl = 2.0 * M * N * K
    rows.append(f"{name},{M},{N},{K},{ours:.1f},{fl / ours / 1e6:.0f},{plain:.1f},{libus:.1f},{fl / libus / 1e6:.0f},{ours / libus:.2f}")
    print(rows[-1], flush=True)
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
with open(out_path, "w") as f:
    f.write("\n".join(rows) + "\n")
