"""GPU box: time gvk_gemm_nt_bf16 for the hot-path shapes across tile choices (random bf16 data, HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaviko_amd import ops, lib

lib.require_device()
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4132
shapes = [("qkv", 2304, 768, ops.EPI_STORE_BF16), ("out", 768, 768, ops.EPI_BIAS_RES_F32), ("fc1", 3072, 768, ops.EPI_BIAS_GELU_BF16),
          ("fc2", 768, 3072, ops.EPI_BIAS_RES_F32), ("fc2_dgrad", 3072, 768, ops.EPI_GELU_BWD_BF16), ("fc1_dgrad", 768, 3072, ops.EPI_STORE_F32),
          ("out_dgrad", 768, 768, ops.EPI_STORE_BF16), ("qkv_dgrad", 768, 2304, ops.EPI_STORE_F32),
          ("fc1_plain16", 3072, 768, ops.EPI_STORE_BF16), ("fc1_plain32", 3072, 768, ops.EPI_STORE_F32), ("big", 4096, 4096, ops.EPI_STORE_BF16)]
if os.environ.get("SHAPES"):          # SHAPES="fc2:1024:4096:1,fc1d:1024:4096:0" -> (name, N, K, epilogue id)
    shapes = [(n, int(N), int(K), int(e)) for n, N, K, e in (x.split(":") for x in os.environ["SHAPES"].split(","))]
if os.environ.get("ONLY"):
    shapes = [s for s in shapes if s[0] in os.environ["ONLY"].split(",")]
tiles = [int(t) for t in os.environ["TILES"].split(",")] if os.environ.get("TILES") else [128128, 128064, 64128, 64064]
for name, N, K, epi in shapes:
    a = ops.act_zeros(M, K, torch.bfloat16, dev); a[:M] = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    f32 = epi in (ops.EPI_BIAS_RES_F32, ops.EPI_STORE_F32)
    out0 = ops.act_zeros(M, N, torch.float32 if f32 else torch.bfloat16, dev)
    out1 = ops.act_zeros(M, N, torch.bfloat16, dev)
    bias = torch.randn(N, device=dev)
    res = ops.act_zeros(M, N, torch.float32, dev)
    aux = ops.act_zeros(M, N, torch.bfloat16, dev); aux.normal_()
    line = f"{name:10s} N={N:5d} K={K:5d} "
    skws = torch.zeros((1024 + 256 * 2 * 65536) // 4, dtype=torch.int32, device=dev)      # tile 2128128: two pieces of 64 KiB per tile
    for t in tiles:
        kw = dict(epilogue=epi, tile=t)
        if t == 2128128: kw.update(splitk_ws=skws, ksplit=int(os.environ.get("KSPLIT", "2")))
        if epi == ops.EPI_BIAS_RES_F32: kw.update(bias=bias, res=res)
        if epi == ops.EPI_BIAS_GELU_BF16: kw.update(bias=bias, out1=out1)
        if epi == ops.EPI_GELU_BWD_BF16: kw.update(aux=aux)
        try:
            for _ in range(5): ops.gemm_nt(a, w, M, out0, **kw)
        except Exception as ex:
            line += f"| {t}: n/a "
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.gemm_nt(a, w, M, out0, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        line += f"| {t}: {us:6.1f}us {2.0 * M * N * K / us / 1e6:6.0f}TF "
    print(line, flush=True)
