"""GPU box: what does the vendor library (torch.mm -> hipBLASLt/rocBLAS) reach on the hot-path GEMM shapes?  Reference point only."""
import torch
dev = torch.device("cuda:0")
for name, M, N, K in [("qkv", 4132, 2304, 768), ("out", 4132, 768, 768), ("fc1", 4132, 3072, 768), ("fc2", 4132, 768, 3072), ("big", 4096, 4096, 4096),
                      ("fc1_B8", 8264, 3072, 768), ("fc2_B8", 8264, 768, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5): torch.mm(a, w.t(), out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): torch.mm(a, w.t(), out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{name:8s} M={M} N={N} K={K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF")
