"""GPU box: shader-clock stamps of one steady-state key tile (kt = 4) of the four-wave attention forward at B=4, H=12, T=1033."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaviko_amd import lib, ops
lib.require_device(); lib.load()
dev = torch.device("cuda:0")
l = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("PROBE_LIB", "libprobe_attn_false.so")))
l.probe_attn_fwd.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
B, T, H = 4, 1033, 12
inner = H * 64
qkv = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev); qkv[:B * T] = (torch.randn(B * T, 3 * inner, device=dev) * 0.7).bfloat16()
out = ops.act_zeros(B * T, inner, torch.bfloat16, dev); lse = torch.empty(B * H * T, device=dev)
nwg = ((T + 127) // 128) * H * B
st = torch.cuda.current_stream().cuda_stream
for _ in range(5): l.probe_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, T, H, None, st)
stamps = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
l.probe_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, T, H, stamps.data_ptr(), st)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(nwg, 4, 8).astype(np.float64)
med = lambda x: float(np.median(x))
names = ["stage issue (8 LDS-DMA)", "S^T = K.Q^T (16 MFMA + 16 ds_read_b128)", "softmax (max, 64 exp2, sums)", "O^T += V^T.P^T (16 MFMA + 32 tr reads + 32 cvt)", "barrier wait"]
tot = s[:, :, 5] - s[:, :, 0]
print(f"one key tile (128 keys x 32 queries per wave): median {med(tot):.0f} cycles, p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}")
for i, n in enumerate(names):
    d = s[:, :, i + 1] - s[:, :, i]
    print(f"  {n:52s} median {med(d):6.0f}  p10 {np.percentile(d, 10):6.0f}  p90 {np.percentile(d, 90):6.0f}")
