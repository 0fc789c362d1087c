"""GPU box: shader-clock stamps of the attention forward at B=4, H=12, T=1033: one steady-state key tile (kt = 4) and the kernel's phases.
build: see probe_attn.hip.  usage: python tools/probe/probe_attn.py [kb] [var]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaviko_amd import lib, ops
lib.require_device(); lib.load()
dev = torch.device("cuda:0")
l = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("PROBE_LIB", "libprobe_attn.so")))
l.probe_attn_fwd.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 2
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 96
var = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B, T, H = 4, 1033, 12
inner = H * 64
qkv = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev); qkv[:B * T] = (torch.randn(B * T, 3 * inner, device=dev) * 0.7).bfloat16()
ops.qkv_prescale(qkv, B * T, H, 0.125)
out = ops.act_zeros(B * T, inner, torch.bfloat16, dev); lse = torch.empty(B * H * T, device=dev)
nwg = ((T + 127) // 128) * H * B
st = torch.cuda.current_stream().cuda_stream
for _ in range(5): l.probe_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, T, H, kb, var, None, st)
stamps = torch.zeros(nwg * 4 * 12, dtype=torch.int64, device=dev)
l.probe_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, T, H, kb, var, stamps.data_ptr(), st)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(nwg, 4, 12).astype(np.float64)
s = s[s[:, :, 9] > 0]                                   # waves that ran the epilogue (active ones)
med = lambda x: float(np.median(x))
names = ["stage issue (LDS-DMA of the next tile)", "S' = K.Q'^T - m (5 MFMA + 4 ds_read_b128 per 32 keys)", "running max + slow-path test", "exp2 / cvt / O^T += V^T.P^T", "barrier wait"]
tot = s[:, 5] - s[:, 0]
print(f"KB={kb} VAR={var}: one key tile ({kb} keys x 32 queries per wave): median {med(tot):.0f} cycles, p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}")
for i, n in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print(f"  {n:56s} median {med(d):6.0f}  p10 {np.percentile(d, 10):6.0f}  p90 {np.percentile(d, 90):6.0f}")
for n, a, b in (("prologue (entry -> first tile staged)", 6, 7), ("main loop", 7, 8), ("epilogue (LDS transpose + stores retired)", 8, 9), ("whole kernel", 6, 9)):
    d = s[:, b] - s[:, a]
    print(f"  {n:56s} median {med(d):6.0f}  p10 {np.percentile(d, 10):6.0f}  p90 {np.percentile(d, 90):6.0f}")
t0 = s[:, 6].min()
print(f"  first wave starts at 0, last wave starts at {s[:, 6].max() - t0:.0f}, last wave ends at {s[:, 9].max() - t0:.0f} cycles (100 MHz-independent shader clock)")
