"""GPU box: where the eight-phase GEMM's time goes.  Wall time of the probe builds (ablations) + in-kernel stamps of the full build.
   python3 tools/probe/probe_gemm8p.py          (libs built by the hipcc line in probe_gemm8p.hip)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaviko_amd import lib, ops
lib.require_device(); lib.load()
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
libs = {ab: ctypes.CDLL(os.path.join(here, f"libprobe_gemm8p_{ab}.so")) for ab in (0, 1, 2, 4, 5) if os.path.exists(os.path.join(here, f"libprobe_gemm8p_{ab}.so"))}
for l in libs.values():
    l.probe_gemm8p.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
NAMES = {0: "full", 1: "no LDS-DMA in loop", 2: "no MFMA", 4: "no ds_read", 5: "no DMA, no ds_read (MFMA + barriers)"}
shapes = [("fc1", 4132, 3072, 768), ("big", 4096, 4096, 4096), ("fc2", 4132, 768, 3072), ("qkv", 4132, 2304, 768)]
st = torch.cuda.current_stream().cuda_stream
for name, M, N, K in shapes:
    a = ops.act_zeros(M, K, torch.bfloat16, dev); a[:M] = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    out = ops.act_zeros(M, N, torch.bfloat16, dev)
    nwg = ((M + 255) // 256) * (N // 256)
    for var in (1, 0):
        line = f"{name:4s} M={M} N={N} K={K} var{var}: "
        for ab, l in libs.items():
            for _ in range(3): l.probe_gemm8p(a.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, var, None, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): l.probe_gemm8p(a.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, var, None, st)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            line += f"| {NAMES[ab]}: {us:6.1f} us "
        print(line, flush=True)
        # stamps of the full build (a few warm launches first)
        stamps = torch.zeros(nwg * 8 * 32, dtype=torch.int64, device=dev)
        for _ in range(3): libs[0].probe_gemm8p(a.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, var, None, st)
        libs[0].probe_gemm8p(a.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, var, stamps.data_ptr(), st)
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(nwg, 8, 32).astype(np.float64)
        nt = K // 64
        cyc = s[:, :, 22] - s[:, :, 0]
        rt = (s[:, :, 24] - s[:, :, 23]) / 100.0                      # us (100 MHz)
        clk = np.median(cyc / rt) / 1e3
        med = lambda x: float(np.median(x))
        print(f"     in-kernel: life {med(rt):6.1f} us, clock {clk:.2f} GHz | prologue {med(s[:,:,1]-s[:,:,0]):7.0f} cyc | loop {med((s[:,:,2]-s[:,:,1])/max(1,nt-3)):6.0f} cyc/k-tile "
              f"| stamped k-tile {med(s[:,:,20]-s[:,:,2]):6.0f} | tail 2 k-tiles {med(s[:,:,21]-s[:,:,20]):6.0f} | epilogue {med(s[:,:,22]-s[:,:,21]):6.0f} cyc", flush=True)
        for g in (0, 1):
            ws = s[:, 4 * g: 4 * g + 4, :]
            parts = []
            for ph in range(4):
                b = 4 + 4 * ph
                prev_end = ws[:, :, b - 1] if ph > 0 else ws[:, :, 2]
                parts.append(f"p{ph}: load {med(ws[:,:,b]-prev_end):4.0f} lgkm {med(ws[:,:,b+1]-ws[:,:,b]):4.0f} mfma {med(ws[:,:,b+2]-ws[:,:,b+1]):4.0f} bar {med(ws[:,:,b+3]-ws[:,:,b+2]):4.0f}")
            print(f"     group {g}: " + " | ".join(parts), flush=True)
