"""GPU box: gvk_param_grads alone at the cfg2 shapes (GPA form: proj_up + proj_down fed by both token streams + six small jobs; MWSA form:
proj_up, LayerNorm + proj_down, qkv), replayed from a launch plan.  ROWS=<frac> scales the row counts (streaming-bound or fixed cost?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaviko_amd import ops, lib
lib.require_device()
dev = torch.device("cuda:0")
frac = float(os.environ.get("ROWS", "1"))
B, T, N, P, C, L = 4, 1033, 1000, 32, 768, 20
M, BN = int(B * T * frac), int(B * N * frac)
f = lambda *s: torch.randn(*s, device=dev)
xl, enh, dG, G1, Lc, dzx, dzl = f(M, L), f(B, P, L), f(M, C), f(M, C), f(BN, C), f(M, L), f(BN, L)
ng = ops.gpa_gate_param_count(L, P)
gate_part, gate_flat = f(B, ng), torch.zeros(ng, device=dev)
dqg, prm = f(B * P, L), f(B * P, L)
wq, bq, wl, bl, gbd = (torch.zeros(L, L, device=dev), torch.zeros(L, device=dev), torch.zeros(L, L, device=dev), torch.zeros(L, device=dev), torch.zeros(L, device=dev))
gup, gupb, gwd = torch.zeros(C, L, device=dev), torch.zeros(C, device=dev), torch.zeros(L, C, device=dev)
nct = (C + 63) // 64
scr = torch.zeros(ops.param_grads_scratch_elems(L, [nct, nct, 1], [ng, L * L, L, L * L, L, L]), device=dev)
tick = torch.zeros(ops.PGRAD_TICKETS, dtype=torch.int32, device=dev)
ctx, dL, lin, dlat, mean, rstd = f(BN, L), f(BN, C), f(BN, C), f(BN, L), f(BN), f(BN).abs() + 0.5
lat, dqkv = f(BN, L), f(BN, 3 * L)
g_, b_, wd = f(C), f(C), f(L, C)
gu2, gub2, gwd2, gg, gb, gdb, gq = (torch.zeros(C, L, device=dev), torch.zeros(C, device=dev), torch.zeros(L, C, device=dev), torch.zeros(C, device=dev),
                                    torch.zeros(C, device=dev), torch.zeros(L, device=dev), torch.zeros(3 * L, L, device=dev))

def gpa(small=True, outer=(0, 1)):
    o = [dict(narrow=xl, wide=dG, lat_override=enh, out=gup, colsum=gupb, M=M, T=T, P=P, transposed=1, accumulate=0),
         dict(narrow=dzx, wide=G1, narrow2=dzl, wide2=Lc, out=gwd, M=M, M2=BN, transposed=0, accumulate=0)]
    s = [(gate_part, None, gate_flat, 0), (dqg, prm, wq, 0), (dqg, None, bq, 0), (dqg, prm, wl, 0), (dqg, None, bl, 0), (dzx, None, gbd, 0, dzl)]
    ops.param_grads([o[k] for k in outer], s if small else [], scr, tick, C, L)

def mwsa():
    ops.param_grads([dict(narrow=ctx, wide=dL, out=gu2, colsum=gub2, M=BN, transposed=1, accumulate=0),
                     dict(narrow=dlat, wide=lin, mean=mean, rstd=rstd, out=gwd2, aff_w=wd, aff_gamma=g_, aff_beta=b_, aff_dgamma=gg, aff_dbeta=gb, aff_dbias=gdb,
                          M=BN, accumulate=0),
                     dict(narrow=lat, wide=dqkv, out=gq, M=BN, C=3 * L, transposed=1, accumulate=0)], [], scr, tick, C, L)

def t(name, fn, bytes_):
    for _ in range(3): fn()
    l = lib.load()
    lib.check(l.gvk_plan_begin(), "begin")
    for _ in range(50): fn()
    pid = l.gvk_plan_end()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    l.gvk_plan_replay(pid)
    e0.record(); l.gvk_plan_replay(pid); e1.record(); torch.cuda.synchronize()
    l.gvk_plan_free(pid)
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{name:40s} {us:7.1f} us   {bytes_ / us / 1e6:6.2f} TB/s of wide-operand bytes", flush=True)

wb = 4 * C
t("GPA  (2 outer + 6 small)", gpa, (2 * M + BN) * wb)
t("GPA  (2 outer, no small)", lambda: gpa(False), (2 * M + BN) * wb)
t("GPA  (proj_up only)", lambda: gpa(False, (0,)), M * wb)
t("GPA  (proj_down only, two streams)", lambda: gpa(False, (1,)), (M + BN) * wb)
t("MWSA (3 outer)", mwsa, 2 * BN * wb)
