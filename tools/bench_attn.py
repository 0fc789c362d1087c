"""GPU box: isolated timing of the bf16 attention kernels at the cfg2 shape (B=4, T=1033, H=12).  Every variant is a launch plan of 50
launches; the plans are replayed in interleaved rounds inside ONE process (min / median over rounds).
usage: python tools/bench_attn.py [B T H] [--fwd-variants]"""
import os, statistics, sys
os.environ.setdefault("GAVIKO_HIP_DIAG", "1")      # kernel variants / A/B switches live in the measurement build (python -m gaviko_amd.build --diag), statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaviko_amd import lib, ops
lib.require_device()
dev = torch.device("cuda:0")
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
B, T, H = (int(a) for a in (pos + ["4", "1033", "12"][len(pos):]))
inner = H * 64
qkv = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev); qkv[:B * T] = (torch.randn(B * T, 3 * inner, device=dev) * 0.7).bfloat16()
ops.qkv_prescale(qkv, B * T, H, 0.125)                  # the operand form the engine hands the kernels
out, dout = ops.act_zeros(B * T, inner, torch.bfloat16, dev), ops.act_zeros(B * T, inner, torch.bfloat16, dev)
dout[:B * T] = torch.randn(B * T, inner, device=dev).bfloat16()
lse, delta = torch.empty(B * H * T, device=dev), torch.empty(B * H * T, device=dev)
dqkv = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev)
l = lib.load()
NL = 50


def plan(fn, env=None):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        for _ in range(3): fn()
        lib.check(l.gvk_plan_begin(), "begin")
        for _ in range(NL): fn()
        pid = l.gvk_plan_end()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    torch.cuda.synchronize()
    return pid


def race(cases, flops, rounds=7):
    """cases: [(name, plan id)] -> interleaved rounds"""
    t = {n: [] for n, _ in cases}
    for _ in range(rounds):
        for n, pid in cases:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            l.gvk_plan_replay(pid)
            e0.record(); l.gvk_plan_replay(pid); e1.record(); torch.cuda.synchronize()
            t[n].append(e0.elapsed_time(e1) * 1e3 / NL)
    for n, _ in cases:
        mn, md = min(t[n]), statistics.median(t[n])
        print(f"{n:36s} min {mn:7.2f} us  median {md:7.2f} us  {flops[n] / md / 1e6:6.0f} TFLOP/s", flush=True)


f = 4.0 * B * H * T * T * 64
fwd = lambda: ops.attention_fwd(qkv, out, lse, B, T, H, 0.125, q_prescaled=True)
bwd = lambda: ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, T, H, 0.125, q_prescaled=True)
cases, flops = [], {}
if "--fwd-variants" in sys.argv:
    for kb in (96, 128):
        for var in (0, 1, 2, 3):
            n = f"fwd KB={kb} VAR={var}"
            cases.append((n, plan(fwd, {"GAVIKO_HIP_ATTN_KB": str(kb), "GAVIKO_HIP_ATTN_VAR": str(var)})))
            flops[n] = f
cases.append(("attention_fwd (default)", plan(fwd))); flops[cases[-1][0]] = f
cases.append(("attention_bwd two-pass (7 products)", plan(bwd))); flops[cases[-1][0]] = 2.5 * f
wsp = ops.attention_bwd_workspace(B, T, H, dev)
bwd1 = lambda: ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, T, H, 0.125, q_prescaled=True, ws=wsp)
cases.append(("attention_bwd one-pass (5 products)", plan(bwd1))); flops[cases[-1][0]] = 2.5 * f
if "--fused-variants" in sys.argv:        # timing ablations of the one-pass kernel (measurement build; results of VAR != 0 are wrong)
    for var, what in ((1, "no waits / sum loads"), (3, "no hand-off at all"), (7, "no dQ product, no hand-off"), (8, "no per-sub-block vmcnt drain"),
                      (11, "no hand-off, no drain"), (15, "no dQ, no hand-off, no drain"), (16, "no polling, sums loaded + stored"), (33, "dQ product only: no loads, stores, polls"), (129, "stores only, barrier without vmcnt drain"), (144, "loads + stores, no polls, barrier without vmcnt drain")):
        n = f"one-pass VAR={var} ({what})"
        cases.append((n, plan(bwd1, {"GAVIKO_HIP_ATTN_VAR": str(var)}))); flops[n] = 2.5 * f
    wsp.zero_()
race(cases, flops)
print('hand-off timeouts:', ops.attention_bwd_timeouts(wsp))
so = int(l.gvk_attention_bwd_status_offset(wsp.numel() * 4)) // 4
wsp[so:so + 4] = 0
for _ in range(10): bwd1()
torch.cuda.synchronize()
print('per launch: late waits', wsp[so + 1].item() / 10, 'of', B * H * ((T + 127) // 128 - 1) * ((T + 31) // 32), ' polls', wsp[so + 2].item() / 10)

if "--stamps" in sys.argv:          # phase totals of the one-pass kernel's loop (VAR = 64, measurement build): sub-blocks 4..27 of wave w of three workgroups
    os.environ["GAVIKO_HIP_ATTN_VAR"] = "64"
    for _ in range(5): bwd1()
    torch.cuda.synchronize()
    os.environ.pop("GAVIKO_HIP_ATTN_VAR")
    raw = wsp[so + 16: so + 16 + 3 * 64].cpu().view(torch.int64).view(3, 4, 8)[:, :, :6]
    names = ["finish_prev", "dK dV + dS write", "drain vmcnt(0)", "poll + barrier", "signal, request, tile switch", "dQ + next scores"]
    for wg in range(3):
        for w in range(4):
            tot = raw[wg, w].sum().item()
            print(f"wg {wg} wave {w}: " + "  ".join(f"{n} {raw[wg, w, k].item() / 24:.0f}" for k, n in enumerate(names)) + f"  | per sub-block {tot / 24:.0f} clocks (100 MHz ticks x ?)")
