"""-m gpu: attention backward, the rank-L 'skinny' kernels, MWSA window attention, the GPA core and the head,
each through the C-ABI against a float64 torch autograd computation (the oracle's functions where one exists)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def _bf(x):
    return x.float().to(torch.bfloat16).double()


def _close(got, want, rtol, name=""):
    got = got.detach().cpu().double()
    err = (got - want).abs().max().item()
    ref = max(1e-6, want.abs().max().item())
    assert err <= rtol * ref, f"{name}: max err {err:.3e} vs scale {ref:.3e} (tol {rtol})"


@pytest.mark.parametrize("kb", [0, 96, 128, "fused"])
@pytest.mark.parametrize("B,T,H", [(2, 1033, 3), (1, 1001, 2), (2, 65, 1), (1, 393, 3), (1, 31, 2), (3, 129, 2), (1, 1, 1), (2, 257, 12)])
def test_attention_bwd(dev, monkeypatch, B, T, H, kb):
    """kb: tile size of both passes of the two-pass kernels (0 = the launcher's choice: whichever pads the sequence less); "fused" = the
    one-pass kernel (five products, dQ summed over the key blocks by the ordered hand-off), which must also be bitwise repeatable and
    leave its progress words zero."""
    from gaviko_amd import ops
    fused = kb == "fused"
    if kb and not fused:
        monkeypatch.setenv("GAVIKO_HIP_ATTN_KB", str(kb))
    inner = H * 64
    C_ = 0.125 * 1.4426950408889634
    # device operand: q block pre-scaled by scale*log2(e), ONE rounding (what the qkv GEMM's epilogue writes); the reference differentiates
    # attention over exactly the q, k, v those bf16 values represent, with respect to the UNSCALED q
    raw = _rand((B, T, 3 * inner), 61, 2.0)
    op = raw.clone(); op[..., :inner] *= C_
    op = op.bfloat16()
    exact = op.double(); exact[..., :inner] /= C_
    qkv = exact.clone().requires_grad_(True)
    dO = _bf(_rand((B, T, inner), 62, 1.0))
    q, k, v = (t.reshape(B, T, H, 64).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    s = q @ k.transpose(-1, -2) * 0.125
    o = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, T, inner)
    o.backward(dO)
    Q = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev)
    Q[: B * T] = op.reshape(B * T, -1).to(dev)
    O = ops.act_zeros(B * T, inner, torch.bfloat16, dev)
    lse = torch.zeros((B, H, T), device=dev)
    ops.attention_fwd(Q, O, lse, B, T, H, 0.125, q_prescaled=True)
    DO = ops.act_zeros(B * T, inner, torch.bfloat16, dev)
    DO[: B * T] = dO.reshape(B * T, -1).to(dev).bfloat16()
    DQ = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev)
    delta = torch.zeros((B, H, T), device=dev)
    wsp = ops.attention_bwd_workspace(B, T, H, dev) if fused else None
    ops.attention_bwd(Q, O, DO, lse, delta, DQ, B, T, H, 0.125, q_prescaled=True, ws=wsp)
    torch.cuda.synchronize()
    if fused:
        st = int(ops.L.load().gvk_attention_bwd_status_offset(wsp.numel() * 4)) // 4
        assert ops.attention_bwd_timeouts(wsp) == 0 and int(wsp[:st].abs().max()) == 0          # no bounded wait hit, every progress word back at zero
        _close(delta, (o.detach() * dO).reshape(B, T, H, 64).sum(-1).permute(0, 2, 1), 2e-2, "delta")
        first = DQ.clone()
        for _ in range(3):                                  # the hand-off order is a function of (tile, key block) only: bit-equal every time
            DQ.zero_()
            ops.attention_bwd(Q, O, DO, lse, delta, DQ, B, T, H, 0.125, q_prescaled=True, ws=wsp)
            torch.cuda.synchronize()
            assert torch.equal(DQ, first)
        assert ops.attention_bwd_timeouts(wsp) == 0 and int(wsp[:st].abs().max()) == 0
    got = DQ[: B * T].view(B, T, 3 * inner).cpu().double()
    want = qkv.grad
    for name, sl in (("dq", slice(0, inner)), ("dk", slice(inner, 2 * inner)), ("dv", slice(2 * inner, 3 * inner))):
        e = (got[..., sl] - want[..., sl]).abs().max().item() / max(want[..., sl].abs().max().item(), 1e-30)     # (T = 1: dq = dk = 0 exactly)
        print(f"attention_bwd B={B} T={T} H={H} kb={kb} {name}: rel err {e:.2e}")
        _close(got[..., sl], want[..., sl], 2.5e-2, name)


@pytest.mark.parametrize("M,C_,Lat", [(1033, 768, 20), (300, 192, 20), (77, 1024, 8)])
def test_skinny_down_up_outer(dev, M, C_, Lat):
    from gaviko_amd import ops
    x = _rand((M, C_), 71, 1.5) + 0.2
    w = _rand((Lat, C_), 72, 1 / math.sqrt(C_))
    bias = _rand((Lat,), 73, 0.1)
    g = 1 + _rand((C_,), 74, 0.2)
    bt = _rand((C_,), 75, 0.1)
    w2 = _rand((3 * Lat, Lat), 76, 0.5)
    f = lambda t: t.float().to(dev).contiguous()
    # LN + down + second stage
    z = torch.zeros((M, Lat), device=dev); y = torch.zeros_like(z); y2 = torch.zeros((M, 3 * Lat), device=dev)
    mean = torch.zeros(M, device=dev); rstd = torch.zeros(M, device=dev)
    ops.skinny_down(x=f(x), w=f(w), bias=f(bias), ln_gamma=f(g), ln_beta=f(bt), mean=mean, rstd=rstd, z=z, y=y, w2=f(w2), y2=y2,
                    M=M, C=C_, L=Lat, L2=3 * Lat, act=0, w_layout=0, eps=1e-5)
    n = F.layer_norm(x, (C_,), g, bt, 1e-5)
    want = n @ w.T + bias
    _close(y, want, 2e-5, "down")
    _close(y2, want @ w2.T, 3e-5, "down second stage")
    # QuickGELU + transposed weight layout, no LN
    ops.skinny_down(x=f(x), w=f(w.T.contiguous()), bias=f(bias), z=z, y=y, M=M, C=C_, L=Lat, act=1, w_layout=1)
    pre = x @ w.T + bias
    _close(z, pre, 2e-5, "down z")
    _close(y, pre * torch.sigmoid(1.702 * pre), 2e-5, "down quickgelu")
    # up: res + lat.W^T + b, with the prompt-row override
    wu = _rand((C_, Lat), 77, 0.3)
    bu = _rand((C_,), 78, 0.1)
    lat = _rand((M, Lat), 79)
    res = _rand((M, C_), 80)
    out = torch.zeros((M, C_), device=dev)
    ops.skinny_up(lat=f(lat), w=f(wu), bias=f(bu), res=f(res), out=out, M=M, C=C_, L=Lat, w_layout=0)
    _close(out, res + lat @ wu.T + bu, 2e-5, "up")
    T, P = M // 2, 5
    ov = _rand((2, P, Lat), 81)
    lat2 = lat.clone()
    for s in range(2):
        lat2[s * T: s * T + P] = ov[s]
    acc0 = _rand((M, C_), 82)
    out2 = f(acc0)
    ops.skinny_up(lat=f(lat), w=f(wu.T.contiguous()), bias=f(bu), out=out2, lat_override=f(ov), M=2 * T, C=C_, L=Lat, T=T, P=P,
                  w_layout=1, accumulate=1)
    _close(out2[: 2 * T], (acc0 + lat2 @ wu.T + bu)[: 2 * T], 2e-5, "up accumulate+override")
    # outer: dW[l][c] = sum_m narrow * LN(wide); transposed + colsum
    scratch = torch.zeros(ops.outer_scratch_elems(Lat, C_), device=dev)
    dW = torch.zeros((Lat, C_), device=dev)
    ops.outer_reduce(narrow=f(lat), wide=f(x), mean=mean, rstd=rstd, ln_gamma=f(g), ln_beta=f(bt), scratch=scratch, out=dW,
                     M=M, C=C_, L=Lat, transposed=0, accumulate=0)
    _close(dW, lat.T @ n, 5e-5, "outer LN")
    dWt = torch.ones((C_, Lat), device=dev)
    cs = torch.ones(C_, device=dev)
    ops.outer_reduce(narrow=f(lat), wide=f(res), scratch=scratch, out=dWt, colsum=cs, M=M, C=C_, L=Lat, transposed=1, accumulate=1)
    _close(dWt, 1 + res.T @ lat, 5e-5, "outer transposed accumulate")
    _close(cs, 1 + res.sum(0), 5e-5, "outer colsum")
    # small wgrad / colsum
    a = _rand((M, 3 * Lat), 83); b = _rand((M, Lat), 84)
    o = torch.zeros((3 * Lat, Lat), device=dev)
    sc = torch.zeros(64 * 3 * Lat * Lat, device=dev)
    ops.small_wgrad(f(a), f(b), o, sc, M, 3 * Lat, Lat)
    _close(o, a.T @ b, 5e-5, "small_wgrad")
    o1 = torch.zeros(Lat, device=dev)
    ops.colsum(f(b), o1, sc, M, Lat)
    _close(o1, b.sum(0), 5e-5, "colsum")


def test_skinny_dropout_consistency(dev):
    """The counter-based mask is identical in up (fwd) and down/outer (bwd): d/dlat of sum(drop(lat.W^T)*dy) matches."""
    from gaviko_amd import ops
    M, C_, Lat, p = 200, 192, 20, 0.2
    f = lambda t: t.float().to(dev).contiguous()
    lat = _rand((M, Lat), 91); wu = _rand((C_, Lat), 92, 0.3); dy = _rand((M, C_), 93)
    out = torch.zeros((M, C_), device=dev)
    ops.skinny_up(lat=f(lat), w=f(wu), out=out, M=M, C=C_, L=Lat, w_layout=0, drop_p=p, seed=1234)
    raw = (lat @ wu.T)
    mask = (out.cpu().double().abs() > 0).double() / (1 - p)
    keep = (out.cpu().double().abs() > 0).double().mean().item()
    assert abs(keep - (1 - p)) < 0.02
    _close(out, raw * mask, 2e-5, "dropout fwd")
    dlat = torch.zeros((M, Lat), device=dev)
    ops.skinny_down(x=f(dy), w=f(wu), y=dlat, M=M, C=C_, L=Lat, act=0, w_layout=1, drop_p=p, seed=1234)
    _close(dlat, (dy * mask) @ wu, 3e-5, "dropout bwd (down)")
    scratch = torch.zeros(ops.outer_scratch_elems(Lat, C_), device=dev)
    dW = torch.zeros((C_, Lat), device=dev)
    ops.outer_reduce(narrow=f(lat), wide=f(dy), scratch=scratch, out=dW, M=M, C=C_, L=Lat, transposed=1, drop_p=p, seed=1234)
    _close(dW, (dy * mask).T @ lat, 5e-5, "dropout bwd (outer)")


@pytest.mark.parametrize("local_k", [(6, 6, 6), (3, 6, 6), (3, 3, 3)])
def test_window_attention_fwd_bwd(dev, local_k):
    from gaviko_amd import ops
    from oracle.gaviko_ref import window_mask
    B, Lat, C_ = 2, 20, 768
    N = 1000
    qkv = _rand((B, N, 3 * Lat), 101, 6.0).requires_grad_(True)
    mask = window_mask((10, 10, 10), local_k, dtype=torch.float64)
    q, k, v = qkv.chunk(3, -1)
    attn = (q @ k.transpose(-2, -1) * C_ ** -0.5 + mask).softmax(-1)
    ctx = attn @ v
    dctx = _rand((B, N, Lat), 102)
    ctx.backward(dctx)
    f = lambda t: t.detach().float().to(dev).contiguous()
    ctx_d = torch.zeros((B * N, Lat), device=dev); lse = torch.zeros(B * N, device=dev)
    kw = dict(qkv=f(qkv.reshape(B * N, -1)), ctx=ctx_d, lse=lse, B=B, D=10, H=10, W=10, kd=local_k[0], kh=local_k[1], kw=local_k[2], L=Lat,
              scale=C_ ** -0.5)
    ops.window_attn_fwd(**kw)
    _close(ctx_d.view(B, N, Lat), ctx.detach(), 3e-5, "window ctx")
    dq = torch.zeros((B * N, 3 * Lat), device=dev); delta = torch.zeros(B * N, device=dev)
    ops.window_attn_bwd(dctx=f(dctx.reshape(B * N, -1)), delta=delta, dqkv=dq, **kw)
    _close(dq.view(B, N, 3 * Lat), qkv.grad, 5e-5, "window dqkv")


def test_window_attention_matches_reference_masks():
    """(CPU part of the contract is in test_oracle; here: the golden allow-maps have the per-row counts the kernel's
    index arithmetic implies.)"""
    from conftest import golden
    for lk in ((6, 6, 6), (3, 6, 6), (3, 3, 3)):
        gz = golden(f"mwsa_mask_{lk[0]}{lk[1]}{lk[2]}")
        cnt = gz["count"]
        n = 0
        for i in range(1000):
            d, h, w = i // 100, (i // 10) % 10, i % 10
            c = 1
            for q, kk in ((d, lk[0]), (h, lk[1]), (w, lk[2])):
                lo, hi = max(0, q - kk // 2), min(10, q - kk // 2 + kk)
                c *= hi - lo
            n += int(c == cnt[i])
        assert n == 1000


def _gpa_params(Lat, P, seed):
    names = {"ca0_g": (Lat,), "ca0_b": (Lat,), "ca1_w": (64, Lat), "ca1_b": (64,), "ca3_w": (P, 64), "ca3_b": (P,),
             "gl0_g": (Lat,), "gl0_b": (Lat,), "gl1_w": (1, Lat), "gl1_b": (1,),
             "wgq": (Lat, Lat), "bgq": (Lat,), "wlq": (Lat, Lat), "blq": (Lat,)}
    out = {}
    for i, (k, shp) in enumerate(names.items()):
        t = _rand(shp, seed + i, 0.6)
        if k in ("ca0_g", "gl0_g"):
            t = 1 + 0.3 * t
        out[k] = t.requires_grad_(True)
    return out


@pytest.mark.parametrize("B,P,N", [(2, 32, 1000), (3, 8, 1000)])
def test_gpa_core_fwd_bwd(dev, B, P, N):
    """gvk_gpa_fwd/bwd vs the oracle's awakening_prompt internals (latent space part) under float64 autograd."""
    from gaviko_amd import ops
    Lat = 20
    T = P + 1 + N
    prm = _gpa_params(Lat, P, 200)
    zx = _rand((B, T, Lat), 111, 2.0).requires_grad_(True)
    zl = _rand((B, N, Lat), 112, 2.0).requires_grad_(True)
    qg_ = lambda t: t * torch.sigmoid(1.702 * t)
    xl, ll = qg_(zx), qg_(zl)
    prompts, cls, img = xl[:, :P], xl[:, P:P + 1], xl[:, P + 1:]
    h = F.gelu(F.linear(F.layer_norm(cls, (Lat,), prm["ca0_g"], prm["ca0_b"]), prm["ca1_w"], prm["ca1_b"]))
    imp = torch.sigmoid(F.linear(h, prm["ca3_w"], prm["ca3_b"]))
    gw = torch.sigmoid(F.linear(F.layer_norm(cls, (Lat,), prm["gl0_g"], prm["gl0_b"]), prm["gl1_w"], prm["gl1_b"]))

    def cross(qq, tok):
        w = torch.einsum("bpd,bnd->bpn", qq, tok) * Lat ** -0.5
        return torch.einsum("bpn,bnd->bpd", w.softmax(-1), tok)

    cg = cross(F.linear(prompts, prm["wgq"], prm["bgq"]), img[:, P + 1:])
    cl = cross(F.linear(prompts, prm["wlq"], prm["blq"]), ll)
    enh = (gw * cg + (1 - gw) * cl) * imp.transpose(1, 2)
    comb = torch.cat([enh, cls, img], 1)
    dcomb = _rand((B, T, Lat), 113)
    comb.backward(dcomb)

    f = lambda t: t.detach().float().to(dev).contiguous()
    z = lambda *s: torch.zeros(s, device=dev)
    pd = {k: f(v) for k, v in prm.items()}
    bufs = dict(imp=z(B, P), gw=z(B), enh=z(B, P, Lat), prm=z(B, P, Lat), qg=z(B, P, Lat), ql=z(B, P, Lat), cg=z(B, P, Lat), cl=z(B, P, Lat),
                lse_g=z(B, P), lse_l=z(B, P))
    common = dict(xl=f(xl.reshape(B * T, Lat)), ll=f(ll.reshape(B * N, Lat)), B=B, T=T, N=N, P=P, L=Lat, scale=Lat ** -0.5, **pd, **bufs)
    ops.gpa_fwd(**common)
    _close(bufs["imp"], imp.detach().reshape(B, P), 2e-5, "imp")
    _close(bufs["gw"], gw.detach().reshape(B), 2e-5, "gw")
    _close(bufs["enh"], enh.detach(), 3e-5, "enh")
    ng = ops.gpa_gate_param_count(Lat, P)
    bw = dict(dimp=z(B, P), dgw_part=z(B, P), dqg=z(B, P, Lat), dql=z(B, P, Lat), dcg=z(B, P, Lat), dcl=z(B, P, Lat), delta_g=z(B, P),
              delta_l=z(B, P), dprm=z(B, P, Lat), dcls=z(B, Lat), gate_partials=z(B, ng), dzx=z(B * T, Lat), dzl=z(B * N, Lat))
    ops.gpa_bwd(dcomb=f(dcomb.reshape(B * T, Lat)), zx=f(zx.reshape(B * T, Lat)), zl=f(zl.reshape(B * N, Lat)), **common, **bw)
    _close(bw["dzx"].view(B, T, Lat), zx.grad, 1e-4, "dzx")
    _close(bw["dzl"].view(B, N, Lat), zl.grad, 1e-4, "dzl")
    gp = bw["gate_partials"].cpu().double().sum(0)
    order = ["ca0_g", "ca0_b", "ca1_w", "ca1_b", "ca3_w", "ca3_b", "gl0_g", "gl0_b", "gl1_w", "gl1_b"]
    off = 0
    for k in order:
        n = prm[k].numel()
        _close(gp[off: off + n].view(prm[k].shape), prm[k].grad, 1e-4, "grad " + k)
        off += n
    assert off == ng
    # query projections: wgrad = dq^T . prompts, bias = colsum(dq)
    sc = z(64 * Lat * Lat)
    for nm, dq in (("wgq", bw["dqg"]), ("wlq", bw["dql"])):
        o = z(Lat, Lat)
        ops.small_wgrad(dq.view(B * P, Lat), bufs["prm"].view(B * P, Lat), o, sc, B * P, Lat, Lat)
        _close(o, prm[nm].grad, 1e-4, "grad " + nm)
        ob = z(Lat)
        ops.colsum(dq.view(B * P, Lat), ob, sc, B * P, Lat)
        _close(ob, prm["b" + nm[1:]].grad, 1e-4, "grad b" + nm[1:])


@pytest.mark.parametrize("B,T,C_,r0,R", [(2, 1033, 768, 0, 33), (3, 1001, 192, 0, 1), (2, 300, 1024, 0, 300)])
def test_head_fwd_bwd(dev, B, T, C_, r0, R):
    from gaviko_amd import ops
    K = 5
    g = (_rand((B, T, C_), 121, 3.0) + 0.5).requires_grad_(True)
    lg = (1 + _rand((C_,), 122, 0.2)); lb = _rand((C_,), 123, 0.1)
    wh = _rand((K, C_), 124, 0.1).requires_grad_(True); bh = _rand((K,), 125, 0.1).requires_grad_(True)
    pooled = F.layer_norm(g, (C_,), lg, lb, 1e-5)[:, r0:r0 + R].mean(1)
    logits = F.linear(pooled, wh, bh)
    dl = _rand((B, K), 126)
    logits.backward(dl)
    f = lambda t: t.detach().float().to(dev).contiguous()
    G = f(g.reshape(B * T, C_))
    lo = torch.zeros((B, K), device=dev); po = torch.zeros((B, C_), device=dev)
    kw = dict(g=G, ln_gamma=f(lg), ln_beta=f(lb), wh=f(wh), bh=f(bh), logits=lo, pooled=po, B=B, T=T, C=C_, K=K, r0=r0, R=R)
    ops.head_fwd(**kw)
    _close(lo, logits.detach(), 3e-5, "logits")
    dG = torch.zeros((B * T, C_), device=dev); dwh = torch.zeros((K, C_), device=dev); dbh = torch.zeros(K, device=dev)
    ops.head_bwd(dlogits=f(dl), dg=dG, dwh=dwh, dbh=dbh, accumulate=0, **kw)
    _close(dG.view(B, T, C_), g.grad, 5e-5, "dG")
    _close(dwh, wh.grad, 5e-5, "dwh")
    _close(dbh, bh.grad, 5e-5, "dbh")


def test_rows_broadcast_and_sum(dev):
    from gaviko_amd import ops
    B, T, C_, R, off = 3, 50, 192, 8, 1
    src = _rand((R, C_), 131); add = _rand((R, C_), 132)
    out = torch.full((B * T, C_), 9.0, device=dev)
    f = lambda t: t.float().to(dev).contiguous()
    ops.rows_broadcast(out, f(src), f(add), B, T, off, R, C_)
    o = out.view(B, T, C_).cpu().double()
    _close(o[:, off:off + R], (src + add).expand(B, R, C_), 1e-6, "rows_broadcast")
    assert (o[:, :off] == 9).all() and (o[:, off + R:] == 9).all()
    dg = _rand((B, T, C_), 133)
    a = torch.zeros((R, C_), device=dev); b2 = torch.ones((R, C_), device=dev)
    ops.rows_batch_sum(f(dg.reshape(B * T, C_)), a, None, B, T, off, R, C_)
    _close(a, dg[:, off:off + R].sum(0), 1e-5, "rows_batch_sum")


@pytest.mark.parametrize("M,C_,Lat", [(1000, 768, 20), (333, 192, 20), (100, 1024, 8)])
def test_skinny_up_ln_backward_epilogue_and_lowrank_affine(dev, M, C_, Lat):
    """y = LN(x) . Wd^T with upstream gradient dlat: dx, dWd, dgamma, dbeta, dbias via the fused kernels vs autograd."""
    from gaviko_amd import ops
    x = (_rand((M, C_), 141, 1.5) + 0.2).requires_grad_(True)
    g = (1 + _rand((C_,), 142, 0.2)).requires_grad_(True)
    b = _rand((C_,), 143, 0.1).requires_grad_(True)
    wd = _rand((Lat, C_), 144, 1 / math.sqrt(C_)).requires_grad_(True)
    bd = _rand((Lat,), 145, 0.1).requires_grad_(True)
    dlat = _rand((M, Lat), 146)
    dres = _rand((M, C_), 147)
    y = F.linear(F.layer_norm(x, (C_,), g, b, 1e-5), wd, bd)
    y.backward(dlat)
    f = lambda t: t.detach().float().to(dev).contiguous()
    X = f(x)
    mu = X.mean(1).contiguous()
    rstd = (X.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dx = torch.zeros((M, C_), device=dev)
    ops.skinny_up(lat=f(dlat), w=f(wd), res=f(dres), out=dx, ln_x=X, ln_mean=mu, ln_rstd=rstd, ln_gamma=f(g), M=M, C=C_, L=Lat, w_layout=1)
    _close(dx, x.grad + dres, 3e-5, "dx (LN' epilogue)")
    scratch = torch.zeros(ops.outer_scratch_elems(Lat, C_), device=dev)
    Q = torch.zeros((Lat, C_), device=dev)
    ops.outer_reduce(narrow=f(dlat), wide=X, mean=mu, rstd=rstd, scratch=scratch, out=Q, M=M, C=C_, L=Lat, transposed=0, accumulate=0)
    S = torch.zeros(Lat, device=dev)
    ops.reduce_batch([(f(dlat), None, S, 0)], torch.zeros(32 * (Lat + 64), device=dev))
    dW = torch.zeros((Lat, C_), device=dev); dg = torch.zeros(C_, device=dev); db = torch.zeros(C_, device=dev); dbias = torch.zeros(Lat, device=dev)
    ops.ln_lowrank_affine(Q, S, f(wd), f(g), f(b), dW, dg, db, dbias, Lat, C_)
    _close(dW, wd.grad, 5e-5, "dWd")
    _close(dg, g.grad, 5e-5, "dgamma")
    _close(db, b.grad, 5e-5, "dbeta")
    _close(dbias, bd.grad, 5e-5, "dbias")


@pytest.mark.parametrize("M,C", [(4132, 768), (258, 192), (1033, 1024)])
def test_layernorm_fwd_with_fused_projection(dev, M, C):
    from gaviko_amd import ops
    _g = torch.Generator(device=dev).manual_seed(M * 7 + C)
    rnd = lambda *sh, device: torch.randn(*sh, device=device, generator=_g)
    """gvk_layernorm_fwd_proj = LayerNorm (bf16 out, stats) + QuickGELU(x.Wd^T + b) of the raw rows (gaviko.py:155-156)."""
    L_ = 20
    x = rnd(M, C, device=dev)
    g, b = 1 + 0.2 * rnd(C, device=dev), 0.1 * rnd(C, device=dev)
    wd, bd = rnd(L_, C, device=dev) * C ** -0.5, 0.1 * rnd(L_, device=dev)
    y16 = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    z, y = torch.empty(M, L_, device=dev), torch.empty(M, L_, device=dev)
    ops.layernorm_fwd_proj(x, g, b, M, C, y16=y16, mean=mean, rstd=rstd, w=wd, bias=bd, z=z, y=y, act=1, w_layout=0)
    want = torch.nn.functional.layer_norm(x, (C,), g, b)
    assert (y16.float() - want).abs().max() < 3e-2
    assert torch.allclose(mean, x.mean(-1), atol=1e-5) and torch.allclose(rstd, (x.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-5)
    zz = (x.double() @ wd.double().t() + bd.double()).float()
    assert torch.allclose(z, zz, atol=2e-5, rtol=1e-5)
    assert torch.allclose(y, zz * torch.sigmoid(1.702 * zz), atol=2e-5, rtol=1e-5)
    # and the unfused pair produces the same numbers
    z2, y2 = torch.empty_like(z), torch.empty_like(y)
    ops.skinny_down(x=x, w=wd, bias=bd, z=z2, y=y2, M=M, C=C, L=L_, act=1, w_layout=0)
    assert torch.allclose(y, y2, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("M,C", [(4132, 768), (258, 192)])
def test_layernorm_bwd_with_fused_projection(dev, M, C):
    from gaviko_amd import ops
    _g = torch.Generator(device=dev).manual_seed(M + C)
    rnd = lambda *sh, device: torch.randn(*sh, device=device, generator=_g)
    """gvk_layernorm_bwd_proj = LayerNorm backward (dx, bf16 copy) + dx . W_up ([C][L] weight, autograd of gaviko.py:187)."""
    L_ = 20
    x, dy, dres = rnd(M, C, device=dev), rnd(M, C, device=dev), rnd(M, C, device=dev)
    g = 1 + 0.2 * rnd(C, device=dev)
    wup = rnd(C, L_, device=dev) * C ** -0.5
    mean, rstd = x.mean(-1).contiguous(), (x.var(-1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dx, dx2 = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev)
    dx16 = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    y = torch.empty(M, L_, device=dev)
    ops.layernorm_bwd_proj(dy, x, mean, rstd, g, M, C, dx=dx, dres=dres, dx16=dx16, w=wup, y=y, w_layout=1)
    ops.layernorm_bwd(dy, x, mean, rstd, g, M, C, dx=dx2, dres=dres)
    # same fp32 arithmetic, different order of the two row sums (wave shuffle tree vs four-lane + eight-wave partials): round-off apart
    assert (dx - dx2).abs().max().item() <= 4e-6 * dx2.abs().max().item()
    assert (dx16.float() - dx).abs().max() <= dx.abs().max() * 2 ** -8
    assert torch.allclose(y, (dx.double() @ wup.double()).float(), atol=3e-5, rtol=1e-5)


def test_layernorm_bwd_with_bf16_gradient_input(dev):
    """gvk_layernorm_bwd_dy16: the three backward forms (all rows / the leading rows of every sample / with the rank-L projection) reading the
    output gradient as bf16 -- the same numbers as the fp32-input kernels fed the upcast values (bit-equal for the two plain forms: one kernel)."""
    from gaviko_amd import ops
    B, T, C, L_, P = 4, 1033, 768, 20, 32
    M = B * T
    g_ = torch.Generator(device=dev).manual_seed(11)
    rnd = lambda *sh: torch.randn(*sh, device=dev, generator=g_)
    x, dres = rnd(M, C), rnd(M, C)
    dy16 = rnd(M, C).bfloat16()
    dy = dy16.float()
    g = 1 + 0.2 * rnd(C)
    wup = rnd(C, L_) * C ** -0.5
    mean, rstd = x.mean(-1).contiguous(), (x.var(-1, unbiased=False) + 1e-5).rsqrt().contiguous()
    a, b = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev)
    a16, b16 = (torch.empty(M, C, dtype=torch.bfloat16, device=dev) for _ in range(2))
    ops.layernorm_bwd_dy16(dy16, x, mean, rstd, g, M, C, dx=a, dres=dres, dx16=a16)
    ops.layernorm_bwd(dy, x, mean, rstd, g, M, C, dx=b, dres=dres, dx16=b16)
    assert torch.equal(a, b) and torch.equal(a16, b16)
    a.fill_(7.0); b.fill_(7.0)
    ops.layernorm_bwd_dy16(dy16, x, mean, rstd, g, M, C, dx=a, dres=dres, rows=(B, P, T))
    ops.layernorm_bwd_rows(dy, x, mean, rstd, g, B, P, T, C, dx=b, dres=dres)
    assert torch.equal(a, b)
    assert bool((a[P:T] == 7.0).all()) and bool((a[T:T + P] != 7.0).any())
    ya, yb = torch.empty(M, L_, device=dev), torch.empty(M, L_, device=dev)
    ops.layernorm_bwd_dy16(dy16, x, mean, rstd, g, M, C, dx=a, dres=dres, dx16=a16, proj=dict(w=wup, y=ya, w_layout=1, L_=L_))
    ops.layernorm_bwd_proj(dy, x, mean, rstd, g, M, C, dx=b, dres=dres, dx16=b16, w=wup, y=yb, w_layout=1)
    assert (a - b).abs().max().item() <= 4e-6 * b.abs().max().item()
    assert torch.allclose(ya, yb, atol=3e-5, rtol=1e-5)
    with pytest.raises(ops.L.GavikoHipError, match="all rows"):
        ops.layernorm_bwd_dy16(dy16, x, mean, rstd, g, M, C, dx=a, rows=(B, P, T), proj=dict(w=wup, y=ya, w_layout=1, L_=L_))


def test_fused_projection_rejects_other_latent_widths(dev):
    from gaviko_amd import ops
    rnd = lambda *sh, device: torch.randn(*sh, device=device)
    x = rnd(64, 768, device=dev)
    with pytest.raises(ops.L.GavikoHipError, match="gvk_skinny_down"):
        ops.layernorm_fwd_proj(x, x[0], x[0], 64, 768, y16=torch.empty(64, 768, dtype=torch.bfloat16, device=dev),
                               w=rnd(32, 768, device=dev), y=torch.empty(64, 32, device=dev), L_=32)


def test_outer_reduce_two_sources(dev):
    """One weight gradient fed by two token streams in one pass: out = n1^T . w1 + n2^T . w2."""
    from gaviko_amd import ops
    gen = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=gen).to(dev)  # noqa: E731
    M1, M2, C, L = 4132, 4000, 768, 20
    n1, w1, n2, w2 = r(M1, L), r(M1, C), r(M2, L), r(M2, C)
    sc = torch.zeros(ops.outer_scratch_elems(L, C), device=dev)
    out = torch.zeros(L, C, device=dev)
    ops.outer_reduce(narrow=n1, wide=w1, narrow2=n2, wide2=w2, scratch=sc, out=out, M=M1, M2=M2, C=C, L=L, transposed=0, accumulate=0)
    want = n1.double().t() @ w1.double() + n2.double().t() @ w2.double()
    assert (out.double() - want).abs().max().item() < 2e-3
    two = torch.zeros(L, C, device=dev)
    ops.outer_reduce(narrow=n1, wide=w1, scratch=sc, out=two, M=M1, C=C, L=L, transposed=0, accumulate=0)
    ops.outer_reduce(narrow=n2, wide=w2, scratch=sc, out=two, M=M2, C=C, L=L, transposed=0, accumulate=1)
    assert (out - two).abs().max().item() < 2e-3
    with pytest.raises(Exception, match="exceeds"):
        ops.outer_reduce(narrow=n1, wide=w1, narrow2=r(8000, L), wide2=r(8000, C), scratch=sc, out=out, M=M1, M2=8000, C=C, L=L, transposed=0, accumulate=0)


def test_reduce_batch_column_sum_over_two_sources(dev):
    from gaviko_amd import ops
    gen = torch.Generator().manual_seed(5)
    a, a2, b = torch.randn(4132, 20, generator=gen).to(dev), torch.randn(4000, 20, generator=gen).to(dev), torch.randn(4132, 20, generator=gen).to(dev)
    out, w = torch.ones(20, device=dev), torch.zeros(20, 20, device=dev)
    sc = torch.zeros(32 * (20 + 400), device=dev)
    ops.reduce_batch([(a, None, out, 1, a2), (a, b, w, 0)], sc)
    want = 1.0 + a.double().sum(0) + a2.double().sum(0)
    assert (out.double() - want).abs().max().item() < 1e-3
    assert (w.double() - a.double().t() @ b.double()).abs().max().item() < 1e-2


def test_reduce_batch_mixed_row_counts(dev):
    """One launch pair, jobs of very different heights (the GPA's call: 4 rows into 3525 gate-parameter sums, 128-row 20 x 20 products, an
    8132-row column sum): every job gets its own number of row slabs; accumulate and overwrite both honoured; bitwise repeatable."""
    from gaviko_amd import ops
    gen = torch.Generator().manual_seed(6)
    r = lambda *s: torch.randn(*s, generator=gen).to(dev)
    gp, dq, prm, dz, dz2 = r(4, 3525), r(128, 20), r(128, 20), r(4132, 20), r(4000, 20)
    tall_a, tall_b = r(4000, 60), r(4000, 20)
    total = 3525 + 400 + 20 + 20 + 1200
    sc = torch.zeros(32 * total, device=dev)
    outs = []
    for _ in range(2):
        o_g, o_w, o_b, o_c, o_t = torch.ones(3525, device=dev), torch.zeros(20, 20, device=dev), torch.zeros(20, device=dev), torch.zeros(20, device=dev), torch.zeros(60, 20, device=dev)
        ops.reduce_batch([(gp, None, o_g, 1), (dq, prm, o_w, 0), (dq, None, o_b, 0), (dz, None, o_c, 0, dz2), (tall_a, tall_b, o_t, 0)], sc)
        outs.append((o_g, o_w, o_b, o_c, o_t))
    o_g, o_w, o_b, o_c, o_t = outs[0]
    assert (o_g.double() - (1.0 + gp.double().sum(0))).abs().max().item() < 1e-5
    assert (o_w.double() - dq.double().t() @ prm.double()).abs().max().item() < 1e-4
    assert (o_b.double() - dq.double().sum(0)).abs().max().item() < 1e-4
    assert (o_c.double() - (dz.double().sum(0) + dz2.double().sum(0))).abs().max().item() < 1e-3
    assert (o_t.double() - tall_a.double().t() @ tall_b.double()).abs().max().item() < 1e-2
    for u, v in zip(*outs):
        assert torch.equal(u, v)


@pytest.mark.parametrize("M,C", [(4132, 768), (2002, 192), (2066, 1024), (37, 768)])
def test_layernorm_bwd_up_matches_two_kernels(dev, M, C):
    """gvk_layernorm_bwd_up (sidepass.hip, LayerNorm' + rank-20 update in one pass) against gvk_layernorm_bwd followed by the accumulating
    gvk_skinny_up it replaces, and against float64 torch."""
    from gaviko_amd import ops
    L = 20
    g_ = torch.Generator().manual_seed(M + C)
    r = lambda *s: torch.randn(*s, generator=g_)
    dy, x, dres, lat, w, gamma = r(M, C), r(M, C) * 2 + 0.3, r(M, C), r(M, L), r(L, C) * 0.05, r(C)
    mean, var = x.double().mean(-1), x.double().var(-1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    xh = (x.double() - mean[:, None]) * rstd[:, None]
    dh = dy.double() * gamma.double()
    want = dres.double() + rstd[:, None] * (dh - dh.mean(-1, keepdim=True) - xh * (dh * xh).mean(-1, keepdim=True)) + lat.double() @ w.double()
    t = lambda a: a.float().to(dev).contiguous()
    dyd, xd, dresd, latd, wd, gd, md, rd = t(dy), t(x), t(dres), t(lat), t(w), t(gamma), t(mean), t(rstd)
    dx = torch.zeros(M, C, device=dev); dx16 = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
    ops.layernorm_bwd_up(dyd, xd, md, rd, gd, M, C, dx=dx, dres=dresd, dx16=dx16, lat=latd, w=wd, L_=L, w_layout=1)
    ref = torch.zeros(M, C, device=dev); ref16 = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
    ops.layernorm_bwd(dyd, xd, md, rd, gd, M, C, dx=ref, dres=dresd)
    ops.skinny_up(lat=latd, w=wd, out=ref, out_bf16=ref16, M=M, C=C, L=L, w_layout=1, accumulate=1)
    sc = want.abs().max().item()
    assert (dx.cpu().double() - want).abs().max().item() < 2e-6 * sc
    assert (dx - ref).abs().max().item() < 4e-6 * sc
    assert torch.equal(dx16, dx.bfloat16())
    # the transposed weight layout ([C][L]) gives the same result
    dx2 = torch.zeros(M, C, device=dev); dx216 = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
    ops.layernorm_bwd_up(dyd, xd, md, rd, gd, M, C, dx=dx2, dres=dresd, dx16=dx216, lat=latd, w=wd.t().contiguous(), L_=L, w_layout=0)
    assert (dx2 - dx).abs().max().item() < 2e-6 * sc


@pytest.mark.parametrize("M,C,p", [(4000, 768, 0.2), (2000, 192, 0.0), (1000, 1024, 0.2), (37, 768, 0.2)])
def test_skinny_up_layer_boundary_form_matches_three_kernels(dev, M, C, p):
    """gvk_skinny_up with lat_b: out = res + LN'(lat . W^T) + lat_b . W_b^T and z2 = (out o dropout mask) . W2^T in one pass, against the
    three launches it replaces in the MWSA backward (LayerNorm-backward up-projection, accumulate up-projection, dropout down-projection)."""
    from gaviko_amd import ops
    Lat = 20
    f = lambda t: t.float().to(dev).contiguous()
    lat, lat_b = f(_rand((M, Lat), 1)), f(_rand((M, Lat), 2))
    wd, wb, wup = f(_rand((Lat, C), 3, 0.1)), f(_rand((Lat, C), 4, 0.1)), f(_rand((C, Lat), 5, 0.1))
    res, x = f(_rand((M, C), 6)), f(_rand((M, C), 7, 2.0))
    gamma = f(1 + 0.3 * _rand((C,), 8))
    mean = x.mean(-1).contiguous()
    rstd = (x.var(-1, unbiased=False) + 1e-5).rsqrt().contiguous()
    seed = torch.full((1,), 77, dtype=torch.int64, device=dev)
    # the three kernels
    out3 = torch.empty_like(res)
    ops.skinny_up(lat=lat, w=wd, res=res, out=out3, ln_x=x, ln_mean=mean, ln_rstd=rstd, ln_gamma=gamma, M=M, C=C, L=Lat, w_layout=1)
    ops.skinny_up(lat=lat_b, w=wb, out=out3, M=M, C=C, L=Lat, w_layout=1, accumulate=1)
    z3 = torch.zeros((M, Lat), device=dev)
    ops.skinny_down(x=out3, w=wup, y=z3, M=M, C=C, L=Lat, act=0, w_layout=1, drop_p=p, seed=5, seed_ptr=seed)
    # one kernel
    out1, z1 = torch.empty_like(res), torch.zeros((M, Lat), device=dev)
    ops.skinny_up(lat=lat, w=wd, res=res, out=out1, ln_x=x, ln_mean=mean, ln_rstd=rstd, ln_gamma=gamma, M=M, C=C, L=Lat, w_layout=1,
                  lat_b=lat_b, w_b=wb, w2=wup, z2=z1, L2=Lat, act2=0, w2_layout=1, drop2_p=p, seed2=5, seed_ptr=seed)
    torch.cuda.synchronize()
    _close(out1, out3.double().cpu(), 2e-6, "boundary out")
    _close(z1, z3.double().cpu(), 2e-5, "boundary dctx")


@pytest.mark.parametrize("M,C,p", [(4000, 768, 0.2), (2000, 192, 0.0), (1000, 1024, 0.0), (37, 768, 0.2)])
def test_skinny_up_with_next_layer_entry_matches_two_kernels(dev, M, C, p):
    """gvk_skinny_up with nx_w: the up-projection (+ residual, dropout, GPA's second projection) and the NEXT layer's LayerNorm + proj_down +
    qkv of the rows it writes, against the up kernel followed by the entry kernel (gvk_skinny_down with the LayerNorm input)."""
    from gaviko_amd import ops
    Lat = 20
    f = lambda t: t.float().to(dev).contiguous()
    lat, wup, bup, res = f(_rand((M, Lat), 1)), f(_rand((C, Lat), 2, 0.2)), f(_rand((C,), 3, 0.1)), f(_rand((M, C), 4))
    w2, b2 = f(_rand((Lat, C), 5, 0.1)), f(_rand((Lat,), 6, 0.1))
    wd, bd, g, b = f(_rand((Lat, C), 7, 0.1)), f(_rand((Lat,), 8, 0.1)), f(1 + 0.3 * _rand((C,), 9)), f(_rand((C,), 10, 0.2))
    wqkv = f(_rand((3 * Lat, Lat), 11, 0.3))
    seed = torch.full((1,), 91, dtype=torch.int64, device=dev)
    mk = lambda *s_: torch.zeros(s_, device=dev)
    up = dict(lat=lat, w=wup, bias=bup, res=res, M=M, C=C, L=Lat, w_layout=0, drop_p=p, seed=3, seed_ptr=seed, w2=w2, bias2=b2, L2=Lat, act2=1)
    out_a, z_a, y_a = mk(M, C), mk(M, Lat), mk(M, Lat)
    ops.skinny_up(out=out_a, z2=z_a, y2=y_a, **up)
    mean_a, rstd_a, lat_a, qkv_a = mk(M), mk(M), mk(M, Lat), mk(M, 3 * Lat)
    ops.skinny_down(x=out_a, w=wd, bias=bd, ln_gamma=g, ln_beta=b, mean=mean_a, rstd=rstd_a, y=lat_a, w2=wqkv, y2=qkv_a, M=M, C=C, L=Lat, L2=3 * Lat,
                    act=0, w_layout=0, eps=1e-5)
    out_b, z_b, y_b = mk(M, C), mk(M, Lat), mk(M, Lat)
    mean_b, rstd_b, lat_b, qkv_b = mk(M), mk(M), mk(M, Lat), mk(M, 3 * Lat)
    ops.skinny_up(out=out_b, z2=z_b, y2=y_b, nx_w=wd, nx_bias=bd, nx_ln_gamma=g, nx_ln_beta=b, nx_mean=mean_b, nx_rstd=rstd_b, nx_lat=lat_b,
                  nx_w2=wqkv, nx_y2=qkv_b, nx_L2=3 * Lat, nx_eps=1e-5, **up)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b) and torch.equal(y_a, y_b)
    _close(mean_b, mean_a.double().cpu(), 2e-6, "next mean")
    _close(rstd_b, rstd_a.double().cpu(), 2e-6, "next rstd")
    _close(lat_b, lat_a.double().cpu(), 1e-5, "next lat")
    _close(qkv_b, qkv_a.double().cpu(), 1e-5, "next qkv")


@pytest.mark.parametrize("C,L,B", [(768, 20, 4), (192, 20, 2), (1024, 20, 2), (768, 8, 1)])
def test_param_grads_one_launch(dev, C, L, B):
    """gvk_param_grads (csrc/paramgrad.hip): every parameter gradient of a side-path module of one layer in ONE launch -- two outer
    products (override rows, a second token stream, a dropout mask, the LayerNorm-affine epilogue) and the small reductions -- against
    float64, in both job mixes the engine issues (GPA stream / MWSA stream); overwrite and accumulate; bitwise repeatable (the partial
    tiles are summed in slab order whichever workgroup arrives last) and the ticket words come back zero."""
    import dropmask
    from gaviko_amd import ops
    gen = torch.Generator().manual_seed(11 + C + L)
    r = lambda *s: torch.randn(*s, generator=gen).to(dev)  # noqa: E731
    T, P, N = 1033, 32, 1000
    M, BN = B * T, B * N
    tick = torch.zeros(ops.PGRAD_TICKETS, dtype=torch.int32, device=dev)
    seedw = torch.full((1,), 77123, dtype=torch.int64, device=dev)
    # ---- GPA mix: proj_up (override rows, transposed, colsum), proj_down over two token streams, six small jobs
    xl, enh, dG = r(M, L), r(B, P, L), r(M, C)
    dzx, G1, dzl, Lc = r(M, L), r(M, C), r(BN, L), r(BN, C)
    gp, dq, prm = r(B, 3525), r(B * P, L), r(B * P, L)
    nct = (C + 63) // 64
    sc = torch.zeros(ops.param_grads_scratch_elems(L, [nct, nct], [3525, L * L, L, L * L, L, L]), device=dev)
    outs = {}
    for acc in (0, 1, 1):
        if acc == 0:
            dWup, dbup, dWd = torch.zeros(C, L, device=dev), torch.zeros(C, device=dev), torch.zeros(L, C, device=dev)
            gflat, wq, bq, wq2, bq2, bd = (torch.zeros(n, device=dev) for n in (3525, L * L, L, L * L, L, L))
        ops.param_grads(
            [dict(narrow=xl, wide=dG, lat_override=enh, out=dWup, colsum=dbup, M=M, T=T, P=P, transposed=1, accumulate=acc),
             dict(narrow=dzx, wide=G1, narrow2=dzl, wide2=Lc, out=dWd, M=M, M2=BN, transposed=0, accumulate=acc)],
            [(gp, None, gflat, acc), (dq, prm, wq, acc), (dq, None, bq, acc), (dq, prm, wq2, acc), (dq, None, bq2, acc), (dzx, None, bd, acc, dzl)],
            sc, tick, C, L)
        outs[len(outs)] = [t.clone() for t in (dWup, dbup, dWd, gflat, wq, bq, bd)]
    comb = xl.double().view(B, T, L).clone()
    comb[:, :P] = enh.double()
    comb = comb.view(M, L)
    want = [dG.double().t() @ comb, dG.double().sum(0), dzx.double().t() @ G1.double() + dzl.double().t() @ Lc.double(), gp.double().sum(0),
            (dq.double().t() @ prm.double()).reshape(-1), dq.double().sum(0), dzx.double().sum(0) + dzl.double().sum(0)]
    for k, (got, w) in enumerate(zip(outs[0], want)):
        assert (got.double() - w).abs().max().item() < 2e-5 * max(1.0, w.abs().max().item()) * (M ** 0.5), k
    for k, (got, w) in enumerate(zip(outs[2], want)):                 # overwrite + two accumulating calls = 3 x
        assert (got.double() - 3 * w).abs().max().item() < 6e-5 * max(1.0, w.abs().max().item()) * (M ** 0.5), k
    assert int(tick.abs().max()) == 0
    # bitwise repeatable
    a1, a2 = torch.zeros(L, C, device=dev), torch.zeros(L, C, device=dev)
    for dst in (a1, a2):
        ops.param_grads([dict(narrow=dzx, wide=G1, narrow2=dzl, wide2=Lc, out=dst, M=M, M2=BN)], [], sc, tick, C, L)
    assert torch.equal(a1, a2)
    # ---- MWSA mix: proj_up behind proj_drop (mask regenerated), LayerNorm + proj_down through the affine epilogue, the qkv matrix
    p_drop, site = 0.2, 2 * 7 + 1
    ctx, dL = r(BN, L), r(BN, C)
    dlat, lin = r(BN, L), r(BN, C) * 2 + 0.5
    mean, var = lin.mean(1), lin.var(1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    Wd, gam, bet = r(L, C) * 0.1, 1 + 0.2 * r(C), 0.1 * r(C)
    dqkv, lat = r(BN, 3 * L), r(BN, L)
    sc2 = torch.zeros(ops.param_grads_scratch_elems(L, [nct, nct, 1], [3 * L * L]), device=dev)
    res = {}
    for acc in (0, 1):
        if acc == 0:
            dWu, dbu, dWdn = torch.zeros(C, L, device=dev), torch.zeros(C, device=dev), torch.zeros(L, C, device=dev)
            dgam, dbet, dbias, dWqkv = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(L, device=dev), torch.zeros(3 * L, L, device=dev)
            dWqkv2 = torch.zeros(3 * L, L, device=dev)
        ops.param_grads(
            [dict(narrow=ctx, wide=dL, out=dWu, colsum=dbu, M=BN, transposed=1, accumulate=acc, drop_p=p_drop, seed=site),
             dict(narrow=dlat, wide=lin, mean=mean, rstd=rstd, out=dWdn, aff_w=Wd, aff_gamma=gam, aff_beta=bet, aff_dgamma=dgam, aff_dbeta=dbet,
                  aff_dbias=dbias, M=BN, accumulate=acc),
             dict(narrow=lat, wide=dqkv, out=dWqkv, M=BN, C=3 * L, transposed=1, accumulate=acc)],          # a 3L-wide "wide": the qkv matrix
            [(dqkv, lat, dWqkv2, acc)], sc2, tick, C, L, seed_ptr=seedw)
        res[acc] = [t.clone() for t in (dWu, dbu, dWdn, dgam, dbet, dbias, dWqkv, dWqkv2)]
    mask = torch.from_numpy(dropmask.rows_mask(site + 77123, BN, C, p_drop)).to(dev).double()
    dLm = dL.double() * mask
    xhat = (lin.double() - mean.double()[:, None]) * rstd.double()[:, None]
    Q, S = dlat.double().t() @ xhat, dlat.double().sum(0)
    want2 = [dLm.t() @ ctx.double(), dLm.sum(0), gam.double()[None, :] * Q + bet.double()[None, :] * S[:, None], (Wd.double() * Q).sum(0),
             (Wd.double() * S[:, None]).sum(0), S, dqkv.double().t() @ lat.double(), dqkv.double().t() @ lat.double()]
    for k, (got, w) in enumerate(zip(res[0], want2)):
        assert (got.double() - w).abs().max().item() < 2e-5 * max(1.0, w.abs().max().item()) * (BN ** 0.5), k
    for k, (got, w) in enumerate(zip(res[1], want2)):
        assert (got.double() - 2 * w).abs().max().item() < 4e-5 * max(1.0, w.abs().max().item()) * (BN ** 0.5), k
    assert int(tick.abs().max()) == 0
    with pytest.raises(Exception, match="scratch holds"):
        ops.param_grads([dict(narrow=ctx, wide=dL, out=dWu, M=BN, transposed=1)], [], torch.zeros(64, device=dev), tick, C, L)


def test_param_grads_hand_off_beside_a_stream_that_dirties_the_caches(dev):
    """The cross-workgroup hand-off of gvk_param_grads (write-through partials, drained, ticket, last arriver acquires) issues no release
    fence (gaviko_hip.h), and its bitwise-repeat check above runs on an otherwise idle GPU.  Here it runs the way the step runs it: on a
    side stream, while another stream streams and REWRITES a buffer far larger than the L2s (dirty lines in every XCD, every CU's L1
    warm with foreign lines), launch after launch with the previous call's partials still in the scratch -- every result must equal the
    quiet one bit for bit and float64 within the usual bound, and the tickets must come back zero."""
    from gaviko_amd import ops
    C, L, B, T, N = 768, 20, 4, 1033, 1000
    gen = torch.Generator().manual_seed(5)
    r = lambda *s_: torch.randn(*s_, generator=gen).to(dev)  # noqa: E731
    M, BN = B * T, B * N
    dzx, G1, dzl, Lc = r(M, L), r(M, C), r(BN, L), r(BN, C)
    nct = (C + 63) // 64
    sc = torch.zeros(ops.param_grads_scratch_elems(L, [nct], []), device=dev)
    tick = torch.zeros(ops.PGRAD_TICKETS, dtype=torch.int32, device=dev)
    job = lambda dst: ops.param_grads([dict(narrow=dzx, wide=G1, narrow2=dzl, wide2=Lc, out=dst, M=M, M2=BN)], [], sc, tick, C, L)  # noqa: E731
    quiet = torch.zeros(L, C, device=dev)
    job(quiet)
    torch.cuda.synchronize()
    want = dzx.double().t() @ G1.double() + dzl.double().t() @ Lc.double()
    assert (quiet.double() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item()) * (M ** 0.5)
    big = torch.zeros(96 * 1024 * 1024, device=dev)                       # 384 MiB: beyond the 32 MiB of L2 and the Infinity Cache
    side, noise = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [torch.zeros(L, C, device=dev) for _ in range(12)]
    torch.cuda.synchronize()
    with torch.cuda.stream(noise):
        for _ in range(24):
            big.add_(1.0)                                                 # read-modify-write of every line, on every CU
    with torch.cuda.stream(side):
        for o in outs:
            sc.normal_()                                                  # stale partials of "the previous layer" in the scratch
            job(o)
    torch.cuda.synchronize()
    assert all(torch.equal(o, quiet) for o in outs)
    assert int(tick.abs().max()) == 0


def test_attention_bwd_first_rows_only(dev):
    """gvk_attention_bwd_bf16_rows: dq, dk, dv of the first need_rows tokens of every sample (rounded up to the 128-row blocks) carry the
    very bits of the full call, every other row of dqkv is left untouched, delta is complete."""
    from gaviko_amd import ops
    B, T, H = 2, 1033, 3
    inner = H * 64
    qkv = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev)
    qkv[: B * T] = _rand((B * T, 3 * inner), 71, 1.0).bfloat16().to(dev)
    ops.qkv_prescale(qkv, B * T, H, 0.125)
    O = ops.act_zeros(B * T, inner, torch.bfloat16, dev)
    lse = torch.zeros((B, H, T), device=dev)
    ops.attention_fwd(qkv, O, lse, B, T, H, 0.125, q_prescaled=True)
    DO = ops.act_zeros(B * T, inner, torch.bfloat16, dev)
    DO[: B * T] = _rand((B * T, inner), 72, 1.0).bfloat16().to(dev)
    full, part = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev), ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev)
    part.fill_(3.0)
    d_full, d_part = torch.zeros((B, H, T), device=dev), torch.zeros((B, H, T), device=dev)
    ops.attention_bwd(qkv, O, DO, lse, d_full, full, B, T, H, 0.125, q_prescaled=True)
    ops.attention_bwd(qkv, O, DO, lse, d_part, part, B, T, H, 0.125, q_prescaled=True, need_rows=32)
    torch.cuda.synchronize()
    assert torch.equal(d_full, d_part)
    f, p = full[: B * T].view(B, T, -1), part[: B * T].view(B, T, -1)
    assert torch.equal(f[:, :128], p[:, :128])
    assert bool((p[:, 128:] == 3.0).all())
