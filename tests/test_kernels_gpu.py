"""-m gpu: each HIP kernel (through the C-ABI) against a plain fp32/fp64 torch computation on the CPU."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _bf16_round(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("M,N,K,tile", [(300, 192, 192, 0), (1033, 768, 768, 128128), (1033, 768, 768, 64064),
                                        (2066, 2304, 768, 0), (517, 576, 192, 128064), (517, 768, 3072, 64128),
                                        (517, 768, 3072, 3128128), (300, 256, 64, 3128128), (300, 256, 128, 3128128), (1033, 768, 192, 3128128),
                                        (4132, 768, 768, 0), (4132, 768, 3072, 0),
                                        # 96-row three-stage tiles: M = 128 k (the last row tile reaches past the 128-row padding of A and clamps),
                                        # the ViT-L B = 2 shapes that pick it (176 tiles in one round), a ragged M
                                        (256, 256, 512, 3096128), (2066, 1024, 1024, 3096128), (2066, 1024, 4096, 0), (1001, 128, 576, 3096128)])
def test_gemm_store_bf16_and_f32(dev, M, N, K, tile):
    from gaviko_amd import ops
    a = _bf16_round(_rand((M, K), 1))
    w = _bf16_round(_rand((N, K), 2, 1 / math.sqrt(K)))
    bias = _rand((N,), 3, 0.1)
    ref = a.double() @ w.double().T
    A = ops.act_zeros(M, K, torch.bfloat16, dev)
    A[:M] = a.to(dev).bfloat16()
    W = w.to(dev).bfloat16().contiguous()
    out = torch.full((ops.pad_rows(M), N), 7.0, dtype=torch.float32, device=dev)
    ops.gemm_nt(A, W, M, out, epilogue=ops.EPI_STORE_F32, tile=tile)
    torch.cuda.synchronize()
    got = out[:M].cpu().double()
    assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    assert (out[M:] == 7.0).all(), "rows >= M must never be stored"
    out16 = torch.zeros((ops.pad_rows(M), N), dtype=torch.bfloat16, device=dev)
    ops.gemm_nt(A, W, M, out16, epilogue=ops.EPI_STORE_BF16, bias=bias.to(dev), tile=tile)
    got = out16[:M].cpu().double()
    refb = ref + bias.double()
    assert (got - refb).abs().max().item() < 1.5 * 2 ** -8 * refb.abs().max().item()


def test_gemm_epilogues(dev):
    from gaviko_amd import ops
    M, N, K = 1033, 768, 192
    a = _bf16_round(_rand((M, K), 11))
    w = _bf16_round(_rand((N, K), 12, 2 / math.sqrt(K)))
    bias = _rand((N,), 13, 0.2)
    res = _rand((M, N), 14)
    ref = a.double() @ w.double().T + bias.double()
    A = ops.act_zeros(M, K, torch.bfloat16, dev)
    A[:M] = a.to(dev).bfloat16()
    W = w.to(dev).bfloat16().contiguous()
    # bias + residual (in place on the residual buffer) and the bf16 twin
    R = ops.act_zeros(M, N, torch.float32, dev)
    R[:M] = res.to(dev)
    O16 = ops.act_zeros(M, N, torch.bfloat16, dev)
    ops.gemm_nt(A, W, M, R, epilogue=ops.EPI_BIAS_RES_F32_BF16, out1=O16, bias=bias.to(dev), res=R)
    want = ref + res.double()
    assert (R[:M].cpu().double() - want).abs().max().item() < 3e-4
    assert (O16[:M].cpu().double() - want).abs().max().item() < 2 ** -7 * want.abs().max().item()
    # bias + erf-GELU, saving the pre-activation
    pre = ops.act_zeros(M, N, torch.bfloat16, dev)
    act = ops.act_zeros(M, N, torch.bfloat16, dev)
    ops.gemm_nt(A, W, M, pre, epilogue=ops.EPI_BIAS_GELU_BF16, out1=act, bias=bias.to(dev))
    assert (pre[:M].cpu().double() - ref).abs().max().item() < 2 ** -7 * ref.abs().max().item()
    g = torch.nn.functional.gelu(ref)
    assert (act[:M].cpu().double() - g).abs().max().item() < 2 ** -7 * max(1.0, g.abs().max().item())
    # dgrad fused with GELU'
    ops.gemm_nt(A, W, M, act, epilogue=ops.EPI_GELU_BWD_BF16, aux=pre)
    x = pre[:M].cpu().double().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    want = (a.double() @ w.double().T) * x.grad
    assert (act[:M].cpu().double() - want).abs().max().item() < 2 ** -6 * want.abs().max().item()


@pytest.mark.parametrize("M,N,K,tile", [(1033, 768, 192, 0), (4132, 2304, 768, 0), (4132, 2304, 768, 128128), (300, 576, 192, 64064)])
def test_gemm_store_bf16_column_scale(dev, M, N, K, tile):
    """gvk_gemm_desc.scale_cols: the first N/3 columns (the q block of a qkv projection) leave the STORE_BF16 epilogue as
    (acc + bias) * col_scale with ONE rounding -- bit for bit the bf16 of the fp32 product -- and the other columns are untouched.
    Covers the eight-phase 256 x 256 kernel (the auto choice at M = 4132, N = 2304) and the four-wave tiles."""
    from gaviko_amd import ops
    a = _bf16_round(_rand((M, K), 15))
    w = _bf16_round(_rand((N, K), 16, 2 / math.sqrt(K)))
    bias = _rand((N,), 17, 0.2)
    A = ops.act_zeros(M, K, torch.bfloat16, dev)
    A[:M] = a.to(dev).bfloat16()
    W = w.to(dev).bfloat16().contiguous()
    cs = 0.125 * 1.4426950408889634
    plain, scaled = ops.act_zeros(M, N, torch.bfloat16, dev), ops.act_zeros(M, N, torch.bfloat16, dev)
    f32 = ops.act_zeros(M, N, torch.float32, dev)
    ops.gemm_nt(A, W, M, plain, epilogue=ops.EPI_STORE_BF16, bias=bias.to(dev), tile=tile)
    ops.gemm_nt(A, W, M, scaled, epilogue=ops.EPI_STORE_BF16, bias=bias.to(dev), tile=tile, scale_cols=N // 3, col_scale=cs)
    ops.gemm_nt(A, W, M, f32, epilogue=ops.EPI_STORE_F32, bias=bias.to(dev), tile=tile)
    assert torch.equal(scaled[:M, N // 3:], plain[:M, N // 3:])
    want = (f32[:M, : N // 3] * cs).bfloat16()                # same fp32 accumulator, same fp32 multiply, one rounding
    assert torch.equal(scaled[:M, : N // 3], want)
    assert (scaled[M:] == 0).all()
    with pytest.raises(Exception, match="scale_cols"):
        ops.gemm_nt(A, W, M, f32, epilogue=ops.EPI_STORE_F32, scale_cols=N // 3, col_scale=cs)


@pytest.mark.parametrize("M,C_,N", [(4132, 768, 2304), (2066, 1024, 3072), (300, 128, 384)])
def test_layernorm_folded_into_gemm(dev, M, C_, N):
    """The first LayerNorm of a layer folded into its qkv projection (vision_transformer.py:49,61-62): the producer GEMM (BIAS_RES_F32_BF16)
    leaves the fp32 rows, their bf16 copy and per-row (sum, sum of squares) partials; gvk_prompt_up_fix_stats fixes P prompt rows and turns
    the partials into mean / rstd; the consumer GEMM runs on the RAW bf16 rows against gamma o W with rstd*(acc - mean*c1) + beta.W^T in its
    epilogue.  Against float64 LayerNorm + Linear of the fp32 rows (a row mean of ~2 standard deviations is included on purpose)."""
    from gaviko_amd import ops
    B, P, Lt, K0 = 2, 8, 20, 256
    T = M // B
    M = B * T
    a0 = _bf16_round(_rand((M, K0), 21))
    w0 = _bf16_round(_rand((C_, K0), 22, 2.0 / math.sqrt(K0)))
    bias0 = _rand((C_,), 23, 0.3) + 1.5                         # a common offset: row mean ~ 1.5 against a row std ~ 2
    res = _rand((M, C_), 24, 1.0)
    x_ref = a0.double() @ w0.double().T + bias0.double() + res.double()
    enh, lat, wup = _rand((B, P, Lt), 25), _rand((M, Lt), 26), _rand((C_, Lt), 27, 0.3)
    for b in range(B):
        x_ref[b * T: b * T + P] += (enh[b].double() - lat[b * T: b * T + P].double()) @ wup.double().T
    A = ops.act_zeros(M, K0, torch.bfloat16, dev); A[:M] = a0.to(dev).bfloat16()
    G = ops.act_zeros(M, C_, torch.float32, dev)
    R = ops.act_zeros(M, C_, torch.float32, dev); R[:M] = res.to(dev)
    G16 = ops.act_zeros(M, C_, torch.bfloat16, dev)
    part = torch.zeros((C_ // 64) * M * 2, device=dev)
    ops.gemm_nt(A, w0.to(dev).bfloat16().contiguous(), M, G, epilogue=ops.EPI_BIAS_RES_F32_BF16, out1=G16, bias=bias0.to(dev), res=R, stat_part=part)
    mean, rstd = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
    ops.prompt_up_fix_stats(enh.to(dev).contiguous(), lat.to(dev).contiguous(), wup.to(dev).contiguous(), G, G16, part, mean, rstd, B, T, P, C_, Lt)
    torch.cuda.synchronize()
    assert (G[:M].cpu().double() - x_ref).abs().max().item() < 2e-4 * x_ref.abs().max().item()
    assert torch.equal(G16[:M], G[:M].bfloat16())               # the bf16 copy is the rounding of the stored fp32 rows, prompt rows included
    mu_ref, var_ref = x_ref.mean(1), x_ref.var(1, unbiased=False)
    assert (mean.cpu().double() - mu_ref).abs().max().item() < 1e-5 * max(1.0, mu_ref.abs().max().item())
    assert ((rstd.cpu().double() - (var_ref + 1e-5).rsqrt()) / (var_ref + 1e-5).rsqrt()).abs().max().item() < 1e-5
    # consumer: y = LN(x) . W^T with the LayerNorm in the epilogue (+ the q-block pre-scale of the real call site)
    gamma, beta = 1 + _rand((C_,), 28, 0.2), _rand((C_,), 29, 0.1)
    W = _rand((N, C_), 30, 1.0 / math.sqrt(C_))
    Wf = (W * gamma[None, :]).to(dev).bfloat16().contiguous()
    c1 = Wf.float().sum(1).contiguous()
    c2 = (W * beta[None, :]).sum(1).to(dev).contiguous()
    Y = ops.act_zeros(M, N, torch.bfloat16, dev)
    cs = 0.125 * 1.4426950408889634
    ops.gemm_nt(G16, Wf, M, Y, epilogue=ops.EPI_STORE_BF16, bias=c2, ln_mean=mean, ln_rstd=rstd, ln_c1=c1, scale_cols=N // 3, col_scale=cs)
    xg = G[:M].cpu().double()
    y_ref = torch.nn.functional.layer_norm(xg, (C_,), gamma.double(), beta.double(), 1e-5) @ W.double().T
    y_ref[:, : N // 3] *= cs
    got = Y[:M].cpu().double()
    err = (got - y_ref).abs().max().item() / y_ref.abs().max().item()
    # the unfolded path for scale: LayerNorm kernel (bf16 out) + plain GEMM
    xn = ops.act_zeros(M, C_, torch.bfloat16, dev)
    ops.layernorm_fwd(G, gamma.to(dev), beta.to(dev), M, C_, y16=xn)
    Y2 = ops.act_zeros(M, N, torch.bfloat16, dev)
    ops.gemm_nt(xn, W.to(dev).bfloat16().contiguous(), M, Y2, epilogue=ops.EPI_STORE_BF16, scale_cols=N // 3, col_scale=cs)
    err2 = (Y2[:M].cpu().double() - y_ref).abs().max().item() / y_ref.abs().max().item()
    print(f"LN fold M={M} C={C_} N={N}: folded rel err {err:.2e}, LayerNorm kernel + GEMM {err2:.2e}")
    assert err < 1.2e-2 and err < 2.5 * err2 + 2e-3


@pytest.mark.parametrize("offset", [0.0, 40.0, 3000.0])
def test_fold_row_statistics_with_large_common_offset(dev, offset):
    """Rows whose mean dwarfs their spread (|mean| / std up to 1500): the fc2 epilogue accumulates its (sum, sum of squares) partials of
    x - pivot[m] (pivot = the residual row's mean, as the engine passes it), so mean / rstd match the two-pass LayerNorm kernel; the
    unshifted single-pass form E[x^2] - mean^2 returns var = 0 (rstd = 1/sqrt(eps)) at the largest offset."""
    from gaviko_amd import ops
    B, T, P, Lt, K0, C_ = 2, 300, 4, 20, 128, 768
    M = B * T
    a0 = _bf16_round(_rand((M, K0), 41))
    w0 = _bf16_round(_rand((C_, K0), 42, 2.0 / math.sqrt(K0)))
    bias0 = _rand((C_,), 43, 0.3)
    res = _rand((M, C_), 44, 1.0) + offset * (1.0 + _rand((M, 1), 45, 0.5))      # every row its own large offset
    A = ops.act_zeros(M, K0, torch.bfloat16, dev); A[:M] = a0.to(dev).bfloat16()
    G = ops.act_zeros(M, C_, torch.float32, dev)
    R = ops.act_zeros(M, C_, torch.float32, dev); R[:M] = res.to(dev)
    G16 = ops.act_zeros(M, C_, torch.bfloat16, dev)
    part = torch.zeros((C_ // 64) * M * 2, device=dev)
    pivot = R[:M].mean(1).contiguous()
    ops.gemm_nt(A, w0.to(dev).bfloat16().contiguous(), M, G, epilogue=ops.EPI_BIAS_RES_F32_BF16, out1=G16, bias=bias0.to(dev), res=R, stat_part=part,
                stat_pivot=pivot)
    mean, rstd = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
    enh, lat, wup = _rand((B, P, Lt), 46), _rand((M, Lt), 47), _rand((C_, Lt), 48, 0.3)
    ops.prompt_up_fix_stats(enh.to(dev).contiguous(), lat.to(dev).contiguous(), wup.to(dev).contiguous(), G, G16, part, mean, rstd, B, T, P, C_, Lt,
                            pivot=pivot)
    m2, r2 = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
    xn = ops.act_zeros(M, C_, torch.bfloat16, dev)
    ops.layernorm_fwd(G, torch.ones(C_, device=dev), torch.zeros(C_, device=dev), M, C_, y16=xn, mean=m2, rstd=r2)     # two-pass kernel, same rows
    torch.cuda.synchronize()
    x = G[:M].cpu().double()
    mu_ref, rs_ref = x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()
    assert (mean.cpu().double() - mu_ref).abs().max().item() < 2e-6 * max(1.0, mu_ref.abs().max().item())
    assert ((rstd.cpu().double() - rs_ref) / rs_ref).abs().max().item() < 2e-4
    assert ((rstd - r2) / r2).abs().max().item() < 2e-4


def test_patch_embed_path(dev):
    """patchify + GEMM(PATCH epilogue) == conv3d + flatten/transpose + pos, scattered into [P+1 .. ] rows."""
    from gaviko_amd import ops
    from gaviko_amd.utils import synth
    B, C_, P = 2, 192, 32
    img = torch.from_numpy(synth.volumes(5, B))
    w = _rand((C_, 1, 12, 16, 16), 21, math.sqrt(3.0 / 3072) * 1.7)
    bias = _rand((C_,), 22, 0.05)
    pos = _rand((1000, C_), 23, 0.3)
    wb = _bf16_round(w)
    ref = torch.nn.functional.conv3d(_bf16_round(img).double(), wb.double(), bias.double(), stride=(12, 16, 16))
    ref = ref.flatten(2).transpose(1, 2) + pos.double()
    T = P + 1 + 1000
    cols = ops.act_zeros(B * 1000, 3072, torch.bfloat16, dev)
    ops.patchify(img.to(dev), cols, (12, 16, 16))
    G = torch.full((ops.pad_rows(B * T), C_), -5.0, dtype=torch.float32, device=dev)
    Lo = torch.zeros((ops.pad_rows(B * 1000), C_), dtype=torch.float32, device=dev)
    W = wb.reshape(C_, 3072).to(dev).bfloat16().contiguous()
    ops.gemm_nt(cols, W, B * 1000, G, epilogue=ops.EPI_PATCH_F32, out1=Lo, bias=bias.to(dev), pos=pos.to(dev).contiguous(),
                rows_in=1000, rows_out=T, row_off=P + 1)
    g = G[: B * T].view(B, T, C_).cpu().double()
    assert (g[:, P + 1:] - ref).abs().max().item() < 5e-4
    assert (g[:, : P + 1] == -5.0).all()
    assert (Lo[: B * 1000].view(B, 1000, C_).cpu().double() - ref).abs().max().item() < 5e-4


@pytest.mark.parametrize("M,C_", [(1033, 768), (77, 192), (515, 1024), (9, 384)])
def test_layernorm_fwd_bwd(dev, M, C_):
    from gaviko_amd import ops
    x = _rand((M, C_), 31, 2.0) + 0.3
    gamma = 1 + _rand((C_,), 32, 0.2)
    beta = _rand((C_,), 33, 0.1)
    dy = _rand((M, C_), 34)
    dres = _rand((M, C_), 35)
    xd = x.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = beta.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xd, (C_,), gd, bd, 1e-5)
    y.backward(dy.double())
    X, Gm, Bt = x.to(dev), gamma.to(dev), beta.to(dev)
    y16 = torch.zeros((M, C_), dtype=torch.bfloat16, device=dev)
    y32 = torch.zeros((M, C_), dtype=torch.float32, device=dev)
    mean = torch.zeros(M, device=dev)
    rstd = torch.zeros(M, device=dev)
    ops.layernorm_fwd(X, Gm, Bt, M, C_, y16=y16, y32=y32, mean=mean, rstd=rstd)
    assert (y32.cpu().double() - y.detach()).abs().max().item() < 2e-5
    assert (y16.cpu().double() - y.detach()).abs().max().item() < 2 ** -7 * y.abs().max().item()
    dx = torch.zeros((M, C_), device=dev)
    dx16 = torch.zeros((M, C_), dtype=torch.bfloat16, device=dev)
    ops.layernorm_bwd(dy.to(dev), X, mean, rstd, Gm, M, C_, dx=dx, dres=dres.to(dev), dx16=dx16)
    want = xd.grad + dres.double()
    assert (dx.cpu().double() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
    assert (dx16.cpu().double() - want).abs().max().item() < 2 ** -7 * want.abs().max().item()
    dg = torch.zeros(C_, device=dev)
    db = torch.zeros(C_, device=dev)
    scratch = torch.zeros(128 * C_, device=dev)
    ops.layernorm_bwd_affine(dy.to(dev), X, mean, rstd, dg, db, scratch, M, C_)
    assert (dg.cpu().double() - gd.grad).abs().max().item() < 1e-4 * max(1.0, gd.grad.abs().max().item())
    assert (db.cpu().double() - bd.grad).abs().max().item() < 1e-4 * max(1.0, bd.grad.abs().max().item())


def test_casts(dev):
    from gaviko_amd import ops
    x = _rand((771, 193), 41)
    o = ops.cast_bf16(x.to(dev).contiguous())
    assert torch.equal(o.cpu(), x.bfloat16())
    t = ops.transpose_cast_bf16(x.to(dev).contiguous())
    assert torch.equal(t.cpu(), x.t().contiguous().bfloat16())


ATTN_C = 0.125 * 1.4426950408889634       # what the q block carries when it reaches the bf16 attention kernels: q * scale * log2(e)


def _attn_ref(qkv, B, T, H):
    q, k, v = (t.reshape(B, T, H, 64).permute(0, 2, 1, 3) for t in qkv.double().chunk(3, dim=-1))
    s = q @ k.transpose(-1, -2) * 0.125
    return (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, T, H * 64), torch.logsumexp(s, -1)


def _attn_operands(qkv32, H):
    """fp32 q | k | v -> (device operand bf16 [.., 3*H*64] with the q block pre-scaled by scale*log2(e) and rounded ONCE, as the qkv GEMM's
    epilogue delivers it; the float64 q | k | v those operands represent exactly -- what the reference attends over)."""
    inner = H * 64
    dev_op = qkv32.clone()
    dev_op[..., :inner] *= ATTN_C
    dev_op = dev_op.bfloat16()
    exact = dev_op.double()
    exact[..., :inner] /= ATTN_C
    return dev_op, exact


@pytest.mark.parametrize("amp", [1.0, 2.5])
@pytest.mark.parametrize("B,T,H", [(2, 1033, 3), (1, 1001, 12), (3, 65, 2), (1, 393, 3), (2, 128, 1)])
def test_attention_fwd(dev, B, T, H, amp):
    """amp = standard deviation of the q / k / v entries (2.5: scores up to +-50, far hotter than any layer of the model).  The operand is
    what the engine hands the kernel: the q block pre-scaled by scale*log2(e) with ONE rounding to bf16; the reference attends over
    exactly the values those bf16 operands represent."""
    from gaviko_amd import ops
    inner = H * 64
    op, exact = _attn_operands(_rand((B, T, 3 * inner), 51, amp), H)
    ref, lse_ref = _attn_ref(exact, B, T, H)
    Q = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev)
    Q[: B * T] = op.reshape(B * T, -1).to(dev)
    O = ops.act_zeros(B * T, inner, torch.bfloat16, dev)
    lse = torch.zeros((B, H, T), device=dev)
    ops.attention_fwd(Q, O, lse, B, T, H, 0.125, q_prescaled=True)
    torch.cuda.synchronize()
    got = O[: B * T].view(B, T, inner).cpu().double()
    lse_err, o_err = (lse.cpu().double() - lse_ref).abs().max().item(), (got - ref).abs().max().item()
    print(f"attention_fwd B={B} T={T} H={H} amp={amp}: lse err {lse_err:.2e}, O err {o_err:.2e} (|O| max {ref.abs().max().item():.2f})")
    assert lse_err < 2e-3
    assert o_err < 1.5e-2 * max(1.0, ref.abs().max().item())
    # a raw qkv through the convenience path (ops makes the pre-scaled copy: one more rounding of q, hence the looser lse bound)
    raw = _bf16_round(_rand((B, T, 3 * inner), 51, amp))
    ref2, lse2 = _attn_ref(raw, B, T, H)
    Q[: B * T] = raw.reshape(B * T, -1).to(dev).bfloat16()
    ops.attention_fwd(Q, O, lse, B, T, H, 0.125)
    assert (lse.cpu().double() - lse2).abs().max().item() < (2e-3 if amp <= 1.0 else 1e-2)
    assert (O[: B * T].view(B, T, inner).cpu().double() - ref2).abs().max().item() < 1.5e-2 * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize("step", [3.0, 1.0, 0.25])
def test_attention_fwd_forced_rescale(dev, step):
    """Online-softmax rescale branch: one key per tile dominates, with the max growing tile after tile.  The kernel raises its running
    maximum only when a tile exceeds it by more than 8 log2 units (attention_fwd.hip kThr): step 3.0 crosses that at every spike (the
    slow path each time), step 1.0 every third spike (deferred in between: P up to 2^8 at the stale maximum), step 0.25 never after the
    first (everything accumulated against a maximum that is up to ~2.6 log2 units stale)."""
    from gaviko_amd import ops
    B, T, H = 1, 300, 1
    qkv = _bf16_round(_rand((B, T, 192), 52, 0.5))
    for j, t in enumerate((10, 70, 110, 140, 200, 250, 299)):
        qkv[0, t, 64:128] = qkv[0, 5, 0:64] * (4.0 + step * j)     # key t aligned with query 5, growing
    op, exact = _attn_operands(qkv, H)
    ref, lse_ref = _attn_ref(exact, B, T, H)
    Q = ops.act_zeros(T, 192, torch.bfloat16, dev)
    Q[:T] = op.reshape(T, -1).to(dev)
    O = ops.act_zeros(T, 64, torch.bfloat16, dev)
    lse = torch.zeros((1, 1, T), device=dev)
    ops.attention_fwd(Q, O, lse, B, T, H, 0.125, q_prescaled=True)
    got = O[:T].view(1, T, 64).cpu().double()
    assert (got - ref).abs().max().item() < 1.5e-2 * max(1.0, ref.abs().max().item())
    assert (lse.cpu().double() - lse_ref).abs().max().item() < 5e-3


@pytest.mark.parametrize("var", [0, 1, 2, 3] if os.environ.get("GAVIKO_HIP_DIAG", "0") == "1" else [0])     # variants 1-3: measurement build only
@pytest.mark.parametrize("kb", [96, 128])
@pytest.mark.parametrize("T", [1033, 1001, 97, 31])
def test_attention_fwd_key_tiles(dev, monkeypatch, T, kb, var):
    """Both key-tile sizes on every sequence length class (the launcher picks the one that pads less; GAVIKO_HIP_ATTN_KB forces one):
    the last tile's key mask rides the augmented MFMA, rows past the sequence are staged from clamped addresses."""
    from gaviko_amd import lib, ops
    if var != 0 and not lib.DIAG:
        pytest.skip("kernel variants other than the shipped one exist in the diag library only (GAVIKO_HIP_DIAG=1)")
    monkeypatch.setenv("GAVIKO_HIP_ATTN_KB", str(kb))
    monkeypatch.setenv("GAVIKO_HIP_ATTN_VAR", str(var))        # bit 0: row sums on the matrix pipe; bit 1: LDS-DMA spread over the S^T blocks
    B, H = 2, 2
    inner = H * 64
    op, exact = _attn_operands(_rand((B, T, 3 * inner), 57 + T, 2.0), H)
    ref, lse_ref = _attn_ref(exact, B, T, H)
    Q = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev)
    Q[: B * T] = op.reshape(B * T, -1).to(dev)
    O = ops.act_zeros(B * T, inner, torch.bfloat16, dev)
    lse = torch.zeros((B, H, T), device=dev)
    ops.attention_fwd(Q, O, lse, B, T, H, 0.125, q_prescaled=True)
    torch.cuda.synchronize()
    got = O[: B * T].view(B, T, inner).cpu().double()
    assert (lse.cpu().double() - lse_ref).abs().max().item() < 2e-3
    assert (got - ref).abs().max().item() < 1.5e-2 * max(1.0, ref.abs().max().item())
    assert torch.isfinite(O.float()).all() and (O[B * T:] == 0).all()          # padding rows of the output stay untouched


# ---- fp32 compute path (BASELINE cfg4: fp32, tolerance 1e-5) ------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (1001, 192, 192), (130, 64, 3072)])
def test_gemm_f32_epilogues(dev, M, N, K):
    """gvk_gemm_nt_f32 (v_mfma_f32_16x16x4_f32) against float64, every epilogue the engine uses on the fp32 path."""
    from gaviko_amd import ops
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = ops.act_zeros(M, K, torch.float32, dev)
    a[:M] = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    bias = torch.randn(N, device=dev, generator=g)
    res = ops.act_zeros(M, N, torch.float32, dev); res[:M] = torch.randn(M, N, device=dev, generator=g)
    aux = ops.act_zeros(M, N, torch.float32, dev); aux[:M] = torch.randn(M, N, device=dev, generator=g)
    ref = a[:M].double() @ w.double().t()
    tol = dict(atol=2e-5, rtol=1e-5)
    out0, out1 = ops.act_zeros(M, N, torch.float32, dev), ops.act_zeros(M, N, torch.float32, dev)
    ops.gemm_nt(a, w, M, out0, epilogue=ops.EPI_STORE_BF16)
    assert torch.allclose(out0[:M].double(), ref, **tol)
    ops.gemm_nt(a, w, M, out0, epilogue=ops.EPI_BIAS_RES_F32, bias=bias, res=res)
    assert torch.allclose(out0[:M].double(), ref + bias.double() + res[:M].double(), **tol)
    ops.gemm_nt(a, w, M, out0, epilogue=ops.EPI_BIAS_GELU_BF16, bias=bias, out1=out1)
    pre = ref + bias.double()
    assert torch.allclose(out0[:M].double(), pre, **tol)
    assert torch.allclose(out1[:M].double(), torch.nn.functional.gelu(pre), **tol)
    ops.gemm_nt(a, w, M, out0, epilogue=ops.EPI_GELU_BWD_BF16, aux=aux)
    x = aux[:M].double().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    assert torch.allclose(out0[:M].double(), ref * x.grad, **tol)
    ops.gemm_nt(a, w, M, out0, epilogue=ops.EPI_BIAS_RELU_BF16, bias=bias)
    assert torch.allclose(out0[:M].double(), pre.clamp_min(0), **tol)
    ops.gemm_nt(a, w, M, out0, epilogue=ops.EPI_RELU_BWD_BF16, aux=aux)
    assert torch.allclose(out0[:M].double(), ref * (aux[:M] > 0), **tol)
    assert out0[M:].abs().max() == 0                      # rows beyond M are never stored


@pytest.mark.parametrize("B,T,H", [(2, 197, 2), (1, 1001, 1), (2, 65, 3), (1, 7, 2), (2, 33, 1), (1, 129, 1), (1, 1, 1)])
def test_attention_f32_fwd_bwd(dev, B, T, H):
    """fp32 flash attention (vision_transformer.py:63-71 and its autograd) against float64 torch."""
    from gaviko_amd import ops
    inner = H * 64
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = torch.randn(B, T, 3 * inner, generator=g, dtype=torch.float64).requires_grad_(True)
    dO = torch.randn(B, T, inner, generator=g, dtype=torch.float64)
    q, k, v = (t.reshape(B, T, H, 64).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    s = q @ k.transpose(-1, -2) * 0.125
    o = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, T, inner)
    o.backward(dO)
    Q = qkv.detach().reshape(B * T, -1).float().to(dev).contiguous()
    O = torch.zeros(B * T, inner, device=dev)
    lse = torch.zeros(B, H, T, device=dev)
    ops.attention_fwd(Q, O, lse, B, T, H, 0.125)
    assert torch.allclose(O.cpu().double(), o.detach().reshape(B * T, -1), atol=2e-6, rtol=1e-5)
    assert torch.allclose(lse.cpu().double(), torch.logsumexp(s.detach(), -1), atol=1e-5, rtol=1e-6)
    dQ = torch.zeros_like(Q)
    delta = torch.zeros(B, H, T, device=dev)
    ops.attention_bwd(Q, O, dO.reshape(B * T, -1).float().to(dev).contiguous(), lse, delta, dQ, B, T, H, 0.125)
    want = qkv.grad.reshape(B * T, -1)
    assert (dQ.cpu().double() - want).abs().max() <= 1e-5 * want.abs().max()


@pytest.mark.parametrize("M,N,K", [(4132, 3072, 768), (1033, 768, 192), (300, 256, 64), (2066, 2304, 768)])
def test_gemm_256x256_eight_wave_tile(dev, M, N, K):
    """tile 256256 (8 waves, 64 x 128 per wave): the three epilogues of the wide GEMMs, against the 128 x 128 kernel and torch."""
    from gaviko_amd import ops
    gen = torch.Generator().manual_seed(M + N + K)
    A = ops.act_zeros(M, K, torch.bfloat16, dev); A[:M] = torch.randn(M, K, generator=gen).bfloat16().to(dev)
    W = (torch.randn(N, K, generator=gen) / K ** 0.5).bfloat16().to(dev)
    bias = torch.randn(N, generator=gen).to(dev)
    aux = ops.act_zeros(M, N, torch.bfloat16, dev); aux[:M] = torch.randn(M, N, generator=gen).bfloat16().to(dev)
    ref = A[:M].float() @ W.float().t()
    for epi, kw in ((ops.EPI_STORE_BF16, dict(bias=bias)), (ops.EPI_BIAS_GELU_BF16, dict(bias=bias)), (ops.EPI_GELU_BWD_BF16, dict(aux=aux))):
        outs = []
        for tile in (256256, 128128):
            o0, o1 = ops.act_zeros(M, N, torch.bfloat16, dev), ops.act_zeros(M, N, torch.bfloat16, dev)
            k2 = dict(kw, out1=o1) if epi == ops.EPI_BIAS_GELU_BF16 else kw
            ops.gemm_nt(A, W, M, o0, epilogue=epi, tile=tile, **k2)
            outs.append((o0[:M].float(), o1[:M].float()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])      # same k order per element: bit-identical
    o = ops.act_zeros(M, N, torch.bfloat16, dev)
    ops.gemm_nt(A, W, M, o, epilogue=ops.EPI_STORE_BF16, tile=256256)
    assert (o[:M].float() - ref).abs().max().item() < 3e-2 * max(1.0, ref.abs().max().item())
    with pytest.raises(Exception, match="256x256"):
        ops.gemm_nt(A, W, M, torch.zeros(M, N, device=dev), epilogue=ops.EPI_STORE_F32, tile=256256)


@pytest.mark.parametrize("tile8", [8256256, 7256256])
@pytest.mark.parametrize("M,N,K", [(4132, 3072, 768), (4132, 768, 3072), (1033, 768, 192), (300, 256, 128), (2066, 2304, 768), (130, 512, 2304)])
def test_gemm_eight_phase_kernel(dev, M, N, K, tile8):
    """tile 8256256 (gemm8p_bf16.hip: 256 x 256, two wave groups one barrier apart, counted vmcnt): every epilogue it is built for,
    bit for bit against the four-wave 128 x 128 kernel (same k order per element) and against torch in float32."""
    from gaviko_amd import ops
    gen = torch.Generator().manual_seed(7 * M + N + K)
    A = ops.act_zeros(M, K, torch.bfloat16, dev); A[:M] = torch.randn(M, K, generator=gen).bfloat16().to(dev)
    W = (torch.randn(N, K, generator=gen) / K ** 0.5).bfloat16().to(dev)
    bias = torch.randn(N, generator=gen).to(dev)
    aux = ops.act_zeros(M, N, torch.bfloat16, dev); aux[:M] = torch.randn(M, N, generator=gen).bfloat16().to(dev)
    res = ops.act_zeros(M, N, torch.float32, dev); res[:M] = torch.randn(M, N, generator=gen).to(dev)
    ref = A[:M].float() @ W.float().t()
    cases = ((ops.EPI_STORE_BF16, dict(bias=bias), torch.bfloat16), (ops.EPI_BIAS_GELU_BF16, dict(bias=bias), torch.bfloat16),
             (ops.EPI_GELU_BWD_BF16, dict(aux=aux), torch.bfloat16), (ops.EPI_STORE_F32, dict(), torch.float32),
             (ops.EPI_BIAS_RES_F32, dict(bias=bias, res=res), torch.float32))
    for rep in range(3):                                    # the LDS-DMA ordering is a race if it is wrong: repeat
        for epi, kw, dt in cases:
            outs = []
            for tile in (tile8, 128128):
                o0, o1 = ops.act_zeros(M, N, dt, dev), ops.act_zeros(M, N, torch.bfloat16, dev)
                k2 = dict(kw, out1=o1) if epi == ops.EPI_BIAS_GELU_BF16 else kw
                ops.gemm_nt(A, W, M, o0, epilogue=epi, tile=tile, **k2)
                outs.append((o0.float(), o1.float()))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), f"epilogue {epi} differs from the 128x128 kernel"
            assert not outs[0][0][M:].any(), "rows >= M must not be written"
    o = ops.act_zeros(M, N, torch.float32, dev)
    ops.gemm_nt(A, W, M, o, epilogue=ops.EPI_STORE_F32, tile=tile8)
    assert (o[:M] - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())



def test_gemm_gelu_derivative_stored_by_the_forward(dev):
    """gvk_gemm_desc.aux_is_grad: the forward GEMM's epilogue leaves bf16 GELU'(pre) (beside GELU(pre)), the dgrad GEMM multiplies by it --
    the same dgrad as evaluating the derivative from the stored pre-activation, to bf16 rounding of the factor; both tile families."""
    from gaviko_amd import ops
    g = torch.Generator().manual_seed(5)
    M, C, H = 4132, 768, 3072
    x = ops.act_zeros(M, C, torch.bfloat16, dev); x[:M] = torch.randn(M, C, generator=g).bfloat16().to(dev)
    w1 = (torch.randn(H, C, generator=g) * C ** -0.5).bfloat16().to(dev)
    b1 = (torch.randn(H, generator=g) * 0.1).to(dev)
    dy = ops.act_zeros(M, C, torch.bfloat16, dev); dy[:M] = torch.randn(M, C, generator=g).bfloat16().to(dev)
    w2t = (torch.randn(H, C, generator=g) * C ** -0.5).bfloat16().to(dev)            # fc2 weight transposed: dgrad as an NT GEMM
    for tile in (0, 3128128):
        pre, gp, act, act2 = (ops.act_zeros(M, H, torch.bfloat16, dev) for _ in range(4))
        ops.gemm_nt(x, w1, M, pre, epilogue=ops.EPI_BIAS_GELU_BF16, out1=act, bias=b1, tile=tile)
        ops.gemm_nt(x, w1, M, gp, epilogue=ops.EPI_BIAS_GELU_BF16, out1=act2, bias=b1, tile=tile, aux_is_grad=1)
        assert torch.equal(act[:M], act2[:M])                                        # the activation itself does not change
        h = (x[:M].float() @ w1.float().t() + b1).double()
        ref = 0.5 * (1 + torch.erf(h / 2 ** 0.5)) + h * torch.exp(-0.5 * h * h) / (2 * torch.pi) ** 0.5
        assert (gp[:M].double() - ref).abs().max().item() < 6e-3                     # bf16 rounding of a value in [-0.13, 1.13] (+ the bf16 operands)
        d0, d1 = ops.act_zeros(M, H, torch.bfloat16, dev), ops.act_zeros(M, H, torch.bfloat16, dev)
        ops.gemm_nt(dy, w2t, M, d0, epilogue=ops.EPI_GELU_BWD_BF16, aux=pre, tile=tile)
        ops.gemm_nt(dy, w2t, M, d1, epilogue=ops.EPI_GELU_BWD_BF16, aux=gp, tile=tile, aux_is_grad=1)
        torch.cuda.synchronize()
        err = (d0[:M].float() - d1[:M].float()).abs().max().item()
        assert err < 3e-2 * d0[:M].float().abs().max().item(), err
    with pytest.raises(Exception, match="aux_is_grad"):
        ops.gemm_nt(x, w1, M, pre, epilogue=ops.EPI_STORE_BF16, aux_is_grad=1)


def test_gemm_strided_row_panels_and_split_k(dev):
    """gvk_gemm_desc.m_panels / m_stride: only the 64-row tiles at rows 0, T, 2T, ... are computed (the first rows of every sample) -- those
    rows must be bit-identical to the full launch on the same 4-wave tile, every other row of the output untouched; with a split-K
    workspace the K loop of every tile is cut into pieces summed in a fixed order: equal to the unsplit rows within fp32 accumulation
    noise, bitwise repeatable, ticket words back at zero."""
    from gaviko_amd import ops
    B, T, N, K = 4, 1033, 768, 3072
    M = B * T
    g = torch.Generator().manual_seed(3)
    a = ops.act_zeros(M, K, torch.bfloat16, dev)
    a[:M] = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(dev)
    full = torch.zeros(M, N, device=dev)
    ops.gemm_nt(a, w, M, full, epilogue=ops.EPI_STORE_F32, tile=3064128)
    part = torch.full((M, N), 7.0, device=dev)
    ops.gemm_nt(a, w, M, part, epilogue=ops.EPI_STORE_F32, m_panels=B, m_stride=T)
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(b * T, b * T + 64) for b in range(B)]).to(dev)
    mask = torch.zeros(M, dtype=torch.bool, device=dev)
    mask[rows] = True
    assert torch.equal(part[mask], full[mask])
    assert bool((part[~mask] == 7.0).all())
    ws = torch.zeros((1024 + 256 * 32768) // 4, dtype=torch.int32, device=dev)
    outs = []
    for _ in range(3):
        o = torch.full((M, N), 7.0, device=dev)
        ops.gemm_nt(a, w, M, o, epilogue=ops.EPI_STORE_F32, m_panels=B, m_stride=T, splitk_ws=ws)
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and int(ws[:256].abs().max()) == 0
    assert bool((outs[0][~mask] == 7.0).all())
    ref = a[:M].float()[rows] @ w.float().t()
    assert (outs[0][mask] - ref).abs().max().item() < 2e-5 * ref.abs().max().item() * (K ** 0.5)
    assert (outs[0][mask] - full[mask]).abs().max().item() < 1e-5 * full[mask].abs().max().item()
    with pytest.raises(Exception, match="row panels"):
        ops.gemm_nt(a, w, M, part, epilogue=ops.EPI_STORE_F32, m_panels=B + 2, m_stride=T)
    # a FULL launch in two pieces per tile (tile 2128128: 396 workgroups, two per CU): close to the unsplit product, bitwise repeatable, and
    # the row panels cut the same way (ksplit = 2) carry the same bits as its rows
    ws2 = torch.zeros((1024 + 256 * 2 * 65536) // 4, dtype=torch.int32, device=dev)
    f2 = []
    for _ in range(3):
        o = torch.zeros(M, N, device=dev)
        ops.gemm_nt(a, w, M, o, epilogue=ops.EPI_STORE_F32, tile=2128128, splitk_ws=ws2, ksplit=2)
        torch.cuda.synchronize()
        f2.append(o)
    assert torch.equal(f2[0], f2[1]) and torch.equal(f2[0], f2[2]) and int(ws2[:256].abs().max()) == 0
    assert (f2[0] - full).abs().max().item() < 1e-5 * full.abs().max().item()
    p2 = torch.full((M, N), 7.0, device=dev)
    ops.gemm_nt(a, w, M, p2, epilogue=ops.EPI_STORE_F32, m_panels=B, m_stride=T, splitk_ws=ws, ksplit=2)
    torch.cuda.synchronize()
    assert torch.equal(p2[mask], f2[0][mask]) and bool((p2[~mask] == 7.0).all())
