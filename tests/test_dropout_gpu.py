"""nn.Dropout sites of the unfrozen-backbone methods at kernel level: every mask is a pure function of (seed word, index) that the host
can rebuild (tests/dropmask.py), so each kernel is checked against a torch fp32 computation with the SAME mask."""
import numpy as np
import pytest
import torch

import dropmask

pytestmark = pytest.mark.gpu
P = 0.1


@pytest.fixture(scope="module")
def dev():
    from gaviko_amd import lib
    lib.require_device()
    return torch.device("cuda:0")


def _seed(dev, value):
    return torch.tensor([value], dtype=torch.int64, device=dev)


def test_dropout_rows_exact_and_row_mapping(dev):
    from gaviko_amd import ops
    M, N = 300, 192
    x = torch.randn(M, N, device=dev)
    word = _seed(dev, 123456789)
    o32, o16 = torch.empty_like(x), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.dropout_rows(x, P, 77, word, out32=o32, out16=o16)
    mask = torch.from_numpy(dropmask.rows_mask(77 + 123456789, M, N, P)).to(dev)
    assert torch.equal(o32, x * mask) and torch.equal(o16, (x * mask).bfloat16())
    keep = (mask > 0).float().mean().item()
    assert abs(keep - (1 - P)) < 0.01
    # rows 3..3+5 of each of 4 samples of 20 rows, in place
    B, T, R = 4, 20, 5
    g = torch.randn(B * T, N, device=dev)
    want = g.clone()
    m2 = torch.from_numpy(dropmask.rows_mask(5 + 123456789, B * R, N, 0.3)).to(dev).view(B, R, N)
    want.view(B, T, N)[:, 3:3 + R] *= m2
    ops.dropout_rows(g, 0.3, 5, word, out32=g, M=B * R, N=N, rows_in=R, rows_out=T, row_off=3)
    assert torch.equal(g, want)


@pytest.mark.parametrize("M,N,K,tile", [(517, 768, 192, 0), (1033, 256, 128, 128128)])
def test_gemm_dropout_epilogues(dev, M, N, K, tile):
    from gaviko_amd import ops
    gen = torch.Generator().manual_seed(M)
    A = ops.act_zeros(M, K, torch.bfloat16, dev); A[:M] = torch.randn(M, K, generator=gen).bfloat16().to(dev)
    W = (torch.randn(N, K, generator=gen) / K ** 0.5).bfloat16().to(dev)
    bias = torch.randn(N, generator=gen).to(dev)
    res = torch.randn(M, N, generator=gen).to(dev)
    word = _seed(dev, 42)
    mask = torch.from_numpy(dropmask.rows_mask(9 + 42, M, N, P)).to(dev)
    acc = A[:M].float() @ W.float().t() + bias
    # BIAS_RES_F32: out = res + dropout(acc + bias)
    out = torch.empty(M, N, device=dev)
    ops.gemm_nt(A, W, M, out, epilogue=ops.EPI_BIAS_RES_F32, bias=bias, res=res, tile=tile, drop_p=P, seed=9, seed_ptr=word)
    assert (out - (res + acc * mask)).abs().max().item() < 2e-3
    assert torch.equal(out[mask == 0], res[mask == 0])                       # dropped elements are exactly the residual
    # BIAS_GELU_BF16: out0 = pre-activation (not dropped), out1 = dropout(GELU(pre))
    pre, act = ops.act_zeros(M, N, torch.bfloat16, dev), ops.act_zeros(M, N, torch.bfloat16, dev)
    ops.gemm_nt(A, W, M, pre, epilogue=ops.EPI_BIAS_GELU_BF16, out1=act, bias=bias, tile=tile, drop_p=P, seed=9, seed_ptr=word)
    assert (pre[:M].float() - acc).abs().max().item() < 3e-2
    want = torch.nn.functional.gelu(acc) * mask
    assert (act[:M].float() - want).abs().max().item() < 4e-2 and (act[:M][mask == 0] == 0).all()
    # GELU_BWD_BF16: out = (acc * mask) * GELU'(aux)
    aux = ops.act_zeros(M, N, torch.bfloat16, dev); aux[:M] = torch.randn(M, N, generator=gen).bfloat16().to(dev)
    dpre = ops.act_zeros(M, N, torch.bfloat16, dev)
    ops.gemm_nt(A, W, M, dpre, epilogue=ops.EPI_GELU_BWD_BF16, aux=aux, tile=tile, drop_p=P, seed=9, seed_ptr=word)
    a = aux[:M].float().requires_grad_(True)
    torch.nn.functional.gelu(a).sum().backward()
    want = (A[:M].float() @ W.float().t()) * mask * a.grad
    assert (dpre[:M].float() - want).abs().max().item() < 4e-2 and (dpre[:M][mask == 0] == 0).all()


@pytest.mark.parametrize("B,T,H", [(2, 1033, 3), (1, 200, 2)])
def test_attention_dropout_forward_backward(dev, B, T, H):
    from gaviko_amd import ops
    gen = torch.Generator().manual_seed(T)
    inner = H * 64
    qkv = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev)
    qkv[:B * T] = (torch.randn(B * T, 3 * inner, generator=gen) * 0.8).bfloat16().to(dev)
    out, lse = ops.act_zeros(B * T, inner, torch.bfloat16, dev), torch.empty(B * H * T, device=dev)
    word = _seed(dev, 2024)
    scale = 64 ** -0.5
    ops.attention_fwd(qkv, out, lse, B, T, H, scale, drop_p=P, seed=31, seed_ptr=word)
    mask = torch.from_numpy(dropmask.attn_mask(31 + 2024, B, H, T, P)).to(dev)
    assert abs((mask > 0).float().mean().item() - (1 - P)) < 0.01
    x = qkv[:B * T].float().view(B, T, 3, H, 64).requires_grad_(True)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    att = (q @ k.transpose(-1, -2) * scale).softmax(-1)
    ref = ((att * mask) @ v).permute(0, 2, 1, 3).reshape(B * T, inner)
    assert (out[:B * T].float() - ref).abs().max().item() < 3e-2
    want_lse = torch.logsumexp(q @ k.transpose(-1, -2) * scale, -1).reshape(-1)
    assert (lse - want_lse).abs().max().item() < 2e-3                        # the statistics are those of the undropped scores
    dout = ops.act_zeros(B * T, inner, torch.bfloat16, dev)
    dout[:B * T] = torch.randn(B * T, inner, generator=gen).bfloat16().to(dev)
    ref.backward(dout[:B * T].float())
    dqkv, delta = ops.act_zeros(B * T, 3 * inner, torch.bfloat16, dev), torch.empty(B * H * T, device=dev)
    ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, T, H, scale, drop_p=P, seed=31, seed_ptr=word)
    want = x.grad.reshape(B * T, 3 * inner)
    err = (dqkv[:B * T].float() - want).abs().max().item()
    assert err < 4e-2 * max(1.0, want.abs().max().item()), err
    # and p = 0 through the same entry points is the plain kernel, bit for bit
    o0, o1 = torch.zeros_like(out), torch.zeros_like(out)
    ops.attention_fwd(qkv, o0, lse, B, T, H, scale)
    ops.attention_fwd(qkv, o1, lse, B, T, H, scale, drop_p=0.0, seed=31, seed_ptr=word)
    assert torch.equal(o0, o1)
