"""Host-side launchers: torch tensors in, C-ABI calls out (include/gaviko_hip.h).

Every wrapper validates device / dtype / contiguity / row padding on the host before launching --
a hand-written kernel never sees a shape it was not built for.  All launches go to torch's current HIP
stream, so they are capturable into a HIP graph with torch.cuda.graph().
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L
from .lib import (EPI_BIAS_GELU_BF16, EPI_BIAS_RELU_BF16, EPI_BIAS_RES_F32, EPI_BIAS_RES_F32_BF16, EPI_GELU_BWD_BF16,  # noqa: F401
                  EPI_PATCH_F32, EPI_RELU_BWD_BF16, EPI_STORE_BF16, EPI_STORE_F32)

ROW_PAD = 128


def pad_rows(m: int) -> int:
    return (m + ROW_PAD - 1) // ROW_PAD * ROW_PAD


def act_zeros(m: int, c: int, dtype, device) -> torch.Tensor:
    """Activation matrix with rows padded to the MFMA panel height (padding rows stay finite)."""
    return torch.zeros((pad_rows(m), c), dtype=dtype, device=device)


def _chk(t, dtype, what, min_elems=0):
    if t is None:
        return
    if not t.is_cuda:
        raise L.GavikoHipError(f"{what}: tensor must live on the HIP device (no CPU path)")
    if t.dtype != dtype:
        raise L.GavikoHipError(f"{what}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise L.GavikoHipError(f"{what}: tensor must be contiguous")
    if t.numel() < min_elems:
        raise L.GavikoHipError(f"{what}: needs >= {min_elems} elements, has {t.numel()}")


def gemm_nt(a, w, M, out0, *, epilogue, out1=None, bias=None, res=None, aux=None, pos=None,
            lda=None, ldo=None, ldres=None, ldaux=None, rows_in=0, rows_out=0, row_off=0, tile=0, N=None, K=None,
            drop_p=0.0, seed=0, seed_ptr=None, scale_cols=0, col_scale=1.0, ln_mean=None, ln_rstd=None, ln_c1=None, stat_part=None, stat_pivot=None,
            m_panels=0, m_stride=0, splitk_ws=None, ksplit=0, aux_is_grad=0):
    """Y[M,N] = A[M,K] . W[N,K]^T with a fused epilogue.  The operand dtype picks the kernel: bf16 -> gvk_gemm_nt_bf16 (MFMA
    bf16, fp32 accumulate), fp32 -> gvk_gemm_nt_f32 (every 16-bit slot of the epilogue table then carries fp32).
    m_panels / m_stride (bf16): only the row tiles starting at rows 0, m_stride, 2 m_stride, ... are computed (gvk_gemm_desc)."""
    adt = a.dtype
    if adt not in (torch.bfloat16, torch.float32):
        raise L.GavikoHipError(f"gemm A: expected bf16 or fp32 operands, got {adt}")
    N = w.shape[0] if N is None else N
    K = w.shape[1] if K is None else K
    lda = a.shape[-1] if lda is None else lda
    ldw = w.shape[-1]
    ldo = N if ldo is None else ldo
    _chk(a, adt, "gemm A", pad_rows(M) * lda if lda == a.shape[-1] else 0)
    _chk(w, adt, "gemm W", N * ldw)
    out_dt = torch.float32 if epilogue in (EPI_BIAS_RES_F32, EPI_PATCH_F32, EPI_STORE_F32, EPI_BIAS_RES_F32_BF16) else adt
    _chk(out0, out_dt, "gemm out0")
    if out1 is not None:
        _chk(out1, torch.float32 if epilogue == EPI_PATCH_F32 else adt, "gemm out1")
    _chk(bias, torch.float32, "gemm bias", N)
    _chk(res, torch.float32, "gemm res")
    _chk(aux, adt, "gemm aux")
    _chk(pos, torch.float32, "gemm pos", rows_in * N)
    d = L.GemmDesc(L.ptr(a), L.ptr(w), L.ptr(out0), L.ptr(out1), L.ptr(bias), L.ptr(res), L.ptr(aux), L.ptr(pos),
                   L.ptr(seed_ptr) if (seed_ptr is not None and drop_p > 0) else None,
                   M, N, K, lda, ldw, ldo, (N if ldres is None else ldres), (N if ldaux is None else ldaux),
                   epilogue, rows_in, rows_out, row_off, tile, float(drop_p), int(seed), int(scale_cols), float(col_scale),
                   L.ptr(ln_mean), L.ptr(ln_rstd), L.ptr(ln_c1), L.ptr(stat_part), L.ptr(stat_pivot), int(m_panels), int(m_stride),
                   L.ptr(splitk_ws), 0 if splitk_ws is None else splitk_ws.numel() * splitk_ws.element_size(), int(ksplit), int(aux_is_grad))
    _chk(ln_mean, torch.float32, "gemm ln_mean", M)
    _chk(ln_rstd, torch.float32, "gemm ln_rstd", M)
    _chk(ln_c1, torch.float32, "gemm ln_c1", N)
    _chk(stat_part, torch.float32, "gemm stat_part", (N // 64) * M * 2)
    _chk(stat_pivot, torch.float32, "gemm stat_pivot", M)
    if adt == torch.float32:
        L.check(L.load().gvk_gemm_nt_f32(C.byref(d), L.stream_ptr()), "gvk_gemm_nt_f32")
    else:
        L.check(L.load().gvk_gemm_nt_bf16(C.byref(d), L.stream_ptr()), "gvk_gemm_nt_bf16")


def cast_bf16(x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    _chk(x, torch.float32, "cast in")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _chk(out, torch.bfloat16, "cast out", x.numel())
    L.check(L.load().gvk_cast_f32_bf16(L.ptr(x), L.ptr(out), x.numel(), L.stream_ptr()), "gvk_cast_f32_bf16")
    return out


def pack_split_bf16(a, dst, col0, rows, b=None, weight_side=False):
    """Split-bf16 packing (hi + lo) of a narrow fp32 operand into 3*ca + 2 spare K columns of a bf16 GEMM operand (gaviko_hip.h)."""
    _chk(a, torch.float32, "pack a", rows * a.shape[-1])
    _chk(dst, torch.bfloat16, "pack dst", rows * dst.shape[-1])
    _chk(b, torch.float32, "pack b", rows)
    L.check(L.load().gvk_pack_split_bf16(L.ptr(a), a.shape[-1], L.ptr(b), L.ptr(dst), dst.shape[-1], col0, rows, int(weight_side), L.stream_ptr()),
            "gvk_pack_split_bf16")


def layernorm_fwd_fix(x, gamma, beta, M, C_, *, y16, mean, rstd, enh, lat, wup, T, P, L_, eps=1e-5):
    """LayerNorm forward (bf16 output) that first applies the previous layer's GPA prompt fix to rows (m % T) < P of x, in place."""
    _chk(x, torch.float32, "ln_fix x", M * C_)
    _chk(gamma, torch.float32, "ln_fix gamma", C_)
    _chk(beta, torch.float32, "ln_fix beta", C_)
    _chk(y16, torch.bfloat16, "ln_fix y16", M * C_)
    _chk(mean, torch.float32, "ln_fix mean", M)
    _chk(rstd, torch.float32, "ln_fix rstd", M)
    _chk(enh, torch.float32, "ln_fix enh", (M // T) * P * L_)
    _chk(lat, torch.float32, "ln_fix lat", M * L_)
    _chk(wup, torch.float32, "ln_fix wup", C_ * L_)
    L.check(L.load().gvk_layernorm_fwd_fix(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(y16), L.ptr(mean), L.ptr(rstd), M, C_, eps, L.ptr(enh),
                                           L.ptr(lat), L.ptr(wup), T, P, L_, L.stream_ptr()), "gvk_layernorm_fwd_fix")


def prompt_up_fix(enh, lat, w, out, B, T, P, C_, L_):
    """out rows b*T + p (p < P) += (enh[b][p] - lat[b*T + p]) . w^T (gaviko_hip.h: gvk_prompt_up_fix)."""
    _chk(enh, torch.float32, "prompt_up_fix enh", B * P * L_)
    _chk(lat, torch.float32, "prompt_up_fix lat", B * T * L_)
    _chk(w, torch.float32, "prompt_up_fix w", C_ * L_)
    _chk(out, torch.float32, "prompt_up_fix out", B * T * C_)
    L.check(L.load().gvk_prompt_up_fix(L.ptr(enh), L.ptr(lat), L.ptr(w), L.ptr(out), B, T, P, C_, L_, L.stream_ptr()), "gvk_prompt_up_fix")


def prompt_up_fix_stats(enh, lat, w, out, out16, part, mean, rstd, B, T, P, C_, L_, eps=1e-5, pivot=None):
    """prompt_up_fix + the bf16 copy of the fixed rows + mean / rstd of EVERY row from the fc2 GEMM's per-row partials (the first LayerNorm of
    the next layer is folded into its qkv projection; gaviko_hip.h: gvk_prompt_up_fix_stats)."""
    _chk(enh, torch.float32, "prompt_up_fix enh", B * P * L_)
    _chk(lat, torch.float32, "prompt_up_fix lat", B * T * L_)
    _chk(w, torch.float32, "prompt_up_fix w", C_ * L_)
    _chk(out, torch.float32, "prompt_up_fix out", B * T * C_)
    _chk(out16, torch.bfloat16, "prompt_up_fix out16", B * T * C_)
    nparts = L.load().gvk_gemm_stat_parts(C_)
    _chk(part, torch.float32, "prompt_up_fix part", nparts * B * T * 2)
    _chk(mean, torch.float32, "prompt_up_fix mean", B * T)
    _chk(rstd, torch.float32, "prompt_up_fix rstd", B * T)
    _chk(pivot, torch.float32, "prompt_up_fix pivot", B * T)
    L.check(L.load().gvk_prompt_up_fix_stats(L.ptr(enh), L.ptr(lat), L.ptr(w), L.ptr(out), L.ptr(out16), L.ptr(part), nparts, L.ptr(pivot), L.ptr(mean), L.ptr(rstd),
                                             B, T, P, C_, L_, eps, L.stream_ptr()), "gvk_prompt_up_fix_stats")


def copy_(dst: torch.Tensor, src: torch.Tensor) -> None:
    """Stream-ordered device copy of src into dst (same dtype, contiguous; dst may be larger)."""
    if dst.dtype != src.dtype or not dst.is_contiguous() or not src.is_contiguous() or dst.numel() < src.numel():
        raise L.GavikoHipError("copy_: need contiguous tensors of one dtype with dst at least as large as src")
    L.check(L.load().gvk_copy_async(L.ptr(dst), L.ptr(src), src.numel() * src.element_size(), L.stream_ptr()), "gvk_copy_async")


def to_operand(x: torch.Tensor, out: torch.Tensor = None, dtype=torch.bfloat16) -> torch.Tensor:
    """GEMM-operand form of an fp32 matrix: a bf16 copy (MFMA path) or, on the fp32 path, the matrix itself / an fp32 copy."""
    if dtype == torch.bfloat16:
        return cast_bf16(x, out)
    if out is None:
        return x
    copy_(out, x)
    return out


def transpose_operand(x: torch.Tensor, out: torch.Tensor = None, dtype=torch.bfloat16) -> torch.Tensor:
    """x f32 [rows, cols] -> [cols, rows] in the operand dtype."""
    if dtype == torch.bfloat16:
        return transpose_cast_bf16(x, out)
    _chk(x, torch.float32, "transpose in")
    rows, cols = x.shape
    if out is None:
        out = torch.empty((cols, rows), dtype=torch.float32, device=x.device)
    _chk(out, torch.float32, "transpose out", rows * cols)
    L.check(L.load().gvk_transpose_f32(L.ptr(x), L.ptr(out), rows, cols, L.stream_ptr()), "gvk_transpose_f32")
    return out


def transpose_any(x: torch.Tensor, out: torch.Tensor, rows: int, cols: int) -> None:
    """out [cols][rows] = x[rows][cols]^T within one dtype (bf16 or fp32): operand transposes of the wgrad GEMMs."""
    if x.dtype != out.dtype or x.dtype not in (torch.bfloat16, torch.float32):
        raise L.GavikoHipError("transpose_any: x and out must both be bf16 or both fp32")
    _chk(x, x.dtype, "transpose_any x", rows * cols)
    _chk(out, x.dtype, "transpose_any out", rows * cols)
    fn = L.load().gvk_transpose_f32 if x.dtype == torch.float32 else L.load().gvk_transpose_bf16
    L.check(fn(L.ptr(x), L.ptr(out), rows, cols, L.stream_ptr()), "gvk_transpose")


def colsum_any(x, out, ones, zeros, junk, scratch, M, N, *, ld=None, rows_in=0, rows_out=0, row_off=0):
    """out[n] = sum_m x[m][n] for a bf16 / fp32 matrix (optionally through the PATCH row mapping): the column-sum half of gvk_ssf_colgrad."""
    ssf_colgrad(x, x, ones, zeros, junk, out, scratch, M, N, ld_dy=ld, ld_y=ld, rows_in=rows_in, rows_out=rows_out, row_off=row_off)


def memset_zero(t: torch.Tensor) -> None:
    """Stream-ordered zero fill (hipMemsetAsync through the library so that it is part of a recorded launch plan)."""
    if not t.is_contiguous():
        raise L.GavikoHipError("memset_zero: tensor must be contiguous")
    L.check(L.load().gvk_memset_async(L.ptr(t), 0, t.numel() * t.element_size(), L.stream_ptr()), "gvk_memset_async")


def seed_advance(seed: torch.Tensor, inc: int) -> None:
    if seed.dtype != torch.int64 or seed.numel() != 1:
        raise L.GavikoHipError("seed_advance: the dropout epoch is one int64 device word")
    L.check(L.load().gvk_seed_advance(L.ptr(seed), inc, L.stream_ptr()), "gvk_seed_advance")


def scale_(t: torch.Tensor, alpha: float) -> None:
    _chk(t, torch.float32, "scale_")
    L.check(L.load().gvk_scale_f32(L.ptr(t), alpha, t.numel(), L.stream_ptr()), "gvk_scale_f32")


def transpose_cast_bf16(x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """x f32 [rows, cols] -> bf16 [cols, rows]."""
    _chk(x, torch.float32, "transpose_cast in")
    rows, cols = x.shape
    if out is None:
        out = torch.empty((cols, rows), dtype=torch.bfloat16, device=x.device)
    _chk(out, torch.bfloat16, "transpose_cast out", rows * cols)
    L.check(L.load().gvk_transpose_cast_f32_bf16(L.ptr(x), L.ptr(out), rows, cols, L.stream_ptr()), "gvk_transpose_cast_f32_bf16")
    return out


def patchify(img: torch.Tensor, out: torch.Tensor, patch) -> None:
    _chk(img, torch.float32, "patchify img")
    B, ch, D, H, W = img.shape
    if ch != 1:
        raise L.GavikoHipError("patchify: single-channel MRI volumes only (channels=1)")
    pd, ph, pw = patch
    n = (D // pd) * (H // ph) * (W // pw)
    if out.dtype == torch.float32:
        _chk(out, torch.float32, "patchify out", B * n * pd * ph * pw)
        L.check(L.load().gvk_patchify_f32(L.ptr(img), L.ptr(out), B, D, H, W, pd, ph, pw, L.stream_ptr()), "gvk_patchify_f32")
        return
    _chk(out, torch.bfloat16, "patchify out", B * n * pd * ph * pw)
    L.check(L.load().gvk_patchify_bf16(L.ptr(img), L.ptr(out), B, D, H, W, pd, ph, pw, L.stream_ptr()), "gvk_patchify_bf16")


def layernorm_fwd(x, gamma, beta, M, C_, *, y16=None, y32=None, mean=None, rstd=None, eps=1e-5):
    if y16 is not None and y16.dtype == torch.float32:       # fp32 compute path: the "operand" output is fp32
        y16, y32 = None, y16
    _chk(x, torch.float32, "ln x", M * C_)
    _chk(gamma, torch.float32, "ln gamma", C_)
    _chk(beta, torch.float32, "ln beta", C_)
    _chk(y16, torch.bfloat16, "ln y16", M * C_)
    _chk(y32, torch.float32, "ln y32", M * C_)
    _chk(mean, torch.float32, "ln mean", M)
    _chk(rstd, torch.float32, "ln rstd", M)
    L.check(L.load().gvk_layernorm_fwd(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(y16), L.ptr(y32), L.ptr(mean), L.ptr(rstd),
                                       M, C_, eps, L.stream_ptr()), "gvk_layernorm_fwd")


ROWPROJ_L = 20      # default latent width (configs/gaviko.yaml prompt_latent_dim)


def side_tile_supported(L_: int, C_: int) -> bool:
    """Shapes the 16-row-tile fp32-MFMA projections (csrc/sidepass.hip) cover -- and with them the second projection fused into
    gvk_skinny_up (w2 / z2 / y2) and the chained layer-boundary forms.  Mirrors the C side exactly (sidepass.hip: kSL = 20,
    groups_per_wave(C) != 0 only for C in {192, 768, 1024}); any other latent width or channel count runs the generic
    row-per-wave / MFMA-tile kernels of rowwise.hip / skinny.hip without the fusions."""
    return str(L.diag_env("GAVIKO_HIP_SIDE", "1"))[:1] != "0" and L_ == 20 and C_ in (192, 768, 1024)


def rowproj_supported(L_: int, C_: int) -> bool:
    """Shapes the fused LayerNorm+projection kernels (row-per-wave form) cover."""
    return L_ in (4, 8, 16, 20) and C_ % 4 == 0 and 128 <= C_ <= 1024


def _rowproj(M, C_, w, y, bias, z, L_, w_layout, act, y_split=None, col_split=0):
    _chk(w, torch.float32, "rowproj w", L_ * C_)
    _chk(y, torch.float32, "rowproj y", M * L_)
    _chk(bias, torch.float32, "rowproj bias", L_)
    _chk(z, torch.float32, "rowproj z", M * L_)
    _chk(y_split, torch.bfloat16, "rowproj y_split", 0 if y_split is None else M * y_split.shape[-1])
    return L.RowProjDesc(w=L.ptr(w), bias=L.ptr(bias), y=L.ptr(y), z=L.ptr(z), y_split=L.ptr(y_split), L=L_, w_layout=w_layout, act=act,
                         ld_split=0 if y_split is None else y_split.shape[-1], col_split=col_split)


def layernorm_fwd_proj(x, gamma, beta, M, C_, *, y16, mean=None, rstd=None, eps=1e-5, w, y, bias=None, z=None, L_=ROWPROJ_L, w_layout=0, act=0,
                       y_split=None, col_split=0):
    """LayerNorm forward + rank-L projection of the raw input rows (one pass over x)."""
    _chk(x, torch.float32, "ln x", M * C_)
    _chk(gamma, torch.float32, "ln gamma", C_)
    _chk(beta, torch.float32, "ln beta", C_)
    _chk(y16, torch.bfloat16, "ln y16", M * C_)
    _chk(mean, torch.float32, "ln mean", M)
    _chk(rstd, torch.float32, "ln rstd", M)
    d = _rowproj(M, C_, w, y, bias, z, L_, w_layout, act, y_split, col_split)
    L.check(L.load().gvk_layernorm_fwd_proj(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(y16), L.ptr(mean), L.ptr(rstd), M, C_, eps,
                                            C.byref(d), L.stream_ptr()), "gvk_layernorm_fwd_proj")


def layernorm_bwd_proj(dy, x, mean, rstd, gamma, M, C_, *, dx, dres=None, dx16=None, w, y, L_=ROWPROJ_L, w_layout=1):
    """LayerNorm backward + rank-L projection of the output rows dx (one pass)."""
    for t, n in ((dy, "dy"), (x, "x"), (dx, "dx")):
        _chk(t, torch.float32, "ln_bwd " + n, M * C_)
    _chk(dres, torch.float32, "ln_bwd dres", M * C_)
    _chk(dx16, torch.bfloat16, "ln_bwd dx16", M * C_)
    _chk(mean, torch.float32, "ln_bwd mean", M)
    _chk(rstd, torch.float32, "ln_bwd rstd", M)
    _chk(gamma, torch.float32, "ln_bwd gamma", C_)
    d = _rowproj(M, C_, w, y, None, None, L_, w_layout, 0)
    L.check(L.load().gvk_layernorm_bwd_proj(L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(dres), L.ptr(dx),
                                            L.ptr(dx16), M, C_, C.byref(d), L.stream_ptr()), "gvk_layernorm_bwd_proj")


def layernorm_bwd_up(dy, x, mean, rstd, gamma, M, C_, *, dx, dres, dx16, lat, w, L_, w_layout):
    """dx = dres + LN'(dy) + lat . W^T (+ bf16 copy): the MLP block's LayerNorm backward and GPA's dG1 += dzx . W_d in one pass."""
    for t, n in ((dy, "dy"), (x, "x"), (dx, "dx"), (dres, "dres")):
        _chk(t, torch.float32, "ln_bwd_up " + n, M * C_)
    _chk(dx16, torch.bfloat16, "ln_bwd_up dx16", M * C_)
    _chk(mean, torch.float32, "ln_bwd_up mean", M)
    _chk(rstd, torch.float32, "ln_bwd_up rstd", M)
    _chk(gamma, torch.float32, "ln_bwd_up gamma", C_)
    _chk(lat, torch.float32, "ln_bwd_up lat", M * L_)
    _chk(w, torch.float32, "ln_bwd_up w", L_ * C_)
    L.check(L.load().gvk_layernorm_bwd_up(L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(dres), L.ptr(dx), L.ptr(dx16),
                                          L.ptr(lat), L.ptr(w), w_layout, M, C_, L_, L.stream_ptr()), "gvk_layernorm_bwd_up")


def layernorm_bwd(dy, x, mean, rstd, gamma, M, C_, *, dx, dres=None, dx16=None):
    if dx16 is not None and dx16.dtype == torch.float32:     # fp32 compute path: the operand copy is a plain copy of dx
        layernorm_bwd(dy, x, mean, rstd, gamma, M, C_, dx=dx, dres=dres)
        copy_(dx16, dx)
        return
    for t, n in ((dy, "dy"), (x, "x"), (dx, "dx")):
        _chk(t, torch.float32, "ln_bwd " + n, M * C_)
    _chk(dres, torch.float32, "ln_bwd dres", M * C_)
    _chk(dx16, torch.bfloat16, "ln_bwd dx16", M * C_)
    _chk(mean, torch.float32, "ln_bwd mean", M)
    _chk(rstd, torch.float32, "ln_bwd rstd", M)
    _chk(gamma, torch.float32, "ln_bwd gamma", C_)
    L.check(L.load().gvk_layernorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(dres), L.ptr(dx),
                                       L.ptr(dx16), M, C_, L.stream_ptr()), "gvk_layernorm_bwd")


def layernorm_bwd_dy16(dy16, x, mean, rstd, gamma, M, C_, *, dx, dres=None, dx16=None, rows=None, proj=None):
    """LayerNorm backward with the output gradient in bf16 (gvk_layernorm_bwd_dy16).  rows = (groups, rows_per_group, group_stride) restricts it
    to the leading rows of every group; proj = dict(w=, y=, L_=, w_layout=) adds the rank-L projection of dx (as layernorm_bwd_proj)."""
    _chk(dy16, torch.bfloat16, "ln_bwd_dy16 dy", M * C_)
    for t, n in ((x, "x"), (dx, "dx")):
        _chk(t, torch.float32, "ln_bwd_dy16 " + n, M * C_)
    _chk(dres, torch.float32, "ln_bwd_dy16 dres", M * C_)
    _chk(dx16, torch.bfloat16, "ln_bwd_dy16 dx16", M * C_)
    _chk(mean, torch.float32, "ln_bwd_dy16 mean", M)
    _chk(rstd, torch.float32, "ln_bwd_dy16 rstd", M)
    _chk(gamma, torch.float32, "ln_bwd_dy16 gamma", C_)
    g, rpg, gs = rows if rows is not None else (0, 0, 0)
    pj = None
    if proj is not None:
        pj = _rowproj(M, C_, proj["w"], proj["y"], None, None, proj.get("L_", ROWPROJ_L), proj.get("w_layout", 1), 0)
    d = L.LnBwdDy16Desc(L.ptr(dy16), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(dres), L.ptr(dx), L.ptr(dx16),
                        C.cast(C.pointer(pj), C.c_void_p) if pj is not None else None, M, C_, int(g), int(rpg), int(gs))
    L.check(L.load().gvk_layernorm_bwd_dy16(C.byref(d), L.stream_ptr()), "gvk_layernorm_bwd_dy16")


def layernorm_bwd_rows(dy, x, mean, rstd, gamma, groups, rows_per_group, group_stride, C_, *, dx, dres=None, dx16=None):
    """layernorm_bwd for the first `rows_per_group` rows of every group of `group_stride` rows (every sample's leading tokens); the other
    rows of dx / dx16 are left as they are."""
    M = (groups - 1) * group_stride + rows_per_group
    for t, n in ((dy, "dy"), (x, "x"), (dx, "dx")):
        _chk(t, torch.float32, "ln_bwd_rows " + n, M * C_)
    _chk(dres, torch.float32, "ln_bwd_rows dres", M * C_)
    _chk(dx16, torch.bfloat16, "ln_bwd_rows dx16", M * C_)
    _chk(mean, torch.float32, "ln_bwd_rows mean", M)
    _chk(rstd, torch.float32, "ln_bwd_rows rstd", M)
    _chk(gamma, torch.float32, "ln_bwd_rows gamma", C_)
    L.check(L.load().gvk_layernorm_bwd_rows(L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(dres), L.ptr(dx), L.ptr(dx16),
                                            groups, rows_per_group, group_stride, C_, L.stream_ptr()), "gvk_layernorm_bwd_rows")


def layernorm_bwd_affine(dy, x, mean, rstd, dgamma, dbeta, scratch, M, C_, accumulate=False):
    for t, n in ((dy, "dy"), (x, "x")):
        _chk(t, torch.float32, "ln_affine " + n, M * C_)
    _chk(dgamma, torch.float32, "ln_affine dgamma", C_)
    _chk(dbeta, torch.float32, "ln_affine dbeta", C_)
    _chk(scratch, torch.float32, "ln_affine scratch", 128 * C_)
    L.check(L.load().gvk_layernorm_bwd_affine(L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(dgamma), L.ptr(dbeta),
                                              L.ptr(scratch), M, C_, int(accumulate), L.stream_ptr()), "gvk_layernorm_bwd_affine")


LOG2E = 1.4426950408889634


def qkv_prescale(qkv, rows, H, scale):
    """In place: q block of a raw bf16 to_qkv output -> q * scale * log2(e), the form gvk_attention_*_bf16 take (the engine gets it from the
    qkv projection's epilogue instead: gemm_nt(scale_cols=H*64, col_scale=scale*LOG2E))."""
    _chk(qkv, torch.bfloat16, "qkv_prescale qkv", rows * 3 * H * 64)
    L.check(L.load().gvk_qkv_prescale_bf16(L.ptr(qkv), rows, H, qkv.shape[-1], scale, L.stream_ptr()), "gvk_qkv_prescale_bf16")


def _prescaled_copy(qkv, rows, H, scale):
    c = qkv.clone()
    qkv_prescale(c, rows, H, scale)
    return c


def attention_fwd(qkv, out, lse, B, T, H, scale, drop_p=0.0, seed=0, seed_ptr=None, q_prescaled=False):
    """qkv bf16 [pad(B*T), 3*H*64] -> out bf16 [pad(B*T), H*64], lse f32 [B,H,T].  drop_p > 0: dropout on the probabilities (bf16 path).
    bf16 path: the kernels take the q block pre-scaled by scale*log2(e) (include/gaviko_hip.h); q_prescaled=False (tests, tools) makes a
    scaled copy of a raw qkv first -- an allocation and one more rounding of q, never on the engine's path."""
    inner = H * 64
    if qkv.dtype == torch.float32:
        _chk(qkv, torch.float32, "attn qkv", B * T * 3 * inner)
        _chk(out, torch.float32, "attn out", B * T * inner)
        _chk(lse, torch.float32, "attn lse", B * H * T)
        L.check(L.load().gvk_attention_fwd_f32_dropout(L.ptr(qkv), L.ptr(out), L.ptr(lse), B, T, H, 3 * inner, inner, scale, float(drop_p), int(seed),
                                                       L.ptr(seed_ptr) if drop_p > 0 else None, L.stream_ptr()), "gvk_attention_fwd_f32")
        return
    _chk(qkv, torch.bfloat16, "attn qkv", pad_rows(B * T) * 3 * inner)
    _chk(out, torch.bfloat16, "attn out", B * T * inner)
    _chk(lse, torch.float32, "attn lse", B * H * T)
    if not q_prescaled:
        qkv = _prescaled_copy(qkv, B * T, H, scale)
    if drop_p > 0:
        L.check(L.load().gvk_attention_fwd_bf16_dropout(L.ptr(qkv), L.ptr(out), L.ptr(lse), B, T, H, 3 * inner, inner, scale, float(drop_p),
                                                        int(seed), L.ptr(seed_ptr), L.stream_ptr()), "gvk_attention_fwd_bf16_dropout")
        return
    L.check(L.load().gvk_attention_fwd_bf16(L.ptr(qkv), L.ptr(out), L.ptr(lse), B, T, H, 3 * inner, inner, scale, L.stream_ptr()),
            "gvk_attention_fwd_bf16")


def _desc(cls, what, **kw):
    """Fill a descriptor struct: tensors -> device pointers (validated fp32, contiguous, on device), None -> NULL."""
    d = cls()
    for name, ctype in cls._fields_:
        v = kw.pop(name, None)
        if ctype is C.c_void_p:
            if v is not None:
                _chk(v, torch.int64 if name == "seed_ptr" else torch.bfloat16 if name in ("out_bf16", "enh16") else torch.float32, f"{what}.{name}")
            setattr(d, name, L.ptr(v))
        elif v is not None:
            setattr(d, name, v)
    if kw:
        raise TypeError(f"{what}: unknown fields {sorted(kw)}")
    return d


def skinny_down(**kw):
    d = _desc(L.SkinnyDownDesc, "skinny_down", **kw)
    L.check(L.load().gvk_skinny_down(C.byref(d), L.stream_ptr()), "gvk_skinny_down")


def skinny_up(**kw):
    d = _desc(L.SkinnyUpDesc, "skinny_up", **kw)
    L.check(L.load().gvk_skinny_up(C.byref(d), L.stream_ptr()), "gvk_skinny_up")


def outer_reduce(**kw):
    d = _desc(L.OuterDesc, "outer_reduce", **kw)
    L.check(L.load().gvk_outer_reduce(C.byref(d), L.stream_ptr()), "gvk_outer_reduce")


OUTER_MAX_ROWS = 10240        # rows one gvk_outer_reduce launch covers (64 slabs x 160 rows), second source included


def outer_scratch_elems(Lat, C_):
    return 128 * (Lat + 1) * C_


def _reduce_jobs(jobs, what):
    arr = (L.ReduceJob * max(1, len(jobs)))()
    for k, job in enumerate(jobs):
        a, b, out, acc = job[:4]
        a2 = job[4] if len(job) > 4 else None
        for t, n in ((a, "a"), (b, "b"), (out, "out"), (a2, "a2")):
            _chk(t, torch.float32, f"{what} {n}")
        M, J = a.shape[0], a.numel() // a.shape[0]
        Lb = 0 if b is None else b.numel() // b.shape[0]
        if b is not None and b.shape[0] != M:
            raise L.GavikoHipError(f"{what}: a and b must have the same number of rows")
        if out.numel() < (J * Lb if b is not None else J):
            raise L.GavikoHipError(f"{what}: out too small")
        if a2 is not None and (b is not None or a2.numel() // a2.shape[0] != J):
            raise L.GavikoHipError(f"{what}: a2 goes with column sums of the same width only")
        arr[k] = L.ReduceJob(L.ptr(a), L.ptr(b), L.ptr(out), L.ptr(a2), M, J, Lb, int(bool(acc)), 0 if a2 is None else a2.shape[0])
    return arr


def param_grads_supported(Lat, C_):
    return C_ % 4 == 0 and 0 < Lat <= 28 and Lat % 4 == 0


PGRAD_TICKETS = 256           # ticket words one gvk_param_grads call may need: column tiles of the outer jobs + 64-output chunks of the small jobs (56 for the GPA gates)


def param_grads(outer, small, scratch, tickets, C_, Lat, seed_ptr=None):
    """One launch for every parameter gradient of a rank-L side-path module of one layer (gaviko_hip.h: gvk_param_grads).
    outer: list of dicts with the gvk_pgrad_outer fields (tensors or None); small: list of (a, b or None, out, accumulate[, a2])."""
    arr = (L.PgradOuter * max(1, len(outer)))()
    fields = [f for f, _ in L.PgradOuter._fields_]
    for k, o in enumerate(outer):
        bad = set(o) - set(fields)
        if bad:
            raise L.GavikoHipError(f"param_grads: unknown field(s) {sorted(bad)}")
        vals = {}
        for f, ct in L.PgradOuter._fields_:
            v = o.get(f)
            if ct is C.c_void_p:
                _chk(v, torch.float32, f"param_grads outer[{k}].{f}")
                vals[f] = L.ptr(v)
            elif ct is C.c_float:
                vals[f] = float(v or 0.0)
            else:
                vals[f] = int(v or 0)
        M, M2, Cj = vals["M"], vals["M2"], vals["C"] or C_
        for f, n in (("narrow", M * Lat), ("wide", M * Cj), ("narrow2", M2 * Lat), ("wide2", M2 * Cj), ("mean", M), ("rstd", M), ("colsum", Cj),
                     ("out", Lat * Cj), ("aff_w", Lat * Cj), ("aff_gamma", Cj), ("aff_beta", Cj), ("aff_dgamma", Cj), ("aff_dbeta", Cj), ("aff_dbias", Lat)):
            if o.get(f) is not None and o[f].numel() < n:
                raise L.GavikoHipError(f"param_grads outer[{k}].{f}: needs >= {n} elements, has {o[f].numel()}")
        arr[k] = L.PgradOuter(**vals)
    jobs = _reduce_jobs(small, "param_grads small")
    _chk(scratch, torch.float32, "param_grads scratch")
    if tickets is None or tickets.dtype != torch.int32 or not tickets.is_cuda:
        raise L.GavikoHipError("param_grads: tickets must be an int32 tensor on the HIP device (zero at allocation)")
    L.check(L.load().gvk_param_grads(arr, len(outer), jobs, len(small), L.ptr(scratch), scratch.numel(), L.ptr(tickets), tickets.numel(),
                                     L.ptr(seed_ptr), C_, Lat, L.stream_ptr()), "gvk_param_grads")


def param_grads_scratch_elems(Lat, col_tiles, small_outputs):
    """Upper bound of the scratch floats of one gvk_param_grads call: col_tiles = 64-column tiles of each outer job (the kernel deals the
    jobs' rows to ~240 workgroups, at most 64 row ranges per column tile), small_outputs = outputs of each small job."""
    slabs = min(240 + sum(col_tiles), 64 * sum(col_tiles))
    return slabs * (Lat + 2) * 64 + sum(((o + 63) // 64) * 32 * 64 for o in small_outputs)


def small_wgrad(a, b, out, scratch, M, J, Lb, accumulate=False):
    for t, n in ((a, "a"), (b, "b"), (out, "out"), (scratch, "scratch")):
        _chk(t, torch.float32, "small_wgrad " + n)
    if scratch.numel() < 64 * J * Lb or a.numel() < M * J or b.numel() < M * Lb or out.numel() < J * Lb:
        raise L.GavikoHipError("small_wgrad: buffer too small")
    L.check(L.load().gvk_small_wgrad(L.ptr(a), L.ptr(b), L.ptr(out), L.ptr(scratch), M, J, Lb, int(accumulate), L.stream_ptr()), "gvk_small_wgrad")


def colsum(x, out, scratch, M, C_, accumulate=False):
    for t, n in ((x, "x"), (out, "out"), (scratch, "scratch")):
        _chk(t, torch.float32, "colsum " + n)
    if scratch.numel() < 64 * C_ or x.numel() < M * C_ or out.numel() < C_:
        raise L.GavikoHipError("colsum: buffer too small")
    L.check(L.load().gvk_colsum(L.ptr(x), L.ptr(out), L.ptr(scratch), M, C_, int(accumulate), L.stream_ptr()), "gvk_colsum")


def window_attn_fwd(**kw):
    d = _desc(L.WindowAttnDesc, "window_attn", **kw)
    L.check(L.load().gvk_window_attn_fwd(C.byref(d), L.stream_ptr()), "gvk_window_attn_fwd")


def window_attn_bwd(**kw):
    d = _desc(L.WindowAttnDesc, "window_attn", **kw)
    L.check(L.load().gvk_window_attn_bwd(C.byref(d), L.stream_ptr()), "gvk_window_attn_bwd")


def gpa_fwd(**kw):
    d = _desc(L.GpaDesc, "gpa", **kw)
    L.check(L.load().gvk_gpa_fwd(C.byref(d), L.stream_ptr()), "gvk_gpa_fwd")


def gpa_bwd(**kw):
    d = _desc(L.GpaDesc, "gpa", **kw)
    L.check(L.load().gvk_gpa_bwd(C.byref(d), L.stream_ptr()), "gvk_gpa_bwd")


def gpa_gate_param_count(Lat, P):
    return L.load().gvk_gpa_gate_param_count(Lat, P)


def rows_broadcast(out, src, add, B, T, row_off, R, C_):
    _chk(out, torch.float32, "rows_broadcast out", B * T * C_)
    _chk(src, torch.float32, "rows_broadcast src", R * C_)
    _chk(add, torch.float32, "rows_broadcast add", R * C_)
    L.check(L.load().gvk_rows_broadcast(L.ptr(out), L.ptr(src), L.ptr(add), B, T, row_off, R, C_, L.stream_ptr()), "gvk_rows_broadcast")


def rows_batch_sum(dg, out, out2, B, T, row_off, R, C_, accumulate=False):
    _chk(dg, torch.float32, "rows_batch_sum dg", B * T * C_)
    _chk(out, torch.float32, "rows_batch_sum out", R * C_)
    _chk(out2, torch.float32, "rows_batch_sum out2", R * C_)
    L.check(L.load().gvk_rows_batch_sum(L.ptr(dg), L.ptr(out), L.ptr(out2), B, T, row_off, R, C_, int(accumulate), L.stream_ptr()),
            "gvk_rows_batch_sum")


def head_fwd(**kw):
    d = _desc(L.HeadDesc, "head", **kw)
    L.check(L.load().gvk_head_fwd(C.byref(d), L.stream_ptr()), "gvk_head_fwd")


def head_bwd(**kw):
    d = _desc(L.HeadDesc, "head", **kw)
    L.check(L.load().gvk_head_bwd(C.byref(d), L.stream_ptr()), "gvk_head_bwd")


def attention_bwd_workspace(B, T, H, device):
    """Zeroed workspace of the one-pass bf16 backward (gvk_attention_bwd_bf16_fused): progress words + the running dQ sums of the
    ordered hand-off.  One per stream that issues the call (launches sharing it must be ordered)."""
    n = int(L.load().gvk_attention_bwd_ws_bytes(B, T, H))
    return torch.zeros((n + 3) // 4, dtype=torch.int32, device=device)


def attention_bwd_timeouts(ws) -> int:
    """Hand-off waits of the one-pass backward that ran into their bound since the workspace was made (0 unless a launch was broken)."""
    return int(ws[int(L.load().gvk_attention_bwd_status_offset(ws.numel() * 4)) // 4].item())


def attention_bwd(qkv, out, dout, lse, delta, dqkv, B, T, H, scale, drop_p=0.0, seed=0, seed_ptr=None, q_prescaled=False, ws=None, need_rows=None):
    """ws (attention_bwd_workspace): run the one-pass kernel (five products, ordered dQ hand-off); without it, or with attention
    dropout, the two-pass kernels.  need_rows (bf16, no dropout): gradients of the first need_rows tokens of every sample only."""
    inner = H * 64
    if qkv.dtype == torch.float32:
        for t, n, k in ((qkv, "qkv", 3), (out, "out", 1), (dout, "dout", 1), (dqkv, "dqkv", 3)):
            _chk(t, torch.float32, "attn_bwd " + n, B * T * k * inner)
        _chk(lse, torch.float32, "attn_bwd lse", B * H * T)
        _chk(delta, torch.float32, "attn_bwd delta", B * H * T)
        L.check(L.load().gvk_attention_bwd_f32_dropout(L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(lse), L.ptr(delta), L.ptr(dqkv), B, T, H,
                                                       3 * inner, inner, scale, float(drop_p), int(seed), L.ptr(seed_ptr) if drop_p > 0 else None,
                                                       L.stream_ptr()), "gvk_attention_bwd_f32")
        return
    _chk(qkv, torch.bfloat16, "attn_bwd qkv", pad_rows(B * T) * 3 * inner)
    _chk(out, torch.bfloat16, "attn_bwd out", pad_rows(B * T) * inner)
    _chk(dout, torch.bfloat16, "attn_bwd dout", pad_rows(B * T) * inner)
    _chk(dqkv, torch.bfloat16, "attn_bwd dqkv", B * T * 3 * inner)
    _chk(lse, torch.float32, "attn_bwd lse", B * H * T)
    _chk(delta, torch.float32, "attn_bwd delta", B * H * T)
    if not q_prescaled:                                      # see attention_fwd
        qkv = _prescaled_copy(qkv, B * T, H, scale)
    if drop_p > 0:
        L.check(L.load().gvk_attention_bwd_bf16_dropout(L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(lse), L.ptr(delta), L.ptr(dqkv), B, T, H,
                                                        3 * inner, inner, scale, float(drop_p), int(seed), L.ptr(seed_ptr), L.stream_ptr()),
                "gvk_attention_bwd_bf16_dropout")
        return
    if need_rows is not None:
        L.check(L.load().gvk_attention_bwd_bf16_rows(L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(lse), L.ptr(delta), L.ptr(dqkv), B, T, H,
                                                     3 * inner, inner, scale, int(need_rows), L.stream_ptr()), "gvk_attention_bwd_bf16_rows")
        return
    if ws is not None:
        _chk(ws, torch.int32, "attn_bwd ws", (int(L.load().gvk_attention_bwd_ws_bytes(B, T, H)) + 3) // 4)
        L.check(L.load().gvk_attention_bwd_bf16_fused(L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(lse), L.ptr(delta), L.ptr(dqkv), L.ptr(ws),
                                                      ws.numel() * 4, B, T, H, 3 * inner, inner, scale, L.stream_ptr()), "gvk_attention_bwd_bf16_fused")
        return
    L.check(L.load().gvk_attention_bwd_bf16(L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(lse), L.ptr(delta), L.ptr(dqkv), B, T, H,
                                            3 * inner, inner, scale, L.stream_ptr()), "gvk_attention_bwd_bf16")


def small_linear_fwd(x, w, b, out, R, K, C_):
    for t, n in ((x, "x"), (w, "w"), (b, "b"), (out, "out")):
        _chk(t, torch.float32, "small_linear " + n)
    if x.numel() < R * K or w.numel() < C_ * K or out.numel() < R * C_:
        raise L.GavikoHipError("small_linear_fwd: buffer too small")
    L.check(L.load().gvk_small_linear_fwd(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(out), R, K, C_, L.stream_ptr()), "gvk_small_linear_fwd")


def small_linear_bwd(x, w, dout, dw, db, dx, R, K, C_, accumulate=False):
    for t, n in ((x, "x"), (w, "w"), (dout, "dout"), (dw, "dw"), (db, "db"), (dx, "dx")):
        _chk(t, torch.float32, "small_linear_bwd " + n)
    L.check(L.load().gvk_small_linear_bwd(L.ptr(x), L.ptr(w), L.ptr(dout), L.ptr(dw), L.ptr(db), L.ptr(dx), R, K, C_, int(accumulate),
                                          L.stream_ptr()), "gvk_small_linear_bwd")


def vpt_repack_fwd(inp, prompt, out, B, Tin, Tout, P, skip, C_):
    _chk(inp, torch.float32, "vpt_repack in", B * Tin * C_)
    _chk(prompt, torch.float32, "vpt_repack prompt", P * C_)
    _chk(out, torch.float32, "vpt_repack out", B * Tout * C_)
    L.check(L.load().gvk_vpt_repack_fwd(L.ptr(inp), L.ptr(prompt), L.ptr(out), B, Tin, Tout, P, skip, C_, L.stream_ptr()), "gvk_vpt_repack_fwd")


def vpt_repack_bwd(dout, din, B, Tin, Tout, P, skip, C_):
    _chk(dout, torch.float32, "vpt_repack_bwd dout", B * Tout * C_)
    _chk(din, torch.float32, "vpt_repack_bwd din", B * Tin * C_)
    L.check(L.load().gvk_vpt_repack_bwd(L.ptr(dout), L.ptr(din), B, Tin, Tout, P, skip, C_, L.stream_ptr()), "gvk_vpt_repack_bwd")


def cast_bf16_f32_strided(inp, out, M, C_, ld_in, col0=0):
    """f32 out[M][C] = inp[M][ld_in] columns col0 .. col0+C (inp bf16, or fp32 on the fp32 compute path)."""
    if inp.dtype == torch.float32:
        _chk(inp, torch.float32, "copy_strided in", M * ld_in)
        _chk(out, torch.float32, "copy_strided out", M * C_)
        L.check(L.load().gvk_copy_f32_strided(inp.data_ptr() + 4 * col0, L.ptr(out), M, C_, ld_in, L.stream_ptr()), "gvk_copy_f32_strided")
        return
    _chk(inp, torch.bfloat16, "cast_bf16_f32 in", M * ld_in)
    _chk(out, torch.float32, "cast_bf16_f32 out", M * C_)
    L.check(L.load().gvk_cast_bf16_f32_strided(inp.data_ptr() + 2 * col0, L.ptr(out), M, C_, ld_in, L.stream_ptr()), "gvk_cast_bf16_f32_strided")


def lora_merge(w, a_q, b_q, a_v, b_v, out, C_, r, s):
    for t, n in ((w, "w"), (a_q, "a_q"), (b_q, "b_q"), (a_v, "a_v"), (b_v, "b_v"), (out, "out")):
        _chk(t, torch.float32, "lora_merge " + n)
    if w.numel() != 3 * C_ * C_ or out.numel() < 3 * C_ * C_ or a_q.numel() != r * C_ or b_q.numel() != C_ * r:
        raise L.GavikoHipError("lora_merge: shape mismatch")
    L.check(L.load().gvk_lora_merge_f32(L.ptr(w), L.ptr(a_q), L.ptr(b_q), L.ptr(a_v), L.ptr(b_v), L.ptr(out), C_, r, float(s), L.stream_ptr()),
            "gvk_lora_merge_f32")


def reduce_batch(jobs, scratch):
    """jobs: list of (a [M,J], b [M,L] or None, out, accumulate[, a2 [M2,J]]).  One launch pair for up to 8 small column-sum / J x L
    wgrad reductions; a2 (column sums only) appends more rows to the same sum.  scratch f32 >= 32 * total outputs."""
    arr = (L.ReduceJob * len(jobs))()
    total = 0
    for k, job in enumerate(jobs):
        a, b, out, acc = job[:4]
        a2 = job[4] if len(job) > 4 else None
        _chk(a2, torch.float32, "reduce_batch a2")
        _chk(a, torch.float32, "reduce_batch a")
        _chk(b, torch.float32, "reduce_batch b")
        _chk(out, torch.float32, "reduce_batch out")
        M, J = a.shape[0], a.numel() // a.shape[0]
        Lb = 0 if b is None else b.numel() // b.shape[0]
        if b is not None and b.shape[0] != M:
            raise L.GavikoHipError("reduce_batch: a and b must have the same number of rows")
        if out.numel() < (J * Lb if b is not None else J):
            raise L.GavikoHipError("reduce_batch: out too small")
        total += J * Lb if b is not None else J
        if a2 is not None and (b is not None or a2.numel() // a2.shape[0] != J):
            raise L.GavikoHipError("reduce_batch: a2 goes with column sums of the same width only")
        arr[k] = L.ReduceJob(L.ptr(a), L.ptr(b), L.ptr(out), L.ptr(a2), M, J, Lb, int(bool(acc)), 0 if a2 is None else a2.shape[0])
    _chk(scratch, torch.float32, "reduce_batch scratch", 32 * total)
    L.check(L.load().gvk_reduce_batch(arr, len(jobs), L.ptr(scratch), L.stream_ptr()), "gvk_reduce_batch")


def ln_lowrank_affine(Q, S, W, gamma, beta, dW, dgamma, dbeta, dbias, Lat, C_, accumulate=False):
    for t, n in ((Q, "Q"), (S, "S"), (W, "W"), (gamma, "gamma"), (beta, "beta"), (dW, "dW"), (dgamma, "dgamma"), (dbeta, "dbeta"), (dbias, "dbias")):
        _chk(t, torch.float32, "ln_lowrank_affine " + n)
    if Q.numel() < Lat * C_ or W.numel() < Lat * C_ or dW.numel() < Lat * C_ or S.numel() < Lat:
        raise L.GavikoHipError("ln_lowrank_affine: buffer too small")
    L.check(L.load().gvk_ln_lowrank_affine(L.ptr(Q), L.ptr(S), L.ptr(W), L.ptr(gamma), L.ptr(beta), L.ptr(dW), L.ptr(dgamma), L.ptr(dbeta),
                                           L.ptr(dbias), Lat, C_, int(bool(accumulate)), L.stream_ptr()), "gvk_ln_lowrank_affine")


# ---- SSF (model/ssf.py): effective parameters and scale/shift gradients ---------------------------------------------------
def ssf_fold_weight(w, s, out, out_t=None):
    """out[n][k] = w[n][k] * s[n] in out's dtype (bf16 / fp32); out_t [K][N] = the transpose (optional)."""
    _chk(w, torch.float32, "ssf_fold_weight w")
    N = w.shape[0]
    K = w.numel() // N
    _chk(s, torch.float32, "ssf_fold_weight s", N)
    if out.dtype not in (torch.bfloat16, torch.float32) or (out_t is not None and out_t.dtype != out.dtype):
        raise L.GavikoHipError("ssf_fold_weight: outputs must be bf16 or fp32 (both the same)")
    _chk(out, out.dtype, "ssf_fold_weight out", N * K)
    _chk(out_t, out.dtype, "ssf_fold_weight out_t", N * K)
    L.check(L.load().gvk_ssf_fold_weight(L.ptr(w), L.ptr(s), L.ptr(out), L.ptr(out_t), N, K, int(out.dtype == torch.float32), L.stream_ptr()),
            "gvk_ssf_fold_weight")


def ssf_fold_vec(a, s, t, out):
    n = s.numel()
    for x, nm in ((a, "a"), (s, "s"), (t, "t"), (out, "out")):
        _chk(x, torch.float32, "ssf_fold_vec " + nm, n)
    L.check(L.load().gvk_ssf_fold_vec(L.ptr(a), L.ptr(s), L.ptr(t), L.ptr(out), n, L.stream_ptr()), "gvk_ssf_fold_vec")


def ssf_colgrad(dy, y0, s, t, ds, dt, scratch, M, N, *, y1=None, pos=None, ld_dy=None, ld_y=None, rows_in=0, rows_out=0, row_off=0,
                y0_cols=0, y0_mul=1.0, y_mul=1.0):
    for x, nm in ((s, "s"), (t, "t"), (ds, "ds"), (dt, "dt")):
        _chk(x, torch.float32, "ssf_colgrad " + nm, N)
    _chk(scratch, torch.float32, "ssf_colgrad scratch", 64 * 2 * N)
    _chk(y1, torch.float32, "ssf_colgrad y1")
    _chk(pos, torch.float32, "ssf_colgrad pos")
    for x, nm in ((dy, "dy"), (y0, "y0")):
        if x.dtype not in (torch.bfloat16, torch.float32) or not x.is_cuda or not x.is_contiguous():
            raise L.GavikoHipError(f"ssf_colgrad {nm}: expected a contiguous bf16 / fp32 device tensor")
    d = L.SsfColgradDesc(dy=L.ptr(dy), y0=L.ptr(y0), y1=L.ptr(y1), pos=L.ptr(pos), s=L.ptr(s), t=L.ptr(t), ds=L.ptr(ds), dt=L.ptr(dt),
                         scratch=L.ptr(scratch), M=M, N=N, ld_dy=N if ld_dy is None else ld_dy, ld_y=N if ld_y is None else ld_y,
                         dy_f32=int(dy.dtype == torch.float32), y0_f32=int(y0.dtype == torch.float32), rows_in=rows_in, rows_out=rows_out,
                         row_off=row_off, y0_cols=int(y0_cols), y0_mul=float(y0_mul), y_mul=float(y_mul))
    L.check(L.load().gvk_ssf_colgrad(C.byref(d), L.stream_ptr()), "gvk_ssf_colgrad")


def ssf_ln_grad(dgamma_eff, dbeta_eff, gamma, beta, ds, dt):
    n = gamma.numel()
    for x, nm in ((dgamma_eff, "dgamma'"), (dbeta_eff, "dbeta'"), (gamma, "gamma"), (beta, "beta"), (ds, "ds"), (dt, "dt")):
        _chk(x, torch.float32, "ssf_ln_grad " + nm, n)
    L.check(L.load().gvk_ssf_ln_grad(L.ptr(dgamma_eff), L.ptr(dbeta_eff), L.ptr(gamma), L.ptr(beta), L.ptr(ds), L.ptr(dt), n, L.stream_ptr()),
            "gvk_ssf_ln_grad")


def ssf_head_grad(g, mean, rstd, wh, dlogits, gamma, beta, ds, dt, B, T, C_, K, r0, R):
    for x, nm in ((g, "g"), (mean, "mean"), (rstd, "rstd"), (wh, "wh"), (dlogits, "dlogits"), (gamma, "gamma"), (beta, "beta"), (ds, "ds"), (dt, "dt")):
        _chk(x, torch.float32, "ssf_head_grad " + nm)
    L.check(L.load().gvk_ssf_head_grad(L.ptr(g), L.ptr(mean), L.ptr(rstd), L.ptr(wh), L.ptr(dlogits), L.ptr(gamma), L.ptr(beta), L.ptr(ds),
                                       L.ptr(dt), B, T, C_, K, r0, R, L.stream_ptr()), "gvk_ssf_head_grad")


# ---- DVPT (model/dvpt.py) ------------------------------------------------------------------------------------------------
def dvpt_fwd(**kw):
    d = _desc(L.DvptDesc, "dvpt_fwd", **kw)
    L.check(L.load().gvk_dvpt_fwd(C.byref(d), L.stream_ptr()), "gvk_dvpt_fwd")


def dvpt_bwd(**kw):
    d = _desc(L.DvptDesc, "dvpt_bwd", **kw)
    L.check(L.load().gvk_dvpt_bwd(C.byref(d), L.stream_ptr()), "gvk_dvpt_bwd")


def scale_dev_(t: torch.Tensor, alpha: torch.Tensor) -> None:
    """t *= alpha[0] with alpha a device scalar (no host read)."""
    _chk(t, torch.float32, "scale_dev_ x")
    _chk(alpha, torch.float32, "scale_dev_ alpha", 1)
    L.check(L.load().gvk_scale_dev(L.ptr(t), L.ptr(alpha), t.numel(), L.stream_ptr()), "gvk_scale_dev")


# ---- EVP (model/evp.py) --------------------------------------------------------------------------------------------------
def evp_highpass(img, hp, depth_mask, out):
    _chk(img, torch.float32, "evp_highpass img")
    B, ch, D, H, W = img.shape
    _chk(hp, torch.float32, "evp_highpass hp", H * H)
    _chk(depth_mask, torch.int32, "evp_highpass depth_mask", D)
    _chk(out, torch.float32, "evp_highpass out", img.numel())
    L.check(L.load().gvk_evp_highpass(L.ptr(img), L.ptr(hp), L.ptr(depth_mask), L.ptr(out), B * ch, D, H, W, L.stream_ptr()), "gvk_evp_highpass")


def pad2d(src, rows, cols, dst, drows, dcols, *, ld_src=None, ld_dst=None, transpose=False):
    """dst[drows x dcols] = src[rows x cols] (optionally transposed) zero-padded."""
    _chk(src, torch.float32, "pad2d src")
    _chk(dst, torch.float32, "pad2d dst")
    L.check(L.load().gvk_pad2d_f32(L.ptr(src), cols if ld_src is None else ld_src, rows, cols, int(transpose), L.ptr(dst),
                                   dcols if ld_dst is None else ld_dst, drows, dcols, L.stream_ptr()), "gvk_pad2d_f32")


def add2d(a, lda, b, ldb, out, ldo, rows, cols):
    for t, n in ((a, "a"), (b, "b"), (out, "out")):
        _chk(t, torch.float32, "add2d " + n)
    L.check(L.load().gvk_add2d_f32(L.ptr(a), lda, L.ptr(b), ldb, L.ptr(out), ldo, rows, cols, L.stream_ptr()), "gvk_add2d_f32")


def gelu_fwd(x, y):
    _chk(x, torch.float32, "gelu_fwd x")
    _chk(y, torch.float32, "gelu_fwd y", x.numel())
    L.check(L.load().gvk_gelu_fwd_f32(L.ptr(x), L.ptr(y), x.numel(), L.stream_ptr()), "gvk_gelu_fwd_f32")


def gelu_bwd(dy, x, dx):
    for t, n in ((dy, "dy"), (x, "x"), (dx, "dx")):
        _chk(t, torch.float32, "gelu_bwd " + n, x.numel())
    L.check(L.load().gvk_gelu_bwd_f32(L.ptr(dy), L.ptr(x), L.ptr(dx), x.numel(), L.stream_ptr()), "gvk_gelu_bwd_f32")


def rows_patch(tok, src, pos, B, T, N, C_, row_off, accumulate):
    _chk(tok, torch.float32, "rows_patch tok", B * T * C_)
    _chk(src, torch.float32, "rows_patch src", B * N * C_)
    _chk(pos, torch.float32, "rows_patch pos", N * C_)
    L.check(L.load().gvk_rows_patch(L.ptr(tok), L.ptr(src), L.ptr(pos), B, T, N, C_, row_off, int(accumulate), L.stream_ptr()), "gvk_rows_patch")


def rows_gather(tok, dst, B, T, N, C_, row_off):
    _chk(tok, torch.float32, "rows_gather tok", B * T * C_)
    _chk(dst, torch.float32, "rows_gather dst", B * N * C_)
    L.check(L.load().gvk_rows_gather(L.ptr(tok), L.ptr(dst), B, T, N, C_, row_off, L.stream_ptr()), "gvk_rows_gather")


# ---- loss seed (train.py:176-179, 283/306, 327-328) -------------------------------------------------------------------------
LOSS_CE, LOSS_FOCAL = 0, 1


def loss_fwd_bwd(logits, target, loss, dlogits, kind, gamma=0.0, eps=1e-16, ignore_index=-100, weights=None, meter=None, reduction="mean"):
    """loss[0] = criterion(logits, target), dlogits = its gradient, meter += (loss*B, #correct, B): one launch, no host read."""
    _chk(logits, torch.float32, "loss logits")
    B, K = logits.shape
    if target.dtype != torch.int64 or not target.is_contiguous() or target.numel() != B:
        raise ValueError("loss target must be a contiguous int64 [B] tensor")
    _chk(loss, torch.float32, "loss out", B if reduction == "none" else 1)
    _chk(dlogits, torch.float32, "loss dlogits", B * K)
    if weights is not None:
        _chk(weights, torch.float32, "loss weights", K)
    if meter is not None:
        _chk(meter, torch.float32, "loss meter", 3)
    d = L.LossDesc(logits=L.ptr(logits), target=L.ptr(target), weights=L.ptr(weights) if weights is not None else None,
                   loss=L.ptr(loss), dlogits=L.ptr(dlogits), meter=L.ptr(meter) if meter is not None else None,
                   B=B, K=K, kind=kind, reduction={"mean": 0, "sum": 1, "none": 2}[reduction], gamma=gamma, eps=eps, ignore_index=ignore_index)
    L.check(L.load().gvk_loss_fwd_bwd(C.byref(d), L.stream_ptr()), "gvk_loss_fwd_bwd")


# ---- data side + evaluation metrics (train.py:38-62, eval.py:103-122) --------------------------------------------------------
def minmax_partials(B: int, device) -> torch.Tensor:
    return torch.empty(B * L.load().gvk_minmax_partials(), dtype=torch.float32, device=device)


def volume_minmax(x: torch.Tensor, partials: torch.Tensor) -> None:
    _chk(x, torch.float32, "volume_minmax x")
    B = x.shape[0]
    _chk(partials, torch.float32, "volume_minmax partials", B * L.load().gvk_minmax_partials())
    L.check(L.load().gvk_volume_minmax(L.ptr(x), L.ptr(partials), B, x.numel() // B, L.stream_ptr()), "gvk_volume_minmax")


def rescale_intensity(x, partials, y, out_min=0.0, out_max=1.0, minmax=None) -> None:
    _chk(x, torch.float32, "rescale_intensity x")
    _chk(y, torch.float32, "rescale_intensity y", x.numel())
    B = x.shape[0]
    if minmax is not None:
        _chk(minmax, torch.float32, "rescale_intensity minmax", 2 * B)
    L.check(L.load().gvk_rescale_intensity(L.ptr(x), L.ptr(partials), L.ptr(y), L.ptr(minmax) if minmax is not None else None, B, x.numel() // B,
                                           out_min, out_max, L.stream_ptr()), "gvk_rescale_intensity")


def spatial_transform(x, out, mats, flags, partials) -> None:
    _chk(x, torch.float32, "spatial_transform in")
    _chk(out, torch.float32, "spatial_transform out", x.numel())
    B, D, H, W = x.shape[0], x.shape[-3], x.shape[-2], x.shape[-1]
    _chk(mats, torch.float32, "spatial_transform mats", 12 * B)
    _chk(flags, torch.int32, "spatial_transform flags", B)
    L.check(L.load().gvk_spatial_transform(L.ptr(x), L.ptr(out), L.ptr(mats), L.ptr(flags), L.ptr(partials), B, D, H, W, L.stream_ptr()),
            "gvk_spatial_transform")


def eval_rows(logits, target, proba, pred, confusion) -> None:
    _chk(logits, torch.float32, "eval_rows logits")
    N, K = logits.shape
    _chk(proba, torch.float32, "eval_rows proba", N * K)
    _chk(pred, torch.int32, "eval_rows pred", N)
    _chk(confusion, torch.int64, "eval_rows confusion", K * K)
    if target.dtype != torch.int64 or target.numel() != N or not target.is_contiguous():
        raise ValueError("eval_rows target must be a contiguous int64 [N] tensor")
    L.check(L.load().gvk_eval_rows(L.ptr(logits), L.ptr(target), L.ptr(proba), L.ptr(pred), L.ptr(confusion), N, K, L.stream_ptr()), "gvk_eval_rows")


def ovr_auc_counts(proba, target, counts) -> None:
    _chk(proba, torch.float32, "ovr_auc proba")
    N, K = proba.shape
    _chk(counts, torch.int64, "ovr_auc counts", 3 * K)
    L.check(L.load().gvk_ovr_auc_counts(L.ptr(proba), L.ptr(target), L.ptr(counts), N, K, L.stream_ptr()), "gvk_ovr_auc_counts")


def dropout_rows(x, drop_p, seed, seed_ptr, out32=None, out16=None, M=None, N=None, rows_in=0, rows_out=0, row_off=0):
    """out = x * mask / (1 - p) over logical rows (optionally a row range of every sample); out32 may be x itself."""
    _chk(x, torch.float32, "dropout_rows x")
    ld = x.shape[-1]
    N = ld if N is None else N
    M = x.numel() // ld if M is None else M
    if out32 is not None:
        _chk(out32, torch.float32, "dropout_rows out32")
    if out16 is not None:
        _chk(out16, torch.bfloat16, "dropout_rows out16")
    d = L.DropoutDesc(x=L.ptr(x), out32=L.ptr(out32) if out32 is not None else None, out16=L.ptr(out16) if out16 is not None else None,
                      seed_ptr=L.ptr(seed_ptr), M=M, N=N, ld=ld, rows_in=rows_in, rows_out=rows_out, row_off=row_off, drop_p=float(drop_p), seed=int(seed))
    L.check(L.load().gvk_dropout_rows(C.byref(d), L.stream_ptr()), "gvk_dropout_rows")
