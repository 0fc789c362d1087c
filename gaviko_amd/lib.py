"""ctypes binding of libgaviko_hip.so (the C-ABI declared in include/gaviko_hip.h).

There is no fallback: if the shared library is missing or a call fails, this raises.  torch is used only
for device memory (tensors -> raw pointers) and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The step runs on three streams (+ the collective's and RCCL's own
# in data-parallel runs); two of them landing on one queue serialises them: 5.9 -> 6.9-7.7 ms per step when a second model instance's
# side streams wrapped around onto the main stream's queue (tools/bench_reducer.py).  Must be set before the first HIP call.
if os.environ.get("GPU_MAX_HW_QUEUES") is None:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    try:                                          # the variable is read when the HIP runtime starts: too late if torch has already started it
        import sys as _sys
        _t = _sys.modules.get("torch")
        if _t is not None and _t.cuda.is_initialized():
            import warnings as _w
            _w.warn("gaviko_amd: HIP was initialised before gaviko_amd was imported, so GPU_MAX_HW_QUEUES=8 cannot take effect; the three "
                    "streams of a step may share hardware queues (slower, not wrong).  Import gaviko_amd first or export the variable.")
    except Exception:
        pass

ABI_VERSION = 12                                  # gvk_abi_version() of the library these declarations describe
# GAVIKO_HIP_DIAG=1 (tools/ only): load the measurement build libgaviko_hip_diag.so (`python -m gaviko_amd.build --diag`) -- the product
# library ignores every A/B switch of the kernel sources and exports no diagnostics (include/gaviko_hip_diag.h)
DIAG = os.environ.get("GAVIKO_HIP_DIAG", "0") == "1"
LIB_PATH = os.environ.get("GAVIKO_HIP_LIB") or os.path.join(_HERE, "libgaviko_hip_diag.so" if DIAG else "libgaviko_hip.so")


def diag_env(name: str, default=None):
    """An A/B switch of the measurement build: read from the environment only when GAVIKO_HIP_DIAG=1, else its measured-best default."""
    return os.environ.get(name, default) if DIAG else default


class GemmDesc(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("w", C.c_void_p), ("out0", C.c_void_p), ("out1", C.c_void_p),
        ("bias", C.c_void_p), ("res", C.c_void_p), ("aux", C.c_void_p), ("pos", C.c_void_p), ("seed_ptr", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldw", C.c_int32), ("ldo", C.c_int32), ("ldres", C.c_int32), ("ldaux", C.c_int32),
        ("epilogue", C.c_int32), ("rows_in", C.c_int32), ("rows_out", C.c_int32), ("row_off", C.c_int32),
        ("tile", C.c_int32), ("drop_p", C.c_float), ("seed", C.c_uint64),
        ("scale_cols", C.c_int32), ("col_scale", C.c_float),
        ("ln_mean", C.c_void_p), ("ln_rstd", C.c_void_p), ("ln_c1", C.c_void_p), ("stat_part", C.c_void_p), ("stat_pivot", C.c_void_p),
        ("m_panels", C.c_int32), ("m_stride", C.c_int32), ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_uint64),
        ("ksplit", C.c_int32), ("aux_is_grad", C.c_int32),
    ]


def _struct(name, ptrs, ints, floats=(), u64=(), i64=()):
    """C struct with the header's field order: pointers, int32s, floats, uint64s, int64s."""
    fields = [(f, C.c_void_p) for f in ptrs] + [(f, C.c_int32) for f in ints] + [(f, C.c_float) for f in floats] + \
             [(f, C.c_uint64) for f in u64] + [(f, C.c_int64) for f in i64]
    return type(name, (C.Structure,), {"_fields_": fields})


SkinnyDownDesc = _struct("SkinnyDownDesc",
                         ["x", "w", "bias", "ln_gamma", "ln_beta", "mean", "rstd", "z", "y", "w2", "y2", "seed_ptr"],
                         ["M", "C", "L", "L2", "act", "w_layout", "act_in"], ["eps", "drop_p"], ["seed"])
SkinnyUpDesc = _struct("SkinnyUpDesc", ["lat", "w", "bias", "res", "out", "lat_override", "seed_ptr", "ln_x", "ln_mean", "ln_rstd", "ln_gamma", "out_bf16",
                                        "alpha_ptr", "gg_x", "w2", "bias2", "z2", "y2", "lat_b", "w_b",
                                        "nx_w", "nx_bias", "nx_ln_gamma", "nx_ln_beta", "nx_mean", "nx_rstd", "nx_lat", "nx_w2", "nx_y2"],
                       ["M", "C", "L", "T", "P", "w_layout", "accumulate", "L2", "act2", "w2_layout", "nx_L2"], ["drop_p", "drop2_p", "nx_eps"],
                       ["seed", "seed2"])
OuterDesc = _struct("OuterDesc", ["narrow", "wide", "lat_override", "mean", "rstd", "ln_gamma", "ln_beta", "scratch", "out", "colsum", "seed_ptr",
                                  "narrow2", "wide2"],
                    ["M", "C", "L", "T", "P", "transposed", "accumulate", "wide_act", "M2"], ["drop_p"], ["seed"])
WindowAttnDesc = _struct("WindowAttnDesc", ["qkv", "ctx", "lse", "dctx", "delta", "dqkv", "seed_ptr"],
                         ["B", "D", "H", "W", "kd", "kh", "kw", "L"], ["scale", "drop_p"], ["seed"])
GpaDesc = _struct("GpaDesc",
                  ["xl", "ll", "ca0_g", "ca0_b", "ca1_w", "ca1_b", "ca3_w", "ca3_b", "gl0_g", "gl0_b", "gl1_w", "gl1_b",
                   "wgq", "bgq", "wlq", "blq", "imp", "gw", "enh", "prm", "qg", "ql", "cg", "cl", "lse_g", "lse_l",
                   "dcomb", "zx", "zl", "dimp", "dgw_part", "dqg", "dql", "dcg", "dcl", "delta_g", "delta_l", "dprm",
                   "dcls", "gate_partials", "dzx", "dzl", "enh16"],
                  ["B", "T", "N", "P", "L", "ld16", "col16"], ["scale"])
SsfColgradDesc = _struct("SsfColgradDesc", ["dy", "y0", "y1", "pos", "s", "t", "ds", "dt", "scratch"],
                         ["M", "N", "ld_dy", "ld_y", "dy_f32", "y0_f32", "rows_in", "rows_out", "row_off", "y0_cols"], ["y0_mul", "y_mul"])
DvptDesc = _struct("DvptDesc", ["z", "enh", "lse", "dcomb", "gate", "bu", "colsum_dy", "delta", "dz", "dgate"], ["B", "T", "P", "L", "C"], ["scale"])
AdamDesc = _struct("AdamDesc", ["ptr_tab", "blk_tab", "grad", "m", "v", "norm_sq"], ["nblocks"],
                   ["lr", "beta1", "beta2", "eps", "bias_c1", "bias_c2", "max_norm"])
LossDesc = _struct("LossDesc", ["logits", "target", "weights", "loss", "dlogits", "meter"], ["B", "K", "kind", "reduction"], ["gamma", "eps"], i64=["ignore_index"])
DropoutDesc = _struct("DropoutDesc", ["x", "out32", "out16", "seed_ptr"], ["M", "N", "ld", "rows_in", "rows_out", "row_off"], ["drop_p"], ["seed"])
RowProjDesc = _struct("RowProjDesc", ["w", "bias", "y", "z", "y_split"], ["L", "w_layout", "act", "ld_split", "col_split"])
LnBwdDy16Desc = _struct("LnBwdDy16Desc", ["dy_bf16", "x", "mean", "rstd", "gamma", "dres", "dx", "dx_bf16", "proj"],
                        ["M", "C", "groups", "rows_per_group", "group_stride"])
ReduceJob = _struct("ReduceJob", ["a", "b", "out", "a2"], ["M", "J", "L", "accumulate", "M2"])
PgradOuter = _struct("PgradOuter", ["narrow", "wide", "narrow2", "wide2", "lat_override", "mean", "rstd", "out", "colsum",
                                    "aff_w", "aff_gamma", "aff_beta", "aff_dgamma", "aff_dbeta", "aff_dbias"],
                     ["M", "M2", "T", "P", "transposed", "accumulate", "C"], ["drop_p"], ["seed"])
HeadDesc = _struct("HeadDesc", ["g", "ln_gamma", "ln_beta", "wh", "bh", "logits", "pooled", "dlogits", "dg", "dwh", "dbh"],
                   ["B", "T", "C", "K", "r0", "R", "accumulate"])

(EPI_STORE_BF16, EPI_BIAS_RES_F32, EPI_BIAS_GELU_BF16, EPI_PATCH_F32, EPI_GELU_BWD_BF16, EPI_STORE_F32, EPI_BIAS_RES_F32_BF16,
 EPI_BIAS_RELU_BF16, EPI_RELU_BWD_BF16) = range(9)

_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64
# name -> argtypes (every function returns int and takes the stream last)
SIGNATURES = {
    "gvk_gemm_nt_bf16": [C.POINTER(GemmDesc), _P],
    "gvk_gemm_nt_f32": [C.POINTER(GemmDesc), _P],
    "gvk_attention_fwd_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "gvk_attention_bwd_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "gvk_patchify_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "gvk_transpose_f32": [_P, _P, _I, _I, _P],
    "gvk_transpose_bf16": [_P, _P, _I, _I, _P],
    "gvk_copy_async": [_P, _P, C.c_size_t, _P],
    "gvk_copy_f32_strided": [_P, _P, _I, _I, _I, _P],
    "gvk_cast_f32_bf16": [_P, _P, _L, _P],
    "gvk_transpose_cast_f32_bf16": [_P, _P, _I, _I, _P],
    "gvk_pack_split_bf16": [_P, _I, _P, _P, _I, _I, _I, _I, _P],
    "gvk_prompt_up_fix": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gvk_layernorm_fwd_fix": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _I, _I, _I, _P],
    "gvk_patchify_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "gvk_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "gvk_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "gvk_layernorm_bwd_rows": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gvk_layernorm_bwd_affine": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "gvk_attention_fwd_bf16": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "gvk_qkv_prescale_bf16": [_P, _I, _I, _I, _F, _P],
    "gvk_prompt_up_fix_stats": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "gvk_attention_bwd_bf16": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "gvk_attention_bwd_bf16_rows": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "gvk_attention_bwd_bf16_fused": [_P, _P, _P, _P, _P, _P, _P, C.c_size_t, _I, _I, _I, _I, _I, _F, _P],
    "gvk_attention_fwd_f32_dropout": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, C.c_uint64, _P, _P],
    "gvk_attention_bwd_f32_dropout": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, C.c_uint64, _P, _P],
    "gvk_attention_fwd_bf16_dropout": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, C.c_uint64, _P, _P],
    "gvk_attention_bwd_bf16_dropout": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, C.c_uint64, _P, _P],
    "gvk_skinny_down": [C.POINTER(SkinnyDownDesc), _P],
    "gvk_skinny_up": [C.POINTER(SkinnyUpDesc), _P],
    "gvk_outer_reduce": [C.POINTER(OuterDesc), _P],
    "gvk_small_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gvk_ln_lowrank_affine": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "gvk_colsum": [_P, _P, _P, _I, _I, _I, _P],
    "gvk_reduce_batch": [C.POINTER(ReduceJob), _I, _P, _P],
    "gvk_param_grads": [C.POINTER(PgradOuter), _I, C.POINTER(ReduceJob), _I, _P, _L, _P, _I, _P, _I, _I, _P],
    "gvk_window_attn_fwd": [C.POINTER(WindowAttnDesc), _P],
    "gvk_window_attn_bwd": [C.POINTER(WindowAttnDesc), _P],
    "gvk_gpa_fwd": [C.POINTER(GpaDesc), _P],
    "gvk_gpa_bwd": [C.POINTER(GpaDesc), _P],
    "gvk_rows_broadcast": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gvk_rows_batch_sum": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "gvk_small_linear_fwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "gvk_small_linear_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gvk_vpt_repack_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "gvk_vpt_repack_bwd": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "gvk_cast_bf16_f32_strided": [_P, _P, _I, _I, _I, _P],
    "gvk_lora_merge_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "gvk_layernorm_fwd_proj": [_P, _P, _P, _P, _P, _P, _I, _I, _F, C.POINTER(RowProjDesc), _P],
    "gvk_layernorm_bwd_dy16": [C.POINTER(LnBwdDy16Desc), _P],
    "gvk_layernorm_bwd_proj": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, C.POINTER(RowProjDesc), _P],
    "gvk_layernorm_bwd_up": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gvk_ssf_fold_weight": [_P, _P, _P, _P, _I, _I, _I, _P],
    "gvk_ssf_fold_vec": [_P, _P, _P, _P, _I, _P],
    "gvk_ssf_colgrad": [C.POINTER(SsfColgradDesc), _P],
    "gvk_ssf_ln_grad": [_P, _P, _P, _P, _P, _P, _I, _P],
    "gvk_ssf_head_grad": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "gvk_dvpt_fwd": [C.POINTER(DvptDesc), _P],
    "gvk_dvpt_bwd": [C.POINTER(DvptDesc), _P],
    "gvk_scale_dev": [_P, _P, C.c_int64, _P],
    "gvk_evp_highpass": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gvk_pad2d_f32": [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P],
    "gvk_add2d_f32": [_P, _I, _P, _I, _P, _I, _I, _I, _P],
    "gvk_gelu_fwd_f32": [_P, _P, C.c_int64, _P],
    "gvk_gelu_bwd_f32": [_P, _P, _P, C.c_int64, _P],
    "gvk_rows_patch": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "gvk_rows_gather": [_P, _P, _I, _I, _I, _I, _I, _P],
    "gvk_sumsq": [_P, C.c_int64, _P, _P, _P],
    "gvk_adam_step": [C.POINTER(AdamDesc), _P],
    "gvk_loss_fwd_bwd": [C.POINTER(LossDesc), _P],
    "gvk_dropout_rows": [C.POINTER(DropoutDesc), _P],
    "gvk_volume_minmax": [_P, _P, _I, _L, _P],
    "gvk_rescale_intensity": [_P, _P, _P, _P, _I, _L, _F, _F, _P],
    "gvk_spatial_transform": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gvk_eval_rows": [_P, _P, _P, _P, _P, _I, _I, _P],
    "gvk_ovr_auc_counts": [_P, _P, _P, _I, _I, _P],
    "gvk_memset_async": [_P, _I, C.c_size_t, _P],
    "gvk_seed_advance": [_P, C.c_uint64, _P],
    "gvk_scale_f32": [_P, _F, C.c_long, _P],
    "gvk_head_fwd": [C.POINTER(HeadDesc), _P],
    "gvk_head_bwd": [C.POINTER(HeadDesc), _P],
}
NO_STREAM = {"gvk_last_error": (C.c_char_p, []), "gvk_device_check": (C.c_int, []), "gvk_abi_version": (C.c_int, []),
             "gvk_attention_bwd_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
             "gvk_attention_bwd_status_offset": (C.c_size_t, [C.c_size_t]),
             "gvk_gpa_gate_param_count": (C.c_int, [C.c_int, C.c_int]), "gvk_gemm_stat_parts": (C.c_int, [C.c_int]), "gvk_minmax_partials": (C.c_int, []),
             "gvk_param_grads_scratch": (C.c_int64, [C.POINTER(PgradOuter), C.c_int, C.POINTER(ReduceJob), C.c_int, C.c_int, C.c_int]),
             "gvk_plan_begin": (C.c_int, []), "gvk_plan_end": (C.c_int, []), "gvk_plan_abort": (C.c_int, []),
             "gvk_plan_size": (C.c_int, [C.c_int]), "gvk_plan_replay": (C.c_int, [C.c_int]), "gvk_plan_free": (C.c_int, [C.c_int]),
             "gvk_plan_event_record": (C.c_int, [_P]), "gvk_plan_event_wait": (C.c_int, [_P, C.c_int]),
             "gvk_plan_event_record_fenced": (C.c_int, [_P]), "gvk_plan_event_stream_wait": (C.c_int, [C.c_int, C.c_int, _P]),
             "gvk_plan_set_timing": (C.c_int, [C.c_int]),
             "gvk_plan_event_elapsed": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)])}
STRUCTS = {"gvk_gemm_desc": GemmDesc, "gvk_skinny_down_desc": SkinnyDownDesc, "gvk_skinny_up_desc": SkinnyUpDesc,
           "gvk_outer_desc": OuterDesc, "gvk_window_attn_desc": WindowAttnDesc, "gvk_gpa_desc": GpaDesc, "gvk_head_desc": HeadDesc, "gvk_reduce_job": ReduceJob, "gvk_pgrad_outer": PgradOuter, "gvk_rowproj_desc": RowProjDesc, "gvk_adam_desc": AdamDesc, "gvk_loss_desc": LossDesc, "gvk_dropout_desc": DropoutDesc, "gvk_ssf_colgrad_desc": SsfColgradDesc, "gvk_dvpt_desc": DvptDesc}

# diag library only (include/gaviko_hip_diag.h): bound when GAVIKO_HIP_DIAG=1 selects libgaviko_hip_diag.so
DIAG_SIGNATURES = {}
DIAG_NO_STREAM = {"gvk_plan_nop_stream": (C.c_int, [_P]), "gvk_plan_nop_clear": (C.c_int, [])}

_lib = None


class GavikoHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library (once).  Raises GavikoHipError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GavikoHipError(
            f"{LIB_PATH} not found: build it with `python -m gaviko_amd.build` (hipcc, gfx950). "
            "gaviko_amd has no CPU or eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in NO_STREAM.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = C.c_int, args
    if DIAG:
        for name, (res, args) in DIAG_NO_STREAM.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        for name, args in DIAG_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = C.c_int, args
    if lib.gvk_abi_version() != ABI_VERSION:
        raise GavikoHipError(f"{LIB_PATH} has ABI version {lib.gvk_abi_version()}, these bindings describe {ABI_VERSION}: rebuild with "
                             "`python -m gaviko_amd.build`")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise GavikoHipError(f"{what} failed (rc={rc}): {load().gvk_last_error().decode()}")


_device_ok = False


def require_device() -> None:
    """The product path runs on gfx950 only."""
    global _device_ok
    if _device_ok:
        return
    if not torch.cuda.is_available():
        raise GavikoHipError("no HIP device visible: gaviko_amd runs on MI355X (gfx950) only and has no CPU fallback")
    rc = load().gvk_device_check()
    if rc != 950:
        raise GavikoHipError(f"gvk_device_check: {load().gvk_last_error().decode()}")
    _device_ok = True


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()
