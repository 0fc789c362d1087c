"""Deterministic, formula-based synthetic weights and MRI volumes.

Everything is a pure function of (tensor name, element index): a splitmix64 finaliser over a
counter, so any process / rank / box regenerates bit-identical values without files and without
depending on torch's RNG streams.  Used by bench.py (random-init weights of the named
architecture, synthetic 120x160x160 volumes), by tools/gen_golden.py (fixtures generated from
the reference) and by the tests.

Scales are chosen so the path is numerically *alive* (softmax far from uniform, LayerNorm affine
non-trivial, gates away from 0.5) while the residual stream stays O(1) -- see DESIGN.md.
"""
from __future__ import annotations

import math
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def uniform01(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n float32 values in [0,1): element i = hash(seed, offset+i) >> 40 scaled by 2^-24."""
    with np.errstate(over="ignore"):
        ctr = np.arange(offset, offset + n, dtype=np.uint64)
        key = _splitmix64(np.full(1, seed & 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))[0]
        h = _splitmix64(ctr ^ key)
    return ((h >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def name_seed(name: str, salt: int = 0) -> int:
    return (zlib.crc32(name.encode()) + 0x100000000 * (salt + 1)) & 0xFFFFFFFFFFFFFFFF


def symmetric(name: str, shape, amp: float, salt: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    return ((uniform01(name_seed(name, salt), n) * 2.0 - 1.0) * np.float32(amp)).reshape(shape).astype(np.float32)


def volume(sample_index: int, shape=(1, 120, 160, 160), seed: int = 2025) -> np.ndarray:
    """One synthetic MRI volume in [0,1) (the range RescaleIntensity(0,1) yields): counter-hash uniform noise times a
    smooth, sample-dependent intensity profile along the depth axis (so different samples give different logits)."""
    n = int(np.prod(shape))
    u = uniform01(seed * 1000003 + sample_index, n).reshape(shape)
    z = np.arange(shape[-3], dtype=np.float64) / shape[-3]
    prof = 0.55 + 0.45 * np.cos(2.0 * np.pi * (z * (1 + sample_index % 3) + 0.37 * sample_index))
    return (u * prof.astype(np.float32).reshape((-1, 1, 1))).astype(np.float32)


def volumes(first: int, count: int, shape=(1, 120, 160, 160), seed: int = 2025) -> np.ndarray:
    return np.stack([volume(first + i, shape, seed) for i in range(count)], 0)


def labels(first: int, count: int, num_classes: int = 5) -> np.ndarray:
    return (np.arange(first, first + count) % num_classes).astype(np.int64)


def _is_norm_name(name: str) -> bool:
    return (
        ".norm." in "." + name
        or name.endswith("net.0.weight") or name.endswith("net.0.bias")
        or "cls_analyzer_.0." in name or "gl_balancer_.0." in name
        or "adapter_layer_norm_before" in name
    )


def fill_param(name: str, shape, salt: int = 0) -> np.ndarray:
    """The recipe. `name` is the state_dict key; rules are by key pattern."""
    shape = tuple(int(s) for s in shape)
    if _is_norm_name(name):                      # LayerNorm affine: gamma ~ 1 +- .2, beta ~ +- .1
        if name.endswith("weight"):
            return 1.0 + symmetric(name, shape, 0.2, salt)
        return symmetric(name, shape, 0.1, salt)
    if name.endswith("prompt_gate"):              # DVPT's scalar gate is zero-initialised (dvpt.py:31): make the adapter path count
        return 0.5 + symmetric(name, shape, 0.2, salt)
    if "ssf_scale_" in name:                      # SSF per-channel scale ~ 1 +- .2 / shift ~ +- .1 (ssf.py:14-20 inits N(1,.02) / N(0,.02))
        return 1.0 + symmetric(name, shape, 0.2, salt)
    if "ssf_shift_" in name:
        return symmetric(name, shape, 0.1, salt)
    if name.endswith("bias"):
        if "gl_balancer_.1" in name:
            return 0.5 + symmetric(name, shape, 0.2, salt)
        return symmetric(name, shape, 0.05, salt)
    if "pos_embedding" in name or name.endswith("prompt_positional_embedding"):
        return symmetric(name, shape, 0.3, salt)
    if name.endswith("cls_token") or name.endswith("prompt_embeddings"):
        return symmetric(name, shape, 0.5, salt)
    if "deep_prompt_embeddings" in name:
        return symmetric(name, shape, 0.5, salt)
    if len(shape) >= 2:                           # Linear [out,in] / Conv3d [out,in,kd,kh,kw]
        fan_in = int(np.prod(shape[1:]))
        gain = 1.0
        if "to_out" in name or name.endswith("net.4.weight"):
            gain = 0.35                           # keep the residual stream O(1) over 12-24 layers
        elif "conv_proj" in name:
            gain = 1.7                            # inputs are U[0,1): bring tokens to ~unit scale
        elif "local_attns" in name and "proj_up" in name:
            gain = 0.3                            # keep the local stream O(1) over 12-24 layers
        elif "proj_up" in name or "up_adapter_proj" in name or "linear_b_" in name:
            gain = 0.5
        elif "query_proj" in name:
            gain = 2.0                            # make GXA/LXA softmaxes peaky
        elif "local_attns" in name and name.endswith("qkv.weight"):
            gain = 2.5                            # MWSA scale is dim^-1/2 over a 20-d latent: score std ~1-2, not chaotic
        elif "mlp_head" in name:
            gain = 1.0
        amp = gain * math.sqrt(3.0 / fan_in)
        return symmetric(name, shape, amp, salt)
    return symmetric(name, shape, 0.05, salt)


def fill_state_dict(shapes: dict, salt: int = 0) -> dict:
    """shapes: {key: shape}. Alias keys (gaviko global_query/local_query) follow their target."""
    out = {}
    for k, shp in shapes.items():
        src = (k.replace(".global_query.", ".global_attention.query_proj.")
                .replace(".local_query.", ".local_attention.query_proj."))
        out[k] = fill_param(src, shp, salt)
    return out
