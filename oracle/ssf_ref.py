"""ORACLE -- test infrastructure only (see oracle/vit_ref.py header).

CPU restatement of the SSF plugin (model/ssf.py, `--method ssf`): a ViT whose LayerNorm and Linear outputs each pass
through a trainable per-channel scale and shift, `ssf_ada(x, s, t) = x * s + t` (ssf.py:24-31).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F
from einops import rearrange

from .vit_ref import SD, Tensor, attention_shapes, ff_shapes, layer_norm, mapping_vit, patch_embed


def ssf_ada(sd: SD, prefix: str, idx: int, x: Tensor) -> Tensor:
    """x * scale + shift over the last dimension (ssf.py:24-27: every call site here is channels-last)."""
    return x * sd[f"{prefix}ssf_scale_{idx}"] + sd[f"{prefix}ssf_shift_{idx}"]


def ssf_attention(sd: SD, p: str, x: Tensor, heads: int, masks=None, layer: int = 0) -> Tensor:
    """ssf.py:104-122: LN -> ssf_0 -> bias-free qkv -> ssf_1 (over 3*inner) -> MHSA (scale after q.k^T) -> to_out -> ssf_2."""
    xn = ssf_ada(sd, p + ".", 0, layer_norm(sd, p + ".norm", x))
    qkv = ssf_ada(sd, p + ".", 1, F.linear(xn, sd[p + ".to_qkv.weight"]))
    q, k, v = (rearrange(t, "b n (h d) -> b h n d", h=heads) for t in qkv.chunk(3, dim=-1))
    dots = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    attn = dots.softmax(dim=-1)
    if masks is not None and ("attn", layer) in masks:       # explicit dropout masks (tests; live only with freeze_vit=False: ssf.py:205-217)
        attn = attn * masks[("attn", layer)]                  # ssf.py:115
    out = rearrange(torch.matmul(attn, v), "b h n d -> b n (h d)")
    y = ssf_ada(sd, p + ".", 2, F.linear(out, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"]))
    if masks is not None and ("proj", layer) in masks:
        y = y * masks[("proj", layer)]                        # ssf.py:121: the Dropout of to_out runs AFTER ssf_2
    return y


def ssf_feed_forward(sd: SD, p: str, x: Tensor, masks=None, layer: int = 0) -> Tensor:
    """ssf.py:64-74: LN -> ssf_0 -> fc1 -> ssf_1 -> erf-GELU -> fc2 -> ssf_2."""
    h = ssf_ada(sd, p + ".", 0, layer_norm(sd, p + ".net.0", x))
    h = ssf_ada(sd, p + ".", 1, F.linear(h, sd[p + ".net.1.weight"], sd[p + ".net.1.bias"]))
    h = F.gelu(h)
    if masks is not None and ("act", layer) in masks:
        h = h * masks[("act", layer)]                         # ssf.py:70
    y = ssf_ada(sd, p + ".", 2, F.linear(h, sd[p + ".net.4.weight"], sd[p + ".net.4.bias"]))
    if masks is not None and ("ff", layer) in masks:
        y = y * masks[("ff", layer)]                          # ssf.py:73: after ssf_2
    return y


def ssf_forward(sd: SD, img: Tensor, cfg: dict, taps: Optional[dict] = None) -> Tensor:
    """ScalingShiftingFeatures.forward (ssf.py:228-246) + Transformer.forward (133-138): the patch tokens are scaled/shifted
    BEFORE the cls token and the positional embedding are added (232-237); final norm -> ssf (138); head on CLS / mean."""
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    patch = (cfg["frame_patch_size"], cfg["image_patch_size"], cfg["image_patch_size"])
    x = ssf_ada(sd, "", 1, patch_embed(sd, "conv_proj.0", img, patch))
    b, n, _ = x.shape
    x = torch.cat((sd["cls_token"].expand(b, -1, -1), x), dim=1) + sd["pos_embedding"][:, : n + 1]
    masks = cfg.get("_masks")
    if masks is not None and ("emb", 0) in masks:
        x = x * masks[("emb", 0)]                             # self.dropout behind the position embedding
    for i in range(depth):
        p = f"transformer.layers.{i}"
        x = ssf_attention(sd, p + ".0", x, heads, masks, i) + x
        if taps is not None:
            taps[f"layer{i}.post_attn"] = x
        f = ssf_feed_forward(sd, p + ".1", x, masks, i)
        x = f + x
        if taps is not None:
            taps[f"layer{i}.ff_out"] = f
            taps[f"layer{i}.post_mlp"] = x
    x = layer_norm(sd, "transformer.norm", x)
    if taps is not None:
        taps["final_norm"] = x                            # the module output (before the scale/shift), as the fixtures tap it
    x = ssf_ada(sd, "transformer.", 1, x)
    x = x.mean(dim=1) if cfg.get("pool", "cls") == "mean" else x[:, 0]
    return F.linear(x, sd["mlp_head.weight"], sd["mlp_head.bias"])


def ssf_param_shapes(cfg: dict) -> Dict[str, tuple]:
    """state_dict of ScalingShiftingFeatures in registration order (ssf.py:166-178, 52-54, 94-96, 129-130)."""
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    fp, ip = cfg["frame_patch_size"], cfg["image_patch_size"]
    n = (cfg["frames"] // fp) * (cfg["image_size"] // ip) ** 2
    inner = heads * cfg.get("dim_head", 64)
    s = {"ssf_scale_1": (dim,), "ssf_shift_1": (dim,), "pos_embedding": (1, n + 1, dim), "cls_token": (1, 1, dim),
         "conv_proj.0.weight": (dim, cfg.get("channels", 3), fp, ip, ip), "conv_proj.0.bias": (dim,),
         "transformer.ssf_scale_1": (dim,), "transformer.ssf_shift_1": (dim,),
         "transformer.norm.weight": (dim,), "transformer.norm.bias": (dim,)}
    for i in range(depth):
        a, f = f"transformer.layers.{i}.0", f"transformer.layers.{i}.1"
        s.update({a + ".ssf_scale_0": (dim,), a + ".ssf_shift_0": (dim,), a + ".ssf_scale_1": (3 * inner,), a + ".ssf_shift_1": (3 * inner,),
                  a + ".ssf_scale_2": (dim,), a + ".ssf_shift_2": (dim,)})
        s.update(attention_shapes(a, dim, inner))
        s.update({f + ".ssf_scale_0": (dim,), f + ".ssf_shift_0": (dim,), f + ".ssf_scale_1": (mlp,), f + ".ssf_shift_1": (mlp,),
                  f + ".ssf_scale_2": (dim,), f + ".ssf_shift_2": (dim,)})
        s.update(ff_shapes(f, dim, mlp))
    s["mlp_head.weight"] = (cfg["num_classes"], dim)
    s["mlp_head.bias"] = (cfg["num_classes"],)
    return s


def ssf_trainable(name: str) -> bool:
    """freeze_vit=True (ssf.py:192-197): transformer / cls / conv / pos frozen, then every scale/shift re-enabled; head trains."""
    if "scale" in name or "shift" in name:
        return True
    return not ("transformer" in name or "cls_token" in name or "conv_proj" in name or "pos_embedding" in name)
