"""ORACLE -- test infrastructure only (see oracle/vit_ref.py header).

CPU restatements of the other --method plugins that ride the same ViT kernels:
VPT shallow/deep (model/vpt.py), AdaptFormer (model/adaptformer.py), MeLO/LoRA (model/melo.py).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .vit_ref import (SD, Tensor, attention, attention_shapes, embed_tokens, feed_forward, ff_shapes,
                      layer_norm, mapping_vit, vit_forward, vit_param_shapes)


def vpt_forward(sd: SD, img: Tensor, cfg: dict, taps: Optional[dict] = None) -> Tensor:
    """PromptedVisionTransformer.forward (vpt.py:163-177): tokens [cls | P prompts | patches]
    (127-131); deep (vpt.py:142-153): before every layer i>0 the rows after CLS are cut at
    1 + deep_prompt_embeddings[i].shape[1] -- and since deep_prompt_embeddings[i] is [P, prompt_dim] that is
    1 + prompt_dim (64), NOT 1 + P: each deep layer drops the previous prompts AND the first prompt_dim-P patch
    tokens, so the sequence shrinks by prompt_dim-P per layer (1009 -> 953 -> ... -> 393 at P=8, prompt_dim=64).
    Reproduced on purpose (quirk 16).  Head reads CLS (row 0).  prompt_dropout is identity (p=0)."""
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    patch = (cfg["frame_patch_size"], cfg["image_patch_size"], cfg["image_patch_size"])
    pre = "vision_transformer."
    x = embed_tokens(sd, img, patch, pre)
    b = x.shape[0]
    deep = cfg.get("deep_prompt", True)
    P = cfg.get("num_prompts", 8)

    masks = cfg.get("_masks")                       # explicit dropout masks (tests): {('prompt', layer): [b, P, dim]}; with freeze_vit=False
    if masks is not None and ("emb", 0) in masks:   # also the backbone's own: ('emb', 0) over [cls | patches] (vpt.py:171, before the prompts
        x = x * masks[("emb", 0)]                   # go in) and the per-layer attention / MLP sites of vit_ref

    def proj(e, i):
        y = F.linear(e, sd["prompt_proj.weight"], sd["prompt_proj.bias"]).expand(b, -1, -1)
        if masks is not None and ("prompt", i) in masks:
            y = y * masks[("prompt", i)]
        return y

    if not deep:
        x = torch.cat((x[:, :1], proj(sd["prompt_embeddings"], 0), x[:, 1:]), dim=1)
    for i in range(depth):
        if deep:
            keep = x[:, 1:] if i == 0 else x[:, 1 + sd["deep_prompt_embeddings"].shape[2]:]
            x = torch.cat((x[:, :1], proj(sd["deep_prompt_embeddings"][i], i), keep), dim=1)
        p = f"{pre}transformer.layers.{i}"
        x = attention(sd, p + ".0", x, heads, masks=masks, layer=i) + x
        f = feed_forward(sd, p + ".1", x, masks=masks, layer=i)
        x = f + x
        if taps is not None:
            taps[f"layer{i}.ff_out"] = f
            taps[f"layer{i}.post_mlp"] = x
    x = layer_norm(sd, pre + "transformer.norm", x)
    if taps is not None:
        taps["final_norm"] = x
    x = x.mean(dim=1) if cfg.get("pool", "cls") == "mean" else x[:, 0]
    return F.linear(x, sd[pre + "mlp_head.weight"], sd[pre + "mlp_head.bias"])


def vpt_param_shapes(cfg: dict) -> Dict[str, tuple]:
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    s = vit_param_shapes(cfg, "vision_transformer.")
    pd, P = cfg.get("prompt_dim", 64), cfg.get("num_prompts", 8)
    s["prompt_proj.weight"] = (dim, pd)
    s["prompt_proj.bias"] = (dim,)
    if cfg.get("deep_prompt", True):
        s["deep_prompt_embeddings"] = (depth, P, pd)
    else:
        s["prompt_embeddings"] = (1, P, pd)
    return s


def vpt_trainable(name: str) -> bool:
    """vpt.py:91-94: inside vision_transformer.*, names with transformer|cls_token|conv_proj|pos_embedding
    are frozen; prompt_proj / (deep_)prompt_embeddings / vision_transformer.mlp_head stay trainable."""
    if name.startswith("vision_transformer."):
        k = name[len("vision_transformer."):]
        return not ("transformer" in k or "cls_token" in k or "conv_proj" in k or "pos_embedding" in k)
    return True


def adapter(sd: SD, prefix: str, x: Tensor, scale: float = 1.0) -> Tensor:
    """Adapter.forward with layernorm_option='in', add_residual=False (adaptformer.py:58-78):
    up(ReLU(down(LN(x)))) * scale; dropout p=0.0."""
    h = layer_norm(sd, prefix + ".adapter_layer_norm_before", x)
    h = F.relu(F.linear(h, sd[prefix + ".down_adapter_proj.weight"], sd[prefix + ".down_adapter_proj.bias"]))
    return F.linear(h, sd[prefix + ".up_adapter_proj.weight"], sd[prefix + ".up_adapter_proj.bias"]) * scale


def adaptformer_forward(sd: SD, img: Tensor, cfg: dict, taps: Optional[dict] = None) -> Tensor:
    """AdaptFormer.forward (adaptformer.py:194-209) + Transformer.forward (93-99):
    x = attn(x)+x ; r = adapter(x) ; x = ff(x)+x+r."""
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    patch = (cfg["frame_patch_size"], cfg["image_patch_size"], cfg["image_patch_size"])
    masks = cfg.get("_masks")                       # explicit dropout masks (tests; live only with freeze_vit=False: adaptformer.py:175-191)
    x = embed_tokens(sd, img, patch)
    if masks is not None and ("emb", 0) in masks:
        x = x * masks[("emb", 0)]                   # self.dropout, adaptformer.py:203
    for i in range(depth):
        p = f"transformer.layers.{i}"
        x = attention(sd, p + ".0", x, heads, masks=masks, layer=i) + x
        r = adapter(sd, p + ".1", x)                # Adapter(dim): its own dropout p = 0.0 (adaptformer.py:23-27,67)
        f = feed_forward(sd, p + ".2", x, masks=masks, layer=i)
        x = f + x + r
        if taps is not None:
            taps[f"layer{i}.ff_out"] = f
            taps[f"layer{i}.post_mlp"] = x
    x = layer_norm(sd, "transformer.norm", x)
    if taps is not None:
        taps["final_norm"] = x
    x = x.mean(dim=1) if cfg.get("pool", "cls") == "mean" else x[:, 0]
    return F.linear(x, sd["mlp_head.weight"], sd["mlp_head.bias"])


def adaptformer_param_shapes(cfg: dict, down_dim: int = 64) -> Dict[str, tuple]:
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    s = vit_param_shapes(cfg, "", block_names=("0", "2"))
    for i in range(depth):
        p = f"transformer.layers.{i}.1"
        s.update({p + ".adapter_layer_norm_before.weight": (dim,), p + ".adapter_layer_norm_before.bias": (dim,),
                  p + ".down_adapter_proj.weight": (down_dim, dim), p + ".down_adapter_proj.bias": (down_dim,),
                  p + ".up_adapter_proj.weight": (dim, down_dim), p + ".up_adapter_proj.bias": (dim,)})
    return s


def adaptformer_trainable(name: str) -> bool:
    """adaptformer.py:163-168 (freeze_vit=True)."""
    rg = True
    if "transformer" in name or "cls_token" in name or "conv_proj" in name or "pos_embedding" in name:
        rg = False
    if "adapter" in name or "head" in name:
        rg = True
    return rg


def melo_forward(sd: SD, img: Tensor, cfg: dict, taps: Optional[dict] = None) -> Tensor:
    """MeLO.forward (melo.py:100-101) == the wrapped VisionTransformer with _LoRA_qkv_timm (41-47)
    in every attention; integer scale alpha // r."""
    return vit_forward(sd, img, cfg, taps, prefix="lora_vit.", lora={"r": cfg["r"], "alpha": cfg["alpha"], "layers": cfg.get("lora_layer") or None})


def melo_param_shapes(cfg: dict) -> Dict[str, tuple]:
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    base = vit_param_shapes(cfg, "lora_vit.")
    s = {}
    wrapped = cfg.get("lora_layer") or list(range(depth))          # melo.py:53-56,67-68
    for k, v in base.items():
        if k.endswith(".to_qkv.weight") and int(k.split(".layers.")[1].split(".")[0]) in wrapped:
            p = k[: -len(".weight")]
            s[p + ".qkv.weight"] = v
            s[p + ".linear_a_q.weight"] = (cfg["r"], dim)
            s[p + ".linear_b_q.weight"] = (dim, cfg["r"])
            s[p + ".linear_a_v.weight"] = (cfg["r"], dim)
            s[p + ".linear_b_v.weight"] = (dim, cfg["r"])
        else:
            s[k] = v
    return s


def melo_trainable(name: str) -> bool:
    """melo.py:63-65,90-91: the whole ViT is frozen; LoRA A/B and the re-created head train."""
    return ".linear_a_" in name or ".linear_b_" in name or "mlp_head" in name
