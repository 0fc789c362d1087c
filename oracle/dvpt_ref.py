"""ORACLE -- test infrastructure only (see oracle/vit_ref.py header).

CPU restatement of the DVPT plugin (model/dvpt.py, `--method dvpt`): tokens [P prompts | cls | N patches]; after every attention
block a rank-20 "share_MLP" adapter lets the prompt latents cross-attend to the patch latents and adds the up-projected result,
times a (zero-initialised) scalar gate, to the MLP block's output.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .vit_ref import SD, Tensor, attention, attention_shapes, feed_forward, ff_shapes, layer_norm, mapping_vit, patch_embed


def share_mlp(sd: SD, p: str, x: Tensor, num: int) -> Tensor:
    """share_MLP.forward (dvpt.py:37-47): QuickGELU is applied to the INPUT, then C->20; prompts (rows 0..num) attend to the patch
    latents (rows num+1..) with scale d_model**-0.5; [attended prompts | cls latent | patch latents] -> 20->C, times prompt_gate."""
    z = F.linear(x * torch.sigmoid(1.702 * x), sd[p + ".prompt_key_proj_d.weight"], sd[p + ".prompt_key_proj_d.bias"])
    cls, prompt, tokens = z[:, num:num + 1], z[:, :num], z[:, num + 1:]
    attn = (prompt @ tokens.transpose(-2, -1) * (x.shape[-1] ** -0.5)).softmax(dim=-1)
    out = torch.cat([attn @ tokens, cls, tokens], dim=1)
    return F.linear(out, sd[p + ".prompt_key_proj_u.weight"], sd[p + ".prompt_key_proj_u.bias"]) * sd[p + ".prompt_gate"]


def dvpt_forward(sd: SD, img: Tensor, cfg: dict, taps: Optional[dict] = None) -> Tensor:
    """DynamicVisualPromptTuning.forward (dvpt.py:187-208) + ResidualAttentionBlock (59-63) + Transformer (78-83).
    pool='cls' reads row 0 of the normed sequence -- which is the FIRST PROMPT, not the cls token (the prompts come first);
    pool='mean' norms and averages rows 0..num (prompts + cls)."""
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    patch = (cfg["frame_patch_size"], cfg["image_patch_size"], cfg["image_patch_size"])
    num = cfg.get("num_prompts", 50)
    x = patch_embed(sd, "conv_proj.0", img, patch)
    b = x.shape[0]
    x = torch.cat((sd["cls_token"].expand(b, -1, -1), x), dim=1)
    x = torch.cat([sd["prompt_embeddings"].expand(b, -1, -1), x], dim=1)
    x = x + torch.cat([sd["prompt_positional_embedding"], sd["pos_embedding"]], dim=1)
    masks = cfg.get("_masks")                       # explicit dropout masks (tests; live only with freeze_vit=False: dvpt.py:168-184)
    if masks is not None and ("emb", 0) in masks:
        x = x * masks[("emb", 0)]                   # self.dropout, dvpt.py:200
    for i in range(depth):
        p = f"transformer.layers.{i}.0"
        x = attention(sd, p + ".attn", x, heads, masks=masks, layer=i) + x
        prompt = share_mlp(sd, p + ".prompt_proj", x, num)
        f = feed_forward(sd, p + ".mlp", x, masks=masks, layer=i)
        x = f + x + prompt
        if taps is not None:
            taps[f"layer{i}.ff_out"] = f
            taps[f"layer{i}.post_mlp"] = x
    if cfg.get("pool", "cls") == "cls":
        x = layer_norm(sd, "transformer.norm", x)
        if taps is not None:
            taps["final_norm"] = x
        x = x[:, 0]
    else:
        x = layer_norm(sd, "transformer.norm", x[:, : num + 1])
        if taps is not None:
            taps["final_norm"] = x
        x = x.mean(dim=1)
    return F.linear(x, sd["mlp_head.weight"], sd["mlp_head.bias"])


def dvpt_param_shapes(cfg: dict) -> Dict[str, tuple]:
    """state_dict of DynamicVisualPromptTuning in registration order (dvpt.py:128-146, 25-31, 50-57)."""
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    fp, ip = cfg["frame_patch_size"], cfg["image_patch_size"]
    n = (cfg["frames"] // fp) * (cfg["image_size"] // ip) ** 2
    inner = heads * cfg.get("dim_head", 64)
    num = cfg.get("num_prompts", 50)
    s = {"prompt_positional_embedding": (1, num, dim), "prompt_embeddings": (1, num, dim),
         "pos_embedding": (1, n + 1, dim), "cls_token": (1, 1, dim),
         "conv_proj.0.weight": (dim, cfg.get("channels", 3), fp, ip, ip), "conv_proj.0.bias": (dim,),
         "transformer.norm.weight": (dim,), "transformer.norm.bias": (dim,)}
    for i in range(depth):
        p = f"transformer.layers.{i}.0"
        s.update(attention_shapes(p + ".attn", dim, inner))
        s.update(ff_shapes(p + ".mlp", dim, mlp))
        s.update({p + ".prompt_proj.prompt_gate": (1,),
                  p + ".prompt_proj.prompt_key_proj_d.weight": (20, dim), p + ".prompt_proj.prompt_key_proj_d.bias": (20,),
                  p + ".prompt_proj.prompt_key_proj_u.weight": (dim, 20), p + ".prompt_proj.prompt_key_proj_u.bias": (dim,)})
    s["mlp_head.weight"] = (cfg["num_classes"], dim)
    s["mlp_head.bias"] = (cfg["num_classes"],)
    return s


def dvpt_trainable(name: str) -> bool:
    """freeze_vit=True (dvpt.py:158-163): everything under transformer / cls / conv / pos frozen, then 'prompt' or 'head' re-enabled."""
    if "prompt" in name or "head" in name:
        return True
    return not ("transformer" in name or "cls_token" in name or "conv_proj" in name or "pos_embedding" in name)
