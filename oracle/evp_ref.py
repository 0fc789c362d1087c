"""ORACLE -- test infrastructure only (see oracle/vit_ref.py header).

CPU restatement of the EVP plugin (model/evp.py, `--method evp`, Explicit Visual Prompting): per-layer prompts, generated from
a high-pass filtered copy of the volume and from the patch embeddings, are added to the patch tokens in front of every layer.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .vit_ref import SD, Tensor, attention, attention_shapes, feed_forward, ff_shapes, layer_norm, mapping_vit


def fft_highpass(x: Tensor, rate: float) -> Tensor:
    """PromptGenerator.fft (evp.py:126-147) AS IT EXECUTES on a 5-D volume [B, C, D, H, W]: fft2 / ifft2 run over the last two
    dims (H, W), but the mask line `mask[:, :, w//2-line:w//2+line, h//2-line:h//2+line] = 1` indexes dims 2 and 3 -- DEPTH and H --
    with a half-side computed from (H, W).  So (quirk 17): only the depth slices w//2-line <= d < w//2+line are filtered at all, and
    there the zeroed band is a range of shifted H-frequencies for EVERY W-frequency (a 1-D high-pass along H); the other slices
    pass through unchanged (|x|).  Reproduced on purpose."""
    w, h = x.shape[-2:]
    line = int((w * h * rate) ** 0.5 // 2)
    mask = torch.zeros_like(x)
    mask[:, :, w // 2 - line: w // 2 + line, h // 2 - line: h // 2 + line] = 1
    f = torch.fft.fftshift(torch.fft.fft2(x, norm="forward")) * (1 - mask)
    return torch.fft.ifft2(torch.fft.ifftshift(f), norm="forward").real.abs()


def evp_forward(sd: SD, img: Tensor, cfg: dict, taps: Optional[dict] = None) -> Tensor:
    """ExplicitVisualPrompting.forward (evp.py:344-373), PromptGenerator.init_embeddings / init_handcrafted / get_prompt (76-95) and
    Transformer.forward (233-241: prompt[i] is added to rows 1.. in front of layer i)."""
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    patch = (cfg["frame_patch_size"], cfg["image_patch_size"], cfg["image_patch_size"])
    pg = "prompt_generator."
    xc = F.conv3d(img, sd["conv_proj.proj.weight"], sd["conv_proj.proj.bias"], stride=patch)           # [B, C, d, h, w]
    b, c = xc.shape[:2]
    tok = xc.view(b, c, -1).permute(0, 2, 1)                                                          # [B, N, C]
    emb = F.linear(tok, sd[pg + "embedding_generator.weight"], sd[pg + "embedding_generator.bias"])
    hp = fft_highpass(img, cfg.get("freq_nums", 0.25))
    hc = F.conv3d(hp, sd[pg + "prompt_generator.proj.weight"], sd[pg + "prompt_generator.proj.bias"], stride=patch)
    hc = hc.view(b, hc.shape[1], -1).permute(0, 2, 1)
    s = hc + emb
    x = torch.cat((sd["cls_token"].expand(b, -1, -1), tok), dim=1) + sd["pos_embedding"][:, : tok.shape[1] + 1]
    masks = cfg.get("_masks")                       # explicit dropout masks (tests; live only with freeze_vit=False: evp.py:333-344)
    if masks is not None and ("emb", 0) in masks:
        x = x * masks[("emb", 0)]                   # self.dropout, evp.py:365
    for i in range(depth):
        u = F.gelu(F.linear(s, sd[pg + f"lightweight_mlp_{i}.0.weight"], sd[pg + f"lightweight_mlp_{i}.0.bias"]))
        prompt = F.linear(u, sd[pg + "shared_mlp.weight"], sd[pg + "shared_mlp.bias"])
        x = torch.cat((x[:, :1], prompt + x[:, 1:]), dim=1)
        p = f"transformer.layers.{i}"
        x = attention(sd, p + ".0", x, heads, masks=masks, layer=i) + x
        f = feed_forward(sd, p + ".1", x, masks=masks, layer=i)
        x = f + x
        if taps is not None:
            taps[f"layer{i}.ff_out"] = f
            taps[f"layer{i}.post_mlp"] = x
    x = layer_norm(sd, "transformer.norm", x)
    if taps is not None:
        taps["final_norm"] = x
    x = x.mean(dim=1) if cfg.get("pool", "cls") == "mean" else x[:, 0]
    return F.linear(x, sd["mlp_head.weight"], sd["mlp_head.bias"])


def evp_param_shapes(cfg: dict) -> Dict[str, tuple]:
    """state_dict of ExplicitVisualPrompting in registration order (evp.py:292-319, 41-54); note conv_proj is a PatchEmbed
    (key conv_proj.proj.*, not conv_proj.0.*)."""
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    fp, ip = cfg["frame_patch_size"], cfg["image_patch_size"]
    n = (cfg["frames"] // fp) * (cfg["image_size"] // ip) ** 2
    inner = heads * cfg.get("dim_head", 64)
    ch = cfg.get("channels", 3)
    r = dim // cfg.get("scale_factor", 32)
    s = {"pos_embedding": (1, n + 1, dim), "cls_token": (1, 1, dim),
         "conv_proj.proj.weight": (dim, ch, fp, ip, ip), "conv_proj.proj.bias": (dim,),
         "transformer.norm.weight": (dim,), "transformer.norm.bias": (dim,)}
    for i in range(depth):
        s.update(attention_shapes(f"transformer.layers.{i}.0", dim, inner))
        s.update(ff_shapes(f"transformer.layers.{i}.1", dim, mlp))
    s["mlp_head.weight"] = (cfg["num_classes"], dim)
    s["mlp_head.bias"] = (cfg["num_classes"],)
    pg = "prompt_generator."
    s.update({pg + "shared_mlp.weight": (dim, r), pg + "shared_mlp.bias": (dim,),
              pg + "embedding_generator.weight": (r, dim), pg + "embedding_generator.bias": (r,)})
    for i in range(depth):
        s.update({pg + f"lightweight_mlp_{i}.0.weight": (r, r), pg + f"lightweight_mlp_{i}.0.bias": (r,)})
    s.update({pg + "prompt_generator.proj.weight": (r, ch, fp, ip, ip), pg + "prompt_generator.proj.bias": (r,)})
    return s


def evp_trainable(name: str) -> bool:
    """freeze_vit=True (evp.py:322-327): transformer / cls / conv_proj / pos frozen, prompt_generator re-enabled; the head was never frozen."""
    if "prompt_generator" in name:
        return True
    return not ("transformer" in name or "cls_token" in name or "conv_proj" in name or "pos_embedding" in name)
