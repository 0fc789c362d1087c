"""ORACLE -- test infrastructure only (see oracle/vit_ref.py header for the rules and the pin).

CPU restatement of model/gaviko.py: masked-window local self-attention (MWSA), gated prompt
awakening (GPA = PRE gate + PCF balance + GXA/LXA cross-attention), the interleaved layer loop
and the prompt+CLS pooled head.  Every quirk in SURVEY.md Appendix C is reproduced on purpose.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .vit_ref import SD, Tensor, attention, attention_shapes, feed_forward, ff_shapes, layer_norm, mapping_vit, patch_embed


def window_mask(DHW, local_k, dtype=torch.float32) -> Tensor:
    """Additive 0/-inf mask [N,N] equal to gaviko.py:212-227, built from index arithmetic instead of
    the padded-volume slicing: query (d,h,w) sees key (d',h',w') iff  d - dk//2 <= d' < d - dk//2 + dk
    (same for h, w), clipped to the grid."""
    D, H, W = DHW
    dk, hk, wk = local_k
    d = torch.arange(D).view(D, 1, 1).expand(D, H, W).reshape(-1)
    h = torch.arange(H).view(1, H, 1).expand(D, H, W).reshape(-1)
    w = torch.arange(W).view(1, 1, W).expand(D, H, W).reshape(-1)

    def ok(q, k, kk):
        lo = q[:, None] - kk // 2
        return (k[None, :] >= lo) & (k[None, :] < lo + kk)

    allowed = ok(d, d, dk) & ok(h, h, hk) & ok(w, w, wk)
    m = torch.full((D * H * W, D * H * W), float("-inf"), dtype=dtype)
    m[allowed] = 0.0
    return m


def local_self_attention(sd: SD, prefix: str, x: Tensor, mask: Optional[Tensor], taps=None, drop_attn: Optional[Tensor] = None,
                         drop_proj: Optional[Tensor] = None) -> Tensor:
    """LocalSelfAttention.forward (gaviko.py:229-244).  Single head over the local_dim latent;
    scale is dim**-0.5 of the *model* dim (gaviko.py:201), qkv has no bias (205, 272).
    attn_drop / proj_drop (gaviko.py:238,242) are identity unless explicit scale masks (0 or 1/(1-p), what nn.Dropout multiplies by) are
    handed in: torch's RNG stream cannot be matched, so a test runs this with the masks the kernels drew (tests/dropmask.py)."""
    c = x.shape[-1]
    lat = F.linear(layer_norm(sd, prefix + ".norm", x), sd[prefix + ".proj_down.weight"], sd[prefix + ".proj_down.bias"])
    q, k, v = F.linear(lat, sd[prefix + ".qkv.weight"]).chunk(3, dim=-1)
    attn = q @ k.transpose(-2, -1) * (c ** -0.5)
    if mask is not None:
        attn = attn + mask.unsqueeze(0)
    attn = attn.softmax(dim=-1)
    if drop_attn is not None:
        attn = attn * drop_attn
    ctx = attn @ v
    if taps is not None:
        taps[prefix + ".ctx"] = ctx
    out = F.linear(ctx, sd[prefix + ".proj_up.weight"], sd[prefix + ".proj_up.bias"])
    return out if drop_proj is None else out * drop_proj


def _quick_gelu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(1.702 * x)  # gaviko.py:15-17


def _cross_attention(q: Tensor, tokens: Tensor) -> Tensor:
    """BaseFusionAttention.forward (gaviko.py:84-94): scale = latent_dim**-0.5."""
    w = torch.einsum("bpd,bnd->bpn", q, tokens) * (q.shape[-1] ** -0.5)
    return torch.einsum("bpn,bnd->bpd", w.softmax(dim=-1), tokens)


def awakening_prompt(sd: SD, prefix: str, x: Tensor, local_tokens: Tensor, num_prompts: int, taps=None) -> Tensor:
    """Awakening_Prompt.forward (gaviko.py:149-187)."""
    P = num_prompts
    wd, bd = sd[prefix + ".proj_down.0.weight"], sd[prefix + ".proj_down.0.bias"]
    x_lat = _quick_gelu(F.linear(x, wd, bd))                   # 155 (shared proj_down for both streams)
    l_lat = _quick_gelu(F.linear(local_tokens, wd, bd))        # 156
    prompts, cls, img = x_lat[:, :P], x_lat[:, P:P + 1], x_lat[:, P + 1:]   # 159-161

    a = prefix + ".cls_analyzer.cls_analyzer_"                 # PRE gate, gaviko.py:23-29
    h = F.layer_norm(cls, (cls.shape[-1],), sd[a + ".0.weight"], sd[a + ".0.bias"])
    h = F.gelu(F.linear(h, sd[a + ".1.weight"], sd[a + ".1.bias"]))
    importance = torch.sigmoid(F.linear(h, sd[a + ".3.weight"], sd[a + ".3.bias"]))  # [B,1,P]

    g = prefix + ".gl_balancer.gl_balancer_"                   # PCF weight, gaviko.py:51-55
    gw = F.layer_norm(cls, (cls.shape[-1],), sd[g + ".0.weight"], sd[g + ".0.bias"])
    gw = torch.sigmoid(F.linear(gw, sd[g + ".1.weight"], sd[g + ".1.bias"]))          # [B,1,1]

    # GXA: forward() is handed the already-sliced image latents (170) and get_tokens slices
    # [:, P+1:] AGAIN (106-107)  ->  only image tokens P+1.. of the image block take part.
    qg = F.linear(prompts, sd[prefix + ".global_attention.query_proj.weight"], sd[prefix + ".global_attention.query_proj.bias"])
    g_ctx = _cross_attention(qg, img[:, P + 1:])
    ql = F.linear(prompts, sd[prefix + ".local_attention.query_proj.weight"], sd[prefix + ".local_attention.query_proj.bias"])
    l_ctx = _cross_attention(ql, l_lat)                        # 118-119, 172: all local latents

    fused = gw * g_ctx + (1 - gw) * l_ctx                      # 175
    enhanced = fused * importance.transpose(1, 2)              # 178
    combined = torch.cat([enhanced, cls, img], dim=1)          # 181-185
    if taps is not None:
        taps[prefix + ".combined"] = combined
        taps[prefix + ".importance"] = importance
        taps[prefix + ".gw"] = gw
    return F.linear(combined, sd[prefix + ".proj_up.weight"], sd[prefix + ".proj_up.bias"])   # 187


def gaviko_forward(sd: SD, img: Tensor, cfg: dict, taps: Optional[dict] = None) -> Tensor:
    """Gaviko.forward (gaviko.py:531-552) + Transformer.forward (291-306) + AdaptiveFusionHead (314-316)."""
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    P = cfg["num_prompts"]
    share = cfg.get("share_factor", 1)
    patch = (cfg["frame_patch_size"], cfg["image_patch_size"], cfg["image_patch_size"])
    x = patch_embed(sd, "conv_proj.0", img, patch)             # 532-533
    b = x.shape[0]
    g = torch.cat([sd["prompt_embeddings"].expand(b, -1, -1), sd["cls_token"].expand(b, -1, -1), x], dim=1)
    g = g + torch.cat([sd["prompt_positional_embedding"], sd["pos_embedding"]], dim=1)   # 540-543
    loc = x + sd["pos_embedding"][:, 1:, :]                    # 546-547
    emasks = cfg.get("_masks") or {}                           # explicit dropout masks (tests).  self.dropout is applied TWICE, with independent
    if ("emb", 0) in emasks:                                   # draws: to the global tokens (544) and to the local tokens (548); live only
        g = g * emasks[("emb", 0)]                             # with freeze_vit=False (gaviko.py:513-528)
    if ("emb_local", 0) in emasks:
        loc = loc * emasks[("emb_local", 0)]
    mask = None
    if cfg.get("DHW") is not None:
        mask = window_mask(tuple(cfg["DHW"]), tuple(cfg["local_k"]), dtype=x.dtype)
    if taps is not None:
        taps["embed.global"] = g
        taps["embed.local"] = loc
    for i in range(depth):
        s = i // share                                         # 299
        masks = cfg.get("_masks") or {}
        loc = local_self_attention(sd, f"transformer.local_attns.{s}", loc, mask, taps, masks.get(("mwsa_attn", i)),
                                   masks.get(("mwsa_proj", i))) + loc                                # 301
        g = attention(sd, f"transformer.attns.{i}", g, heads, None, masks=masks, layer=i) + g        # 302
        if taps is not None:
            taps[f"layer{i}.local"] = loc
            taps[f"layer{i}.post_attn"] = g
        prompt = awakening_prompt(sd, f"transformer.prompt_projs.{s}", g, loc, P, None)              # 303
        g = feed_forward(sd, f"transformer.mlps.{i}", g, masks=masks, layer=i) + g + prompt          # 304
        if taps is not None:
            taps[f"layer{i}.gpa"] = prompt
            taps[f"layer{i}.post_mlp"] = g
    out = layer_norm(sd, "transformer.norm", g)                # 306
    if taps is not None:
        taps["final_norm"] = out
    pooled = out[:, 0:P + 1].mean(dim=1)                       # 316: prompts + CLS
    return F.linear(pooled, sd["mlp_head.head.weight"], sd["mlp_head.head.bias"])


# --------------------------------------------------------------------------- schema / freeze rule
def gaviko_param_shapes(cfg: dict, with_alias: bool = False) -> Dict[str, tuple]:
    """SURVEY Appendix A 'Gaviko'.  named_parameters() order is not reproduced; names/shapes are."""
    depth, heads, dim, mlp = mapping_vit(cfg["backbone"])
    fp, ip = cfg["frame_patch_size"], cfg["image_patch_size"]
    n = (cfg["frames"] // fp) * (cfg["image_size"] // ip) ** 2
    P, l, ld = cfg["num_prompts"], cfg.get("prompt_latent_dim", 20), cfg.get("local_dim", 20)
    inner = heads * cfg.get("dim_head", 64)
    ch = cfg.get("channels", 1)
    s = {
        "pos_embedding": (1, n + 1, dim), "cls_token": (1, 1, dim),
        "prompt_positional_embedding": (1, P, dim), "prompt_embeddings": (1, P, dim),
        "conv_proj.0.weight": (dim, ch, fp, ip, ip), "conv_proj.0.bias": (dim,),
        "transformer.norm.weight": (dim,), "transformer.norm.bias": (dim,),
    }
    nshared = math.ceil(depth / cfg.get("share_factor", 1))
    for i in range(nshared):
        p = f"transformer.local_attns.{i}"
        s.update({p + ".norm.weight": (dim,), p + ".norm.bias": (dim,),
                  p + ".proj_down.weight": (ld, dim), p + ".proj_down.bias": (ld,),
                  p + ".qkv.weight": (3 * ld, ld),
                  p + ".proj_up.weight": (dim, ld), p + ".proj_up.bias": (dim,)})
        p = f"transformer.prompt_projs.{i}"
        s.update({p + ".proj_down.0.weight": (l, dim), p + ".proj_down.0.bias": (l,),
                  p + ".proj_up.weight": (dim, l), p + ".proj_up.bias": (dim,),
                  p + ".cls_analyzer.cls_analyzer_.0.weight": (l,), p + ".cls_analyzer.cls_analyzer_.0.bias": (l,),
                  p + ".cls_analyzer.cls_analyzer_.1.weight": (64, l), p + ".cls_analyzer.cls_analyzer_.1.bias": (64,),
                  p + ".cls_analyzer.cls_analyzer_.3.weight": (P, 64), p + ".cls_analyzer.cls_analyzer_.3.bias": (P,),
                  p + ".gl_balancer.gl_balancer_.0.weight": (l,), p + ".gl_balancer.gl_balancer_.0.bias": (l,),
                  p + ".gl_balancer.gl_balancer_.1.weight": (1, l), p + ".gl_balancer.gl_balancer_.1.bias": (1,),
                  p + ".global_attention.query_proj.weight": (l, l), p + ".global_attention.query_proj.bias": (l,),
                  p + ".local_attention.query_proj.weight": (l, l), p + ".local_attention.query_proj.bias": (l,)})
        if with_alias:  # state_dict() (not named_parameters()) also lists the aliases, gaviko.py:144-145
            s.update({p + ".global_query.weight": (l, l), p + ".global_query.bias": (l,),
                      p + ".local_query.weight": (l, l), p + ".local_query.bias": (l,)})
    for i in range(depth):
        s.update(attention_shapes(f"transformer.attns.{i}", dim, inner))
        s.update(ff_shapes(f"transformer.mlps.{i}", dim, mlp))
    s["mlp_head.head.weight"] = (cfg["num_classes"], dim)
    s["mlp_head.head.bias"] = (cfg["num_classes"],)
    return s


def gaviko_trainable(name: str) -> bool:
    """Freeze rule gaviko.py:429-434 (freeze_vit=True): freeze transformer|cls_token|conv_proj|pos_embedding,
    then re-enable head|prompt|local_attn -- applied in that order to every name."""
    rg = True
    if "transformer" in name or "cls_token" in name or "conv_proj" in name or "pos_embedding" in name:
        rg = False
    if "head" in name or "prompt" in name or "local_attn" in name:
        rg = True
    return rg
