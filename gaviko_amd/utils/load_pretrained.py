"""Backbone table of the reference (utils/load_pretrained.py:103-120).  The timm download / key remap of the same
file is outside the hot path (needs network) and is not rebuilt here."""
from __future__ import annotations

VIT_CONFIGS = {
    "vit-b16": dict(depth=12, heads=12, dim=768, mlp_dim=3072),
    "vit-t16": dict(depth=12, heads=3, dim=192, mlp_dim=768),
    "vit-s16": dict(depth=12, heads=6, dim=384, mlp_dim=1536),
    "vit-l16": dict(depth=24, heads=16, dim=1024, mlp_dim=4096),
}


def mapping_vit(backbone):
    """backbone name -> (depth, heads, dim, mlp_dim); same names, same errors as the reference."""
    if backbone is None:
        raise ValueError("Backbone must be specified.")
    cfg = VIT_CONFIGS.get(backbone.lower())
    if cfg is None:
        raise ValueError(f"Unsupported backbone: {backbone}. Supported backbones are: {list(VIT_CONFIGS.keys())}")
    return cfg["depth"], cfg["heads"], cfg["dim"], cfg["mlp_dim"]
