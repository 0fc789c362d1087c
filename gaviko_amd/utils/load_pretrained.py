"""Pretrained-weight and checkpoint interchange with the reference (utils/load_pretrained.py, train.py:460-485).

Host-side, offline, outside the hot path.  Same function names and argument meaning as the reference module, with one
difference forced by the environment: there is no network, so `load_pretrain` never calls `timm.create_model(pretrained=True)`
-- it reads the raw timm state dict that the reference itself leaves behind in `save_dir` (load_pretrained.py:26-28 saves
`model.state_dict()` to `./pretrained/<timm model name>`), and fails loudly when that file is not there.

What is reproduced exactly (fixtures `tests/golden/pretrain_convert*.npz` come from the reference function):
* key remap 63-98: `blocks.i.norm1/attn.qkv/attn.proj` -> `transformer.attns.i.norm/to_qkv/to_out.0`, `blocks.i.norm2/mlp.fc1/
  mlp.fc2` -> `transformer.mlps.i.net.0/net.1/net.4`, `patch_embed.proj` -> `conv_proj.0`, `pos_embed` -> `pos_embedding`,
  `norm` -> `transformer.norm`, `cls_token`; everything else (head, pre_logits, ...) is dropped.  timm's `attn.qkv.bias`
  is carried over as `transformer.attns.i.to_qkv.bias` although `to_qkv` is bias-free: it has no destination and
  `load_state_dict(strict=False)` skips it, as in the reference;
* only `Gaviko` names its blocks `transformer.attns/mlps` (gaviko.py:281-289); every other class uses `transformer.layers.i.j`,
  so with `strict=False` their block weights are silently NOT loaded.  `remap_blocks_to_layers` is the fix a
  user can opt into; the default stays bug-compatible;
* position embedding 34-44: cls row kept, the 14x14 grid reshaped to [1, C, 1, 14, 14] and trilinearly resampled
  (align_corners=False) to round(num_patches^(1/3))^3 positions;
* patch kernel 46-51: mean over the RGB input channels, repeated `depth_dim` times along the new depth axis (no 1/depth scale).
"""
from __future__ import annotations

import logging
import os
from typing import Dict, Optional

import torch
import torch.nn.functional as F

VIT_CONFIGS = {
    "vit-b16": dict(depth=12, heads=12, dim=768, mlp_dim=3072),
    "vit-t16": dict(depth=12, heads=3, dim=192, mlp_dim=768),
    "vit-s16": dict(depth=12, heads=6, dim=384, mlp_dim=1536),
    "vit-l16": dict(depth=24, heads=16, dim=1024, mlp_dim=4096),
}

TIMM_NAMES = {                                   # load_pretrained.py:14-21
    "vit-b16": "vit_base_patch16_224_in21k",
    "vit-t16": "vit_tiny_patch16_224_in21k",
    "vit-s16": "vit_small_patch16_224_in21k",
    "vit-l16": "vit_large_patch16_224_in21k",
}


def mapping_vit(backbone):
    """backbone name -> (depth, heads, dim, mlp_dim); same names, same errors as the reference (load_pretrained.py:103-120)."""
    if backbone is None:
        raise ValueError("Backbone must be specified.")
    cfg = VIT_CONFIGS.get(backbone.lower())
    if cfg is None:
        raise ValueError(f"Unsupported backbone: {backbone}. Supported backbones are: {list(VIT_CONFIGS.keys())}")
    return cfg["depth"], cfg["heads"], cfg["dim"], cfg["mlp_dim"]


def interpolate_pos_embedding(pre_pos_embed: torch.Tensor, num_patches: int) -> torch.Tensor:
    """[1, 1 + g*g, C] -> [1, 1 + n^3, C], n = round(num_patches^(1/3)) (load_pretrained.py:34-44)."""
    cls_token, grid = pre_pos_embed[:, :1, :], pre_pos_embed[:, 1:, :]
    g = int(grid.shape[1] ** 0.5)
    grid = grid.reshape(1, g, g, -1).permute(0, 3, 1, 2).unsqueeze(2)                       # [1, C, 1, g, g]
    n = round(num_patches ** (1 / 3))
    grid = F.interpolate(grid, size=(n, n, n), mode="trilinear", align_corners=False)       # [1, C, n, n, n]
    grid = grid.permute(0, 2, 3, 4, 1).reshape(1, n * n * n, -1)
    return torch.cat([cls_token, grid], dim=1)


def mean_kernel(patch_emb_weight: torch.Tensor, depth_dim: int) -> torch.Tensor:
    """[C, 3, p, p] -> [C, 1, depth_dim, p, p]: RGB mean, repeated along depth (load_pretrained.py:46-51)."""
    w = patch_emb_weight.mean(dim=1, keepdim=True)
    return w.unsqueeze(2).repeat(1, 1, depth_dim, 1, 1)


def convert_timm_state_dict(timm_dict: Dict[str, torch.Tensor], num_patches: int, depth_dim: int) -> Dict[str, torch.Tensor]:
    """The key remap + tensor surgery of load_pretrained.py:29-99 on an already loaded timm ViT state dict."""
    new_dict: Dict[str, torch.Tensor] = {}
    block_rules = (                                 # first match wins, in the reference's order (63-84)
        ("norm1", "norm", "transformer.attns"), ("attn.qkv", "to_qkv", "transformer.attns"), ("attn.proj", "to_out.0", "transformer.attns"),
        ("norm2", "net.0", "transformer.mlps"), ("mlp.fc1", "net.1", "transformer.mlps"), ("mlp.fc2", "net.4", "transformer.mlps"))
    for key, value in timm_dict.items():
        if key == "cls_token":
            new_dict[key] = value
            continue
        for old, new, root in block_rules:
            if old in key:
                new_dict[key.replace(old, new).replace("blocks", root)] = value
                break
        else:
            if "patch_embed.proj.weight" in key:
                new_dict[key.replace("patch_embed.proj.weight", "conv_proj.0.weight").replace("blocks", "transformer")] = mean_kernel(value, depth_dim)
            elif "patch_embed.proj.bias" in key:
                new_dict[key.replace("patch_embed.proj.bias", "conv_proj.0.bias").replace("blocks", "transformer")] = value
            elif key == "pos_embed":
                new_dict["pos_embedding"] = interpolate_pos_embedding(value, num_patches)
            elif key == "norm.weight":
                new_dict["transformer.norm.weight"] = value
            elif key == "norm.bias":
                new_dict["transformer.norm.bias"] = value
    return new_dict


def remap_blocks_to_layers(converted: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """`transformer.attns.i.X` -> `transformer.layers.i.0.X`, `transformer.mlps.i.X` -> `transformer.layers.i.1.X`: the names of every
    class except Gaviko (vision_transformer.py:78-83).  NOT applied by default -- the reference drops these weights."""
    out = {}
    for k, v in converted.items():
        for root, j in (("transformer.attns.", 0), ("transformer.mlps.", 1)):
            if k.startswith(root):
                i, rest = k[len(root):].split(".", 1)
                k = f"transformer.layers.{i}.{j}.{rest}"
                break
        out[k] = v
    return out


def pretrained_path(backbone: str, save_dir: str) -> Optional[str]:
    name = TIMM_NAMES.get(backbone.replace("_", "-").lower())
    return None if name is None else os.path.join(save_dir, name)


def load_pretrain(backbone, num_patches, depth_dim, save_dir):
    """load_pretrained.py:8-99 without the download: reads `save_dir/<timm model name>` (the file the reference writes at 26-28)."""
    backbone = backbone.replace("_", "-")
    path = pretrained_path(backbone, save_dir)
    if path is None:
        logging.info("Warning: The model initizalizes without pretrained knowledge!")
        raise ValueError(f"Unsupported backbone: {backbone}. Supported backbones are: {list(TIMM_NAMES.keys())}")
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: this build has no network access and does not call timm.  Run the reference once (it saves the raw "
            f"timm state dict there, load_pretrained.py:26-28) or torch.save(timm_model.state_dict(), {path!r}) yourself.")
    jax_dict = torch.load(path, map_location="cpu")
    return convert_timm_state_dict(jax_dict, num_patches, depth_dim)


def load_vanilla_pretrain(backbone, config, save_dir="./pretrained"):
    """load_pretrained.py:124-147: patch geometry from config['model'] -> converted backbone dict."""
    m = config["model"]
    pair = lambda t: t if isinstance(t, tuple) else (t, t)  # noqa: E731
    ih, iw = pair(m["image_size"])
    ph, pw = pair(m["image_patch_size"])
    assert ih % ph == 0 and iw % pw == 0, "Image dimensions must be divisible by the patch size."
    assert m["frames"] % m["frame_patch_size"] == 0, "Frames must be divisible by frame patch size"
    num_patches = (ih // ph) * (iw // pw) * (m["frames"] // m["frame_patch_size"])
    return load_pretrain(backbone, num_patches, m["frame_patch_size"], save_dir=save_dir)


def load_vanilla_pretrain_with_adapters(backbone, config, checkpoint_path, save_dir="./pretrained"):
    """load_pretrained.py:150-156 (eval.py:91, inference.py:90): converted backbone overlaid with a trainable-only checkpoint."""
    base = load_vanilla_pretrain(backbone, config, save_dir=save_dir)
    checkpoint = torch.load(checkpoint_path, map_location="cpu")
    return {**base, **checkpoint}


# ---- trainable-only checkpoints (train.py:460-485) -------------------------------------------------------------------------
def checkpoint_name(method: str, backbone: str, epoch: int, val_acc: float) -> str:
    """train.py:466-469: '<method>_<backbone with _>_best_model_epoch<E>_acc<A:.4f>.pt'."""
    return f"{method}_{backbone.replace('-', '_')}_best_model_epoch{epoch}_acc{val_acc:.4f}.pt"


def tuning_param_names(model) -> list:
    """train.py:160-164: the names train.py collects as `tuning_params` (requires_grad parameters, module order)."""
    return [n for n, p in model.named_parameters() if p.requires_grad]


def save_trainable(model, save_dir: str, method: str, backbone: str, epoch: int, val_acc: float, tuning_params=None) -> str:
    """train.py:464-483: state_dict filtered to the trainable names, torch.save'd under save_dir/experiments/<method>/."""
    names = set(tuning_params if tuning_params is not None else tuning_param_names(model))
    out_dir = os.path.join(save_dir, "experiments", method)
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, checkpoint_name(method, backbone, epoch, val_acc))
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items() if k in names}, path)
    return path
