"""Host-side mirror of the reference's model/ssf.py (`--method ssf`, Scaling & Shifting Features) for the MI355X path.

`ScalingShiftingFeatures` keeps the reference's kwargs (ssf.py:141-157), parameter / state_dict names (`ssf_scale_k` /
`ssf_shift_k` inside every Attention and FeedForward, `transformer.ssf_*_1` after the final norm, top-level `ssf_*_1` on the patch
tokens), freeze rule (ssf.py:192-197) and `train()` override (205-217, returns None).  The sub-modules are parameter
containers; the arithmetic runs in the HIP kernels behind include/gaviko_hip.h (gaviko_amd/engine.py, kind "ssf"): every
`ssf_ada(x, s, t) = x*s + t` sits directly after a LayerNorm or a Linear, so the engine folds it into effective LayerNorm
affines / effective weights and biases each step and runs the plain ViT kernels; the scale/shift gradients are two column
sums per site.
"""
from __future__ import annotations

import torch
from torch import nn

from ..utils.load_pretrained import mapping_vit
from .vision_transformer import HotPathModule, _Container, pair


def init_ssf_scale_shift(dim):
    """ssf.py:14-21: scale ~ N(1, .02), shift ~ N(0, .02)."""
    scale, shift = nn.Parameter(torch.ones(dim)), nn.Parameter(torch.zeros(dim))
    nn.init.normal_(scale, mean=1, std=0.02)
    nn.init.normal_(shift, std=0.02)
    return scale, shift


class FeedForward(_Container):
    def __init__(self, dim, hidden_dim, dropout=0.0):               # ssf.py:49-62
        super().__init__()
        self.ssf_scale_0, self.ssf_shift_0 = init_ssf_scale_shift(dim)
        self.ssf_scale_1, self.ssf_shift_1 = init_ssf_scale_shift(hidden_dim)
        self.ssf_scale_2, self.ssf_shift_2 = init_ssf_scale_shift(dim)
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))


class Attention(_Container):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0):    # ssf.py:77-102
        super().__init__()
        inner = dim_head * heads
        self.heads, self.scale = heads, dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.attend = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))
        self.ssf_scale_0, self.ssf_shift_0 = init_ssf_scale_shift(dim)
        self.ssf_scale_1, self.ssf_shift_1 = init_ssf_scale_shift(inner * 3)
        self.ssf_scale_2, self.ssf_shift_2 = init_ssf_scale_shift(dim)


class Transformer(_Container):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0):   # ssf.py:124-138 (LayerScale is Identity: init_values None)
        super().__init__()
        self.ls1, self.ls2 = nn.Identity(), nn.Identity()
        self.norm = nn.LayerNorm(dim)
        self.ssf_scale_1, self.ssf_shift_1 = init_ssf_scale_shift(dim)
        self.layers = nn.ModuleList([nn.ModuleList([Attention(dim, heads, dim_head, dropout), FeedForward(dim, mlp_dim, dropout)])
                                     for _ in range(depth)])


class ScalingShiftingFeatures(HotPathModule):
    _kind = "ssf"

    def __init__(self, *, image_size, image_patch_size, frames, frame_patch_size, num_classes, pool="cls", channels=3, dim_head=64,
                 dropout=0.0, emb_dropout=0.0, backbone=None, freeze_vit=False, **kwargs):
        super().__init__()
        depth, heads, dim, mlp_dim = mapping_vit(backbone)
        ih, iw = pair(image_size)
        ph, pw = pair(image_patch_size)
        assert ih % ph == 0 and iw % pw == 0, "Image dimensions must be divisible by the patch size."
        assert frames % frame_patch_size == 0, "Frames must be divisible by frame patch size"
        assert pool in {"cls", "mean"}, "pool type must be either cls (cls token) or mean (mean pooling)"
        self.num_patches = (ih // ph) * (iw // pw) * (frames // frame_patch_size)
        self.image_size, self.image_patch_size = image_size, image_patch_size
        self.frames, self.frame_patch_size = frames, frame_patch_size
        self.conv_proj = nn.Sequential(nn.Conv3d(channels, dim, kernel_size=(frame_patch_size, image_patch_size, image_patch_size),
                                                 stride=(frame_patch_size, image_patch_size, image_patch_size)))
        self.ssf_scale_1, self.ssf_shift_1 = init_ssf_scale_shift(dim)
        self.pos_embedding = nn.Parameter(torch.randn(1, self.num_patches + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Linear(dim, num_classes)
        self.freeze_vit = freeze_vit
        nn.init.xavier_uniform_(self.mlp_head.weight)
        nn.init.zeros_(self.mlp_head.bias)
        if freeze_vit:                                              # ssf.py:192-197
            for k, p in self.named_parameters():
                if "transformer" in k or "cls_token" in k or "conv_proj" in k or "pos_embedding" in k:
                    p.requires_grad = False
                if "scale" in k or "shift" in k:
                    p.requires_grad = True
        self.__dict__["_kw_precision"] = kwargs.get("precision")      # "bf16" (default) | "fp32": see HotPathModule.set_precision
        self._cfg = dict(image_size=ih, image_patch_size=ph, frames=frames, frame_patch_size=frame_patch_size, num_classes=num_classes,
                         pool=pool, channels=channels, dim_head=dim_head, backbone=backbone, dropout=dropout, emb_dropout=emb_dropout)
        self._load_backbone()

    def train(self, mode=True):
        """ssf.py:205-217 (returns None)."""
        if mode:
            super().train(mode)
            if self.freeze_vit:
                self.transformer.eval()
                self.conv_proj.eval()
                self.dropout.eval()
                self.mlp_head.train()
        else:
            for module in self.children():
                module.eval()

    def _drop_config(self):
        # freeze_vit=True keeps transformer / conv_proj / dropout in eval (train() above); otherwise every nn.Dropout follows .training
        return {"dropout": self._cfg["dropout"] if self.transformer.layers[0][0].dropout.training else 0.0,
                "emb_dropout": self._cfg["emb_dropout"] if self.dropout.training else 0.0}

    def forward(self, img):
        return self._run(img)
