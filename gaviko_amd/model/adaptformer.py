"""Host-side mirror of the reference's model/adaptformer.py (parallel bottleneck adapter beside the MLP).

`AdaptFormer` keeps the reference's kwargs (adaptformer.py:102-119), parameter names (`transformer.layers.{i}.{0,1,2}`:
Attention / Adapter / FeedForward), freeze rule (163-168), init (52-56, 170-173) and train() override (175-191).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import lib as L
from ..utils.load_pretrained import mapping_vit
from .vision_transformer import Attention, FeedForward, HotPathModule, _Container, pair


class Adapter(_Container):
    """Parameters of adaptformer.py:22-56: LayerNorm 'in', down d->64, ReLU, up 64->d (zero-init 'lora' option), scalar 1.0."""

    def __init__(self, d_dim, down_dim=64, dropout=0.0, init_option="lora", adapter_scalar="1.0", adapter_layernorm_option="in"):
        super().__init__()
        if adapter_layernorm_option != "in" or adapter_scalar == "learnable_scalar" or init_option != "lora":
            raise L.GavikoHipError("only the reference's default Adapter (layernorm 'in', fixed scalar, 'lora' init) is built")
        self.d_dim, self.down_dim = d_dim, down_dim
        self.adapter_layernorm_option = adapter_layernorm_option
        self.adapter_layer_norm_before = nn.LayerNorm(d_dim)
        self.scale = float(adapter_scalar)
        if self.scale != 1.0:
            raise L.GavikoHipError("adapter_scalar != 1.0 is not built")
        self.down_adapter_proj = nn.Linear(d_dim, down_dim)
        self.non_linear_func = nn.ReLU()
        self.up_adapter_proj = nn.Linear(down_dim, d_dim)
        self.dropout = dropout                       # F.dropout(p=0.0) in the reference (adaptformer.py:65): identity
        with torch.no_grad():
            nn.init.kaiming_uniform_(self.down_adapter_proj.weight, a=math.sqrt(5))
            nn.init.zeros_(self.up_adapter_proj.weight)
            nn.init.zeros_(self.down_adapter_proj.bias)
            nn.init.zeros_(self.up_adapter_proj.bias)


class Transformer(_Container):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0):     # adaptformer.py:81-91
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList([nn.ModuleList([Attention(dim, heads, dim_head, dropout), Adapter(dim), FeedForward(dim, mlp_dim, dropout)])
                                     for _ in range(depth)])


class AdaptFormer(HotPathModule):
    _kind = "adaptformer"

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
        if freeze_vit:                                            # adaptformer.py:163-168
            for k, p in self.named_parameters():
                if "transformer" in k or "cls_token" in k or "conv_proj" in k or "pos_embedding" in k:
                    p.requires_grad = False
                if "adapter" in k or "head" in k:
                    p.requires_grad = True
        self.__dict__["_kw_precision"] = kwargs.get("precision")      # "bf16" (default) | "fp32": see HotPathModule.set_precision
        self._cfg = dict(image_size=ih, image_patch_size=ph, frames=frames, frame_patch_size=frame_patch_size, num_classes=num_classes,
                         pool=pool, channels=channels, dim_head=dim_head, backbone=backbone, dropout=dropout, emb_dropout=emb_dropout)
        self._load_backbone()

    def train(self, mode=True):
        """adaptformer.py:175-191 (returns None)."""
        if mode:
            super().train(mode)
            if self.freeze_vit:
                self.transformer.eval()
                self.conv_proj.eval()
                self.dropout.eval()
                self.mlp_head.train()
                for layer in self.transformer.layers:
                    layer[1].train()
        else:
            for module in self.children():
                module.eval()

    def _drop_config(self):
        # freeze_vit=True keeps transformer / conv_proj / dropout in eval (train() above); with freeze_vit=False every nn.Dropout of the
        # backbone follows module.training (adaptformer.py:175-191) -- the Adapter's own dropout is p = 0.0 (adaptformer.py:23-27)
        return {"dropout": self._cfg["dropout"] if self.transformer.layers[0][0].dropout.training else 0.0,
                "emb_dropout": self._cfg["emb_dropout"] if self.dropout.training else 0.0}

    def forward(self, img):
        return self._run(img)
