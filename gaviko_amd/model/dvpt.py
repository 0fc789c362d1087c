"""Host-side mirror of the reference's model/dvpt.py (`--method dvpt`, Dynamic Visual Prompt Tuning) for the MI355X path.

`DynamicVisualPromptTuning` keeps the reference's kwargs (dvpt.py:85-103), parameter / state_dict names
(`transformer.layers.{i}.0.{attn,mlp,prompt_proj}`, `prompt_embeddings`, `prompt_positional_embedding`), freeze rule
(dvpt.py:158-163: anything with 'prompt' or 'head' in its name trains) and `train()` override (170-184, returns None).
The sub-modules are parameter containers; the arithmetic runs in the HIP kernels behind include/gaviko_hip.h
(gaviko_amd/engine.py, kind "dvpt").
"""
from __future__ import annotations

import torch
from torch import nn

from ..utils.load_pretrained import mapping_vit
from .vision_transformer import Attention, FeedForward, HotPathModule, _Container, pair


class share_MLP(_Container):                                      # noqa: N801  (the reference's class name)
    def __init__(self, d_model, num_prompts):                     # dvpt.py:24-35
        super().__init__()
        self.latent_dim = 20
        self.prompt_key_proj_d = nn.Linear(d_model, self.latent_dim)
        self.prompt_key_proj_u = nn.Linear(self.latent_dim, d_model)
        self.prompt_gate = nn.Parameter(torch.zeros(1))
        self.num = num_prompts
        self.scale = d_model ** -0.5


class ResidualAttentionBlock(_Container):
    def __init__(self, dim, heads, dim_head, mlp_dim, num_prompts, dropout):     # dvpt.py:49-57
        super().__init__()
        self.attn = Attention(dim, heads, dim_head, dropout)
        self.mlp = FeedForward(dim, mlp_dim, dropout)
        self.prompt_proj = share_MLP(dim, num_prompts)


class Transformer(_Container):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, num_prompts, dropout=0.0, pool="cls"):      # dvpt.py:65-76
        super().__init__()
        self.num = num_prompts
        self.norm = nn.LayerNorm(dim)
        self.pool = pool
        self.layers = nn.ModuleList([nn.ModuleList([ResidualAttentionBlock(dim, heads, dim_head, mlp_dim, num_prompts, dropout)])
                                     for _ in range(depth)])


class DynamicVisualPromptTuning(HotPathModule):
    _kind = "dvpt"

    def __init__(self, *, image_size, image_patch_size, frames, frame_patch_size, num_classes, pool="cls", channels=3, dim_head=64,
                 dropout=0.0, emb_dropout=0.0, num_prompts=50, freeze_vit=False, backbone=None, **kwargs):
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
        self.prompt_positional_embedding = nn.Parameter(dim ** -0.5 * torch.randn(1, num_prompts, dim))
        self.prompt_embeddings = nn.Parameter(torch.randn(1, num_prompts, dim))
        self.conv_proj = nn.Sequential(nn.Conv3d(channels, dim, kernel_size=(frame_patch_size, image_patch_size, image_patch_size),
                                                 stride=(frame_patch_size, image_patch_size, image_patch_size)))
        self.pos_embedding = nn.Parameter(torch.randn(1, self.num_patches + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, num_prompts, dropout, pool)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Linear(dim, num_classes)
        self.freeze_vit = freeze_vit
        nn.init.xavier_uniform_(self.mlp_head.weight)
        nn.init.zeros_(self.mlp_head.bias)
        if freeze_vit:                                              # dvpt.py:158-163
            for k, p in self.named_parameters():
                if "transformer" in k or "cls_token" in k or "conv_proj" in k or "pos_embedding" in k:
                    p.requires_grad = False
                if "prompt" in k or "head" in k:
                    p.requires_grad = True
        self.__dict__["_kw_precision"] = kwargs.get("precision")      # "bf16" (default) | "fp32": see HotPathModule.set_precision
        self._cfg = dict(image_size=ih, image_patch_size=ph, frames=frames, frame_patch_size=frame_patch_size, num_classes=num_classes,
                         pool=pool, channels=channels, dim_head=dim_head, backbone=backbone, dropout=dropout, emb_dropout=emb_dropout,
                         num_prompts=num_prompts)
        self._load_backbone()

    def train(self, mode=True):
        """dvpt.py:170-184 (returns None)."""
        if mode:
            super().train(mode)
            if self.freeze_vit:
                self.transformer.eval()
                self.conv_proj.eval()
                self.dropout.eval()
                self.mlp_head.train()
                for layer in self.transformer.layers:
                    layer[0].prompt_proj.train()
        else:
            for module in self.children():
                module.eval()

    def _drop_config(self):
        # freeze_vit=True keeps transformer / conv_proj / dropout in eval (train() above); otherwise every nn.Dropout follows .training
        return {"dropout": self._cfg["dropout"] if self.transformer.layers[0][0].attn.dropout.training else 0.0,
                "emb_dropout": self._cfg["emb_dropout"] if self.dropout.training else 0.0}

    def forward(self, img):
        return self._run(img)
