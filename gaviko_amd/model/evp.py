"""Host-side mirror of the reference's model/evp.py (`--method evp`, Explicit Visual Prompting) for the MI355X path.

`ExplicitVisualPrompting` keeps the reference's kwargs (evp.py:244-266), parameter / state_dict names (note `conv_proj.proj.*`:
the patch embedding is a `PatchEmbed` here, and `prompt_generator.{shared_mlp, embedding_generator, lightweight_mlp_i.0,
prompt_generator.proj}`), freeze rule (evp.py:322-327) and `train()` override (333-344, returns None).  The sub-modules are
parameter containers; the arithmetic runs in the HIP kernels behind include/gaviko_hip.h (gaviko_amd/engine.py, kind "evp").
"""
from __future__ import annotations

import torch
from torch import nn

from ..utils.load_pretrained import mapping_vit
from .vision_transformer import HotPathModule, Transformer, _Container, pair


class PatchEmbed(_Container):
    def __init__(self, img_size=160, frames=120, image_patch_size=16, frame_patch_size=12, in_chans=3, dim=768):   # evp.py:149-158
        super().__init__()
        self.img_size = img_size
        self.proj = nn.Conv3d(in_chans, dim, kernel_size=(frame_patch_size, image_patch_size, image_patch_size),
                              stride=(frame_patch_size, image_patch_size, image_patch_size))


class PromptGenerator(_Container):
    def __init__(self, scale_factor, dim, depth, input_type, freq_nums, handcrafted_tune, embedding_tune, img_size, frames, image_patch_size,
                 frame_patch_size, channels):                                                                        # evp.py:24-55
        super().__init__()
        self.mode = "stack"
        self.scale_factor, self.embed_dim, self.input_type, self.freq_nums, self.depth = scale_factor, dim, input_type, freq_nums, depth
        self.handcrafted_tune, self.embedding_tune = handcrafted_tune, embedding_tune
        r = dim // scale_factor
        self.shared_mlp = nn.Linear(r, dim)
        self.embedding_generator = nn.Linear(dim, r)
        for i in range(depth):
            setattr(self, f"lightweight_mlp_{i}", nn.Sequential(nn.Linear(r, r), nn.GELU()))
        self.prompt_generator = PatchEmbed(img_size=img_size, frames=frames, image_patch_size=image_patch_size, frame_patch_size=frame_patch_size,
                                           in_chans=channels, dim=r)
        for m in self.modules():                                                                                     # _init_weights, evp.py:57-70
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)


class ExplicitVisualPrompting(HotPathModule):
    _kind = "evp"

    def __init__(self, *, image_size, image_patch_size, frames, frame_patch_size, num_classes, pool="cls", channels=3, dim_head=64,
                 dropout=0.0, emb_dropout=0.0, backbone=None, freeze_vit=False, scale_factor=32, input_type="fft", freq_nums=0.25,
                 handcrafted_tune=True, embedding_tune=True, **kwargs):
        super().__init__()
        depth, heads, dim, mlp_dim = mapping_vit(backbone)
        ih, iw = pair(image_size)
        ph, pw = pair(image_patch_size)
        assert ih % ph == 0 and iw % pw == 0, "Image dimensions must be divisible by the patch size."
        assert frames % frame_patch_size == 0, "Frames must be divisible by frame patch size"
        assert pool in {"cls", "mean"}, "pool type must be either cls (cls token) or mean (mean pooling)"
        if input_type != "fft":
            raise NotImplementedError("EVP input_type other than 'fft' (the shipped configuration) is not built")
        self.num_patches = (ih // ph) * (iw // pw) * (frames // frame_patch_size)
        self.image_size, self.image_patch_size = image_size, image_patch_size
        self.frames, self.frame_patch_size = frames, frame_patch_size
        self.conv_proj = PatchEmbed(image_size, frames, image_patch_size, frame_patch_size, channels, dim)
        self.pos_embedding = nn.Parameter(torch.randn(1, self.num_patches + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Linear(dim, num_classes)
        self.scale_factor, self.input_type, self.freq_nums = scale_factor, input_type, freq_nums
        self.handcrafted_tune, self.embedding_tune = handcrafted_tune, embedding_tune
        self.prompt_generator = PromptGenerator(scale_factor, dim, depth, input_type, freq_nums, handcrafted_tune, embedding_tune, image_size,
                                                frames, image_patch_size, frame_patch_size, channels)
        self.freeze_vit = freeze_vit
        if freeze_vit:                                              # evp.py:322-327
            for k, p in self.named_parameters():
                if "transformer" in k or "cls_token" in k or "conv_proj" in k or "pos_embedding" in k:
                    p.requires_grad = False
                if "prompt_generator" in k:
                    p.requires_grad = True
        self.__dict__["_kw_precision"] = kwargs.get("precision")      # "bf16" (default) | "fp32": see HotPathModule.set_precision
        self._cfg = dict(image_size=ih, image_patch_size=ph, frames=frames, frame_patch_size=frame_patch_size, num_classes=num_classes,
                         pool=pool, channels=channels, dim_head=dim_head, backbone=backbone, dropout=dropout, emb_dropout=emb_dropout,
                         scale_factor=scale_factor, freq_nums=freq_nums)
        self._load_backbone()

    def train(self, mode=True):
        """evp.py:333-344 (returns None)."""
        if mode:
            super().train(mode)
            if self.freeze_vit:
                self.transformer.eval()
                self.conv_proj.eval()
                self.dropout.eval()
                self.mlp_head.train()
                self.prompt_generator.train()
        else:
            for module in self.children():
                module.eval()

    def _drop_config(self):
        # freeze_vit=True keeps transformer / conv_proj / dropout in eval (train() above); otherwise every nn.Dropout follows .training
        return {"dropout": self._cfg["dropout"] if self.transformer.layers[0][0].dropout.training else 0.0,
                "emb_dropout": self._cfg["emb_dropout"] if self.dropout.training else 0.0}

    def forward(self, img):
        return self._run(img)
