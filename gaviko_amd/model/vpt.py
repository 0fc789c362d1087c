"""Host-side mirror of the reference's model/vpt.py (VPT shallow / deep).

`PromptedVisionTransformer` keeps the reference's kwargs (vpt.py:19-41), parameter names (`vision_transformer.*`,
`prompt_proj.*`, `deep_prompt_embeddings` | `prompt_embeddings`), freeze rule (91-94), inits (96-104) and train()
override (106-119).  Token order is [cls | prompts | patches] (127-131); deep VPT re-inserts prompts before every layer
and -- reference quirk, reproduced -- drops `prompt_dim - num_prompts` patch tokens per layer (147-153).
"""
from __future__ import annotations

import torch
from torch import nn

from .vision_transformer import HotPathModule, VisionTransformer


class PromptedVisionTransformer(HotPathModule):
    _kind = "vpt"

    def __init__(self, image_size, image_patch_size, frames, frame_patch_size, dropout=0.0, emb_dropout=0.0, num_classes=5, channels=3,
                 dim_head=64, freeze_vit=True, pool="cls", backbone=None, prompt_dropout=0.0, prompt_dim=64, num_prompts=8,
                 deep_prompt=True, **kwargs):
        super().__init__()
        self.vision_transformer = VisionTransformer(image_size=image_size, image_patch_size=image_patch_size, frames=frames,
                                                    frame_patch_size=frame_patch_size, num_classes=num_classes, pool=pool, channels=channels,
                                                    dim_head=dim_head, dropout=dropout, emb_dropout=emb_dropout, backbone=backbone)
        depth = self.vision_transformer.depth
        hidden = self.vision_transformer.mlp_head.in_features
        self.image_size, self.image_patch_size = image_size, image_patch_size
        self.num_layers, self.hidden_dim, self.num_classes = depth, hidden, num_classes
        self.emb_dropout, self.dropout, self.deep_prompt = emb_dropout, dropout, deep_prompt
        # (the reference appends a line to ./deep_prompt.txt here, vpt.py:54-55: a debugging side effect, not reproduced)
        self.prompt_proj = nn.Linear(prompt_dim, hidden)
        self.prompt_dropout = nn.Dropout(prompt_dropout)
        if deep_prompt:
            self.deep_prompt_embeddings = nn.Parameter(torch.zeros(depth, num_prompts, prompt_dim))
            nn.init.xavier_uniform_(self.deep_prompt_embeddings.data)
        else:
            self.prompt_embeddings = nn.Parameter(torch.zeros(1, num_prompts, prompt_dim))
            nn.init.xavier_uniform_(self.prompt_embeddings.data)
        self.freeze_vit = freeze_vit
        nn.init.xavier_uniform_(self.vision_transformer.mlp_head.weight)
        nn.init.zeros_(self.vision_transformer.mlp_head.bias)
        nn.init.xavier_uniform_(self.prompt_proj.weight)
        nn.init.zeros_(self.prompt_proj.bias)
        if freeze_vit:                                            # vpt.py:91-94
            for k, p in self.vision_transformer.named_parameters():
                if "transformer" in k or "cls_token" in k or "conv_proj" in k or "pos_embedding" in k:
                    p.requires_grad = False
        self.__dict__["_kw_precision"] = kwargs.get("precision")      # "bf16" (default) | "fp32": see HotPathModule.set_precision
        self._cfg = dict(self.vision_transformer._cfg, num_prompts=num_prompts, prompt_dim=prompt_dim, deep_prompt=deep_prompt,
                         prompt_dropout=prompt_dropout)

    def train(self, mode=True):
        """vpt.py:106-119 (returns None)."""
        if mode:
            super().train(mode)
            if self.freeze_vit:
                vt = self.vision_transformer
                vt.transformer.eval()
                vt.conv_proj.eval()
                vt.dropout.eval()
                vt.mlp_head.train()
                self.prompt_proj.train()
        else:
            for module in self.children():
                module.eval()

    def _drop_config(self):
        # vpt.py:106-119 leaves prompt_dropout in training mode while the frozen backbone (and its dropouts) go to eval
        # ... and with freeze_vit=False the backbone's own dropouts follow .training as well
        vt = self.vision_transformer
        return {"prompt_dropout": self._cfg["prompt_dropout"] if self.prompt_dropout.training else 0.0,
                "dropout": self._cfg["dropout"] if vt.transformer.layers[0][0].dropout.training else 0.0,
                "emb_dropout": self._cfg["emb_dropout"] if vt.dropout.training else 0.0}

    def forward(self, x):
        return self._run(x)
