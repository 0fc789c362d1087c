"""Host-side mirror of the reference's model/melo.py (LoRA r on the q and v blocks of the fused qkv Linear).

`MeLO(vit=VisionTransformer(**cfg), **cfg)` as in train.py:148-150.  Parameter names follow the reference's wrapping
(`lora_vit.transformer.layers.{i}.0.to_qkv.{qkv,linear_a_q,linear_b_q,linear_a_v,linear_b_v}.weight`, melo.py:66-87) and the
head is re-created (90-91).  On the MI355X path the rank-r update is folded into the bf16 QKV operand every step
(engine._melo_merge), so the forward is the plain MFMA GEMM; the integer scale alpha // r of melo.py:45-46 is kept.
"""
from __future__ import annotations

import math

from torch import nn

from .vision_transformer import HotPathModule, _Container


class _LoRA_qkv_timm(_Container):
    def __init__(self, qkv, linear_a_q, linear_b_q, linear_a_v, linear_b_v, r, alpha):   # melo.py:19-39
        super().__init__()
        self.qkv, self.linear_a_q, self.linear_b_q, self.linear_a_v, self.linear_b_v = qkv, linear_a_q, linear_b_q, linear_a_v, linear_b_v
        self.dim, self.r, self.alpha = qkv.in_features, r, alpha


class MeLO(HotPathModule):
    _kind = "melo"

    def __init__(self, vit, r: int, alpha: int, num_classes: int, lora_layer=None, **kwargs):
        super().__init__()
        assert r > 0
        assert alpha > 0
        layers = vit.transformer.layers
        self.lora_layer = list(lora_layer) if lora_layer else list(range(len(layers)))       # melo.py:53-56
        self.w_As, self.w_Bs = [], []
        for p in vit.parameters():                       # melo.py:63-65
            p.requires_grad = False
        for i, (attn, mlp) in enumerate(layers):
            if i not in self.lora_layer:                 # melo.py:67-68: that layer keeps its plain (frozen) to_qkv Linear
                continue
            base = attn.to_qkv
            self.dim = base.in_features
            aq, bq = nn.Linear(self.dim, r, bias=False), nn.Linear(r, self.dim, bias=False)
            av, bv = nn.Linear(self.dim, r, bias=False), nn.Linear(r, self.dim, bias=False)
            self.w_As += [aq, av]
            self.w_Bs += [bq, bv]
            attn.to_qkv = _LoRA_qkv_timm(base, aq, bq, av, bv, r, alpha)
        for w_a in self.w_As:                            # melo.py:94-98
            nn.init.kaiming_uniform_(w_a.weight, a=math.sqrt(5))
        for w_b in self.w_Bs:
            nn.init.zeros_(w_b.weight)
        self.lora_vit = vit
        if num_classes > 0:
            self.lora_vit.mlp_head = nn.Linear(self.dim, num_classes)
        self.__dict__["_kw_precision"] = kwargs.get("precision")      # "bf16" (default) | "fp32": see HotPathModule.set_precision
        self._cfg = dict(vit._cfg, r=r, alpha=alpha, lora_layer=sorted(self.lora_layer))

    def _drop_config(self):
        # no train() override (melo.py:56-110): the wrapped ViT's nn.Dropout modules follow module.training
        vt = self.lora_vit
        return {"dropout": vt._cfg["dropout"] if vt.transformer.layers[0][0].dropout.training else 0.0,
                "emb_dropout": vt._cfg["emb_dropout"] if vt.dropout.training else 0.0}

    def forward(self, x):
        return self._run(x)
