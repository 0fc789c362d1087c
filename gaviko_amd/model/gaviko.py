"""Host-side mirror of the reference's model/gaviko.py (GAViKO: masked-window local attention + gated prompt awakening).

Class names, constructor kwargs (gaviko.py:328-355), parameter names (including the
global_query / local_query aliases that state_dict() lists), the freeze rule (428-434), init scheme (445-511), the
train() override (513-528, returns None) and `forward(img) -> logits` are the reference's; the computation is the HIP
path (engine.py).  Sub-modules are parameter containers.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import lib as L
from ..utils.load_pretrained import mapping_vit
from .vision_transformer import Attention, FeedForward, HotPathModule, _Container, pair


class QuickGELU(nn.Module):
    def forward(self, x):                      # gaviko.py:15-17 (kept callable: it is a pure activation)
        return x * torch.sigmoid(1.702 * x)


class PromptRelevantEstimator(_Container):
    """PRE gate parameters (gaviko.py:20-47): [LayerNorm(l), Linear(l,64), GELU, Linear(64,P), Sigmoid]."""

    def __init__(self, latent_dim, num_prompts):
        super().__init__()
        self.cls_analyzer_ = nn.Sequential(nn.LayerNorm(latent_dim), nn.Linear(latent_dim, 64), nn.GELU(), nn.Linear(64, num_prompts),
                                           nn.Sigmoid())

    @property
    def cls_analyzer(self):
        return self.cls_analyzer_

    def __getitem__(self, index):
        return self.cls_analyzer_[index]


class PromptContextFusion(_Container):
    """PCF balance parameters (gaviko.py:48-70): [LayerNorm(l), Linear(l,1), Sigmoid]."""

    def __init__(self, latent_dim):
        super().__init__()
        self.gl_balancer_ = nn.Sequential(nn.LayerNorm(latent_dim), nn.Linear(latent_dim, 1), nn.Sigmoid())

    @property
    def gl_balancer(self):
        return self.gl_balancer_

    def __getitem__(self, index):
        return self.gl_balancer_[index]


class GlobalAttention(_Container):
    def __init__(self, latent_dim, num_prompts):   # gaviko.py:97-107
        super().__init__()
        self.latent_dim, self.num_prompts, self.scale = latent_dim, num_prompts, latent_dim ** -0.5
        self.query_proj = nn.Linear(latent_dim, latent_dim)


class LocalAttention(_Container):
    def __init__(self, latent_dim):                # gaviko.py:110-119
        super().__init__()
        self.latent_dim, self.scale = latent_dim, latent_dim ** -0.5
        self.query_proj = nn.Linear(latent_dim, latent_dim)


class Awakening_Prompt(_Container):
    """GPA parameters (gaviko.py:121-147)."""

    def __init__(self, dim, num_prompts, prompt_latent_dim=20):
        super().__init__()
        self.latent_dim, self.num_prompts, self.scale = prompt_latent_dim, num_prompts, dim ** -0.5
        self.proj_down = nn.Sequential(nn.Linear(dim, prompt_latent_dim), QuickGELU())
        self.proj_up = nn.Linear(prompt_latent_dim, dim)
        self.cls_analyzer = PromptRelevantEstimator(prompt_latent_dim, num_prompts)
        self.gl_balancer = PromptContextFusion(prompt_latent_dim)
        self.global_attention = GlobalAttention(prompt_latent_dim, num_prompts)
        self.local_attention = LocalAttention(prompt_latent_dim)
        self.global_query = self.global_attention.query_proj     # aliases: extra state_dict keys, same storage
        self.local_query = self.local_attention.query_proj
        self.attend = nn.Softmax(dim=-1)


class LocalSelfAttention(_Container):
    """MWSA parameters (gaviko.py:189-210).  The reference's dense [1,N,N] mask attribute is not materialised: the
    window is index arithmetic inside gvk_window_attn_*; `mask` is provided lazily for code that inspects it."""

    def __init__(self, dim, local_k=(3, 6, 6), DHW=None, attn_drop=0.0, proj_drop=0.0, local_dim=20, qkv_bias=False, dtype=torch.float32):
        super().__init__()
        self.dim, self.scale, self.latent_dim = dim, dim ** -0.5, local_dim
        self.norm = nn.LayerNorm(dim)
        self.proj_down = nn.Linear(dim, local_dim)
        self.qkv = nn.Linear(local_dim, local_dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj_up = nn.Linear(local_dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.DHW, self.local_k, self._mask_dtype = DHW, tuple(local_k), dtype
        if qkv_bias:
            raise L.GavikoHipError("MWSA qkv bias is not built (the reference always passes qkv_bias=False, gaviko.py:272)")

    @property
    def mask(self):
        """Additive 0/-inf [1,N,N] mask equal to gaviko.py:212-227 (built on demand, never used by the kernels)."""
        if self.DHW is None:
            return None
        D, H, W = self.DHW
        idx = [torch.arange(n) for n in (D, H, W)]
        allow = None
        for ax, (n, k) in enumerate(zip((D, H, W), self.local_k)):
            q = idx[ax].view(-1, 1)
            lo = q - k // 2
            a = (idx[ax].view(1, -1) >= lo) & (idx[ax].view(1, -1) < lo + k)        # [n, n]
            shape_q = [1, 1, 1, 1, 1, 1]
            shape_q[ax], shape_q[3 + ax] = n, n
            a = a.view(*shape_q)
            allow = a if allow is None else allow & a
        allow = allow.expand(D, H, W, D, H, W).reshape(D * H * W, D * H * W)
        m = torch.full(allow.shape, float("-inf"), dtype=self._mask_dtype)
        m[allow] = 0.0
        return m.unsqueeze(0)


class Transformer(_Container):
    """gaviko.py:246-289: ceil(depth/share_factor) shared local_attns / prompt_projs, depth attns / mlps, final norm."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, num_prompts, prompt_latent_dim, DHW, local_k, share_factor=1, attn_drop=0.0,
                 proj_drop=0.0, local_dim=20, dropout=0.0, dtype=torch.float32):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.num_prompts, self.depth, self.share_factor = num_prompts, depth, share_factor
        n_unique = math.ceil(depth / share_factor)
        self.local_attns = nn.ModuleList([LocalSelfAttention(dim, local_k, DHW, attn_drop, proj_drop, local_dim, qkv_bias=False, dtype=dtype)
                                          for _ in range(n_unique)])
        self.prompt_projs = nn.ModuleList([Awakening_Prompt(dim, num_prompts, prompt_latent_dim) for _ in range(n_unique)])
        self.attns = nn.ModuleList([Attention(dim, heads, dim_head, dropout) for _ in range(depth)])
        self.mlps = nn.ModuleList([FeedForward(dim, mlp_dim, dropout) for _ in range(depth)])


class AdaptiveFusionHead(_Container):
    def __init__(self, dim, num_prompts, num_classes):   # gaviko.py:308-325: Linear over mean(prompts + CLS)
        super().__init__()
        self.head = nn.Linear(dim, num_classes)
        self.num_prompts = num_prompts


class Gaviko(HotPathModule):
    _kind = "gaviko"

    def __init__(self, *, image_size, image_patch_size, frames, frame_patch_size, num_classes, pool="cls", channels=1, dim_head=64,
                 dropout=0.0, emb_dropout=0.0, backbone=None, num_prompts=8, prompt_latent_dim=20, local_dim=20, local_k=(3, 6, 6),
                 DHW=(10, 10, 10), attn_drop=0.2, proj_drop=0.2, freeze_vit=False, share_factor=1, fp16=False, **kwargs):
        super().__init__()
        self.dtype = torch.float32 if not fp16 else torch.float16
        depth, heads, dim, mlp_dim = mapping_vit(backbone)
        ih, iw = pair(image_size)
        ph, pw = pair(image_patch_size)
        assert ih % ph == 0 and iw % pw == 0, "Image dimensions must be divisible by the patch size."
        assert frames % frame_patch_size == 0, "Frames must be divisible by frame patch size"
        assert pool in {"cls", "mean"}, "pool type must be either cls (cls token) or mean (mean pooling)"
        self.num_patches = (ih // ph) * (iw // pw) * (frames // frame_patch_size)
        self.image_size, self.image_patch_size = image_size, image_patch_size
        self.frames, self.frame_patch_size = frames, frame_patch_size
        self.num_prompts, self.local_dim, self.local_k, self.prompt_latent_dim = num_prompts, local_dim, local_k, prompt_latent_dim
        self.conv_proj = nn.Sequential(nn.Conv3d(channels, dim, kernel_size=(frame_patch_size, image_patch_size, image_patch_size),
                                                 stride=(frame_patch_size, image_patch_size, image_patch_size)))
        self.pos_embedding = nn.Parameter(torch.randn(1, self.num_patches + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, num_prompts, prompt_latent_dim,
                                       None if DHW is None else tuple(DHW), tuple(local_k), share_factor, attn_drop, proj_drop, local_dim,
                                       dropout, dtype=self.dtype)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = AdaptiveFusionHead(dim, num_prompts, num_classes)
        self.prompt_positional_embedding = nn.Parameter(dim ** -0.5 * torch.randn(1, num_prompts, dim))
        self.prompt_embeddings = nn.Parameter(torch.randn(1, num_prompts, dim))
        self.freeze_vit = freeze_vit
        if freeze_vit:                              # gaviko.py:429-434, both tests applied to every name in this order
            for k, p in self.named_parameters():
                if "transformer" in k or "cls_token" in k or "conv_proj" in k or "pos_embedding" in k:
                    p.requires_grad = False
                if "head" in k or "prompt" in k or "local_attn" in k:
                    p.requires_grad = True
        # (timm weight fetch of gaviko.py:436-441 is outside the hot path)
        self.init_weights()
        self.__dict__["_kw_precision"] = kwargs.get("precision")      # "bf16" (default) | "fp32": see HotPathModule.set_precision
        self._cfg = dict(image_size=ih, image_patch_size=ph, frames=frames, frame_patch_size=frame_patch_size, num_classes=num_classes,
                         pool=pool, channels=channels, dim_head=dim_head, backbone=backbone, num_prompts=num_prompts,
                         prompt_latent_dim=prompt_latent_dim, local_dim=local_dim, local_k=tuple(local_k),
                         DHW=None if DHW is None else tuple(DHW), share_factor=share_factor, attn_drop=attn_drop, proj_drop=proj_drop,
                         dropout=dropout, emb_dropout=emb_dropout)
        self._load_backbone()

    def init_weights(model, scale_factor=1.0):
        """Initialisation scheme of gaviko.py:445-511 (including its quirk: proj_down.bias zeroed twice, proj_up.bias left
        at the nn.Linear default)."""
        with torch.no_grad():
            std = 0.02 * scale_factor
            model.prompt_embeddings.normal_(0.0, std).clamp_(-2 * std, 2 * std)
            model.prompt_positional_embedding.normal_(0.0, 0.01 * scale_factor)
        xav, zero = nn.init.xavier_uniform_, nn.init.zeros_
        for pp in model.transformer.prompt_projs:
            xav(pp.proj_down[0].weight, gain=0.7 * scale_factor); zero(pp.proj_down[0].bias)
            xav(pp.proj_up.weight, gain=0.7 * scale_factor); zero(pp.proj_up.bias)
            for q in (pp.global_query, pp.local_query):
                nn.init.orthogonal_(q.weight, gain=scale_factor); zero(q.bias)
            xav(pp.cls_analyzer[1].weight, gain=1.0); zero(pp.cls_analyzer[1].bias)
            xav(pp.cls_analyzer[3].weight, gain=1.0); nn.init.constant_(pp.cls_analyzer[3].bias, 0.0)
            xav(pp.gl_balancer[1].weight, gain=1.0); nn.init.constant_(pp.gl_balancer[1].bias, 0.5)
        for la in model.transformer.local_attns:
            xav(la.proj_down.weight, gain=0.5 * scale_factor); zero(la.proj_down.bias)
            xav(la.qkv.weight, gain=1.0)
            xav(la.proj_up.weight, gain=0.5 * scale_factor); zero(la.proj_down.bias)
        xav(model.mlp_head.head.weight); zero(model.mlp_head.head.bias)

    def train(self, mode=True):
        """gaviko.py:513-528: with a frozen ViT the backbone (and its dropouts) stays in eval while the side paths train.
        Returns None like the reference (so `Gaviko(...).eval()` is not chainable)."""
        if mode:
            super().train(mode)
            if self.freeze_vit:
                self.transformer.eval()
                self.conv_proj.eval()
                self.dropout.eval()
                self.transformer.local_attns.train()
                self.transformer.prompt_projs.train()
                self.mlp_head.train()
        else:
            for module in self.children():
                module.eval()

    def _drop_config(self):
        la = self.transformer.local_attns
        live = len(la) > 0 and la[0].training
        # the backbone's own nn.Dropout modules: in eval under freeze_vit=True (train() above), following .training otherwise (gaviko.py:513-528)
        return {"attn_drop": self._cfg["attn_drop"] if live else 0.0, "proj_drop": self._cfg["proj_drop"] if live else 0.0,
                "dropout": self._cfg["dropout"] if self.transformer.attns[0].dropout.training else 0.0,
                "emb_dropout": self._cfg["emb_dropout"] if self.dropout.training else 0.0}

    def forward(self, img):
        return self._run(img)
