"""Host-side mirror of the reference's model/vision_transformer.py for the MI355X path.

Same class names, constructor kwargs, parameter / state_dict names and forward signature
(`forward(img[B,1,D,H,W]) -> logits[B,num_classes]`, vision_transformer.py:91-164), so the reference's train.py /
eval.py loop and its trainable-only checkpoints (train.py:473-483) work unchanged.  The sub-modules below are
*parameter containers*: the arithmetic of the whole forward/backward runs in the HIP kernels behind
include/gaviko_hip.h (see gaviko_amd/engine.py); there is no eager or CPU implementation here.
"""
from __future__ import annotations

import logging
import os

import torch
from torch import nn

from .. import lib as L
from ..engine import Engine
from ..utils.load_pretrained import mapping_vit


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


class _Container(nn.Module):
    """A module that only owns parameters; calling it is a usage error (the engine runs the math)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise L.GavikoHipError(f"{type(self).__name__} is a parameter container: call the top-level model, "
                               "which runs the fused HIP path")


class FeedForward(_Container):
    """Parameters of vision_transformer.py:26-38: net = [LayerNorm, Linear, GELU, Dropout, Linear, Dropout]."""

    def __init__(self, dim, hidden_dim, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))


class Attention(_Container):
    """Parameters of vision_transformer.py:40-58: pre-norm, bias-free fused qkv, to_out = [Linear, Dropout]."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        if heads == 1 and dim_head == dim:
            raise L.GavikoHipError("the single-head identity-projection corner of the reference Attention is not built")
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.attend = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))


class Transformer(_Container):
    """vision_transformer.py:74-89: layers[i] = [Attention, FeedForward]; final norm."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList([nn.ModuleList([Attention(dim, heads, dim_head, dropout), FeedForward(dim, mlp_dim, dropout)])
                                     for _ in range(depth)])


class _HotPathFn(torch.autograd.Function):
    """Bridges torch autograd to the engine: one node for the whole model.  The engine writes parameter gradients
    straight into views of its flat gradient buffer (what the data-parallel all-reduce sends), so the per-parameter
    results returned to autograd are None."""

    @staticmethod
    def forward(ctx, owner, img, drop, *params):
        ctx.owner = owner
        ctx.nparams = len(params)
        eng = owner._engine()
        out = eng.forward(img, train=True, drop=drop)
        ctx.gen = eng._fwd_gen                  # the engine keeps ONE set of saved activations: this node owns it until the next forward
        return out

    @staticmethod
    def backward(ctx, dlogits):
        owner = ctx.owner
        eng = owner._engine()
        # The node hands autograd no input gradients (the engine writes every parameter's .grad itself), so torch.autograd.grad(loss, params)
        # or backward(inputs=...) would come back with None for all of them without a word.  Those calls run the engine with a non-empty
        # execution plan; a plain loss.backward() runs it with an empty one, for which _will_engine_execute_node answers True for any node.
        probe = getattr(torch._C, "_will_engine_execute_node", None)
        node = next((fn for fn, _ in ctx.next_functions if fn is not None), None)
        if probe is not None and node is not None:
            try:
                plain = bool(probe(node))
            except RuntimeError:
                plain = False                   # (a leaf that autograd.grad captures: the probe refuses to answer -- same situation)
            if not plain:
                raise L.GavikoHipError("gaviko_amd models write parameter gradients into .grad themselves: use loss.backward() "
                                       "(torch.autograd.grad(loss, params) / backward(inputs=...) would silently receive None for every "
                                       "parameter); read the gradients from p.grad afterwards")
        if eng._fwd_gen != ctx.gen:
            raise L.GavikoHipError("backward() of a forward whose saved activations were overwritten by a later training-mode forward of the "
                                   "same model (e.g. loss = f(model(x1), model(x2))): the engine keeps one forward's state -- concatenate "
                                   "the inputs into one batch, or call backward() before the next forward")
        named = owner._named_cache()[1]
        token = owner.__dict__.get("_grads_zeroed")
        owner.__dict__["_grads_zeroed"] = None
        if token is not None:
            # model.zero_grad(set_to_none=False) zeroed the flat buffer in ONE launch after verifying that every trainable .grad is its view:
            # the engine overwrites, nothing to walk (the per-parameter bookkeeping below costs the host ~0.6 ms per step).  That was THEN:
            # a torch optimizer's zero_grad() (set_to_none=True) since, or a re-allocated flat buffer (make_reducer / set_bucket_layers, a
            # requires_grad flip), and the engine would write into a buffer no .grad points at -- re-check against the state at backward.
            views = eng._grad_views(dlogits.device)              # (re-allocates when the trainable set changed)
            if eng._flat_grad is token and all(named[n].grad is v for n, v in views.items()):
                eng.backward(dlogits, reducer=owner.__dict__.get("_reducer"))
                return (None, None, None) + (None,) * ctx.nparams
        had_grads = [n for n in eng.trainable_names() if named[n].grad is not None]
        if not had_grads:
            gv = eng.backward(dlogits, reducer=owner.__dict__.get("_reducer"))
            for n, g in gv.items():
                named[n].grad = g
        else:                                   # gradient accumulation: keep what is there, add the new contribution
            old = {n: named[n].grad.clone() for n in had_grads}
            gv = eng.backward(dlogits, reducer=owner.__dict__.get("_reducer"))
            for n, g in gv.items():
                if n in old:
                    g.add_(old[n])
                named[n].grad = g
        return (None, None, None) + (None,) * ctx.nparams


class HotPathModule(nn.Module):
    """Shared plumbing of every --method class: engine construction, the autograd bridge, loud failure off-GPU."""

    _kind = "vit"

    def _engine(self) -> Engine:
        eng = self.__dict__.get("_eng")
        if eng is None:
            depth, heads, dim, mlp_dim = mapping_vit(self._cfg["backbone"])
            cfg = dict(self._cfg)
            prec = self.__dict__.get("_precision") or self.__dict__.get("_kw_precision")
            if prec:
                cfg["precision"] = prec
            eng = Engine(self._kind, cfg, dict(self.named_parameters()), depth, heads, dim, mlp_dim)
            self.__dict__["_eng"] = eng
        return eng

    def _apply(self, fn, *a, **k):
        # .to()/.cuda()/.float() re-create parameter storage: drop the engine (bf16 shadows, workspaces) with it
        self.__dict__.pop("_eng", None)
        self.__dict__.pop("_named", None)
        return super()._apply(fn, *a, **k)

    def _named_cache(self):
        """([(name, parameter)], {name: parameter}) -- walking the module tree costs ~1.3 ms per call at ViT-B (442 tensors), which
        is a fifth of a training step; the Parameter objects themselves only change under _apply / load_state_dict(assign=True)."""
        c = self.__dict__.get("_named")
        if c is None or c[2] != len(self._parameters) + sum(1 for _ in self.children()):
            lst = list(self.named_parameters())
            c = self.__dict__["_named"] = (lst, dict(lst), len(self._parameters) + sum(1 for _ in self.children()))
        return c

    def _drop_config(self) -> dict:
        return {}

    def zero_grad(self, set_to_none: bool = True) -> None:
        """nn.Module.zero_grad.  set_to_none=False on a model whose gradients are already the views of the engine's flat buffer (i.e. after
        the first backward) is ONE memset of that buffer instead of a walk over 442 tensors; the next backward then overwrites it without
        per-parameter bookkeeping.  Everything else (set_to_none=True, the first step, gradients assigned by the caller) takes
        torch's own path."""
        self.__dict__["_grads_zeroed"] = None
        eng = self.__dict__.get("_eng")
        if not set_to_none and eng is not None and eng._flat_grad is not None:
            named, views = self._named_cache()[1], eng._flat_grad["views"]
            names = eng.trainable_names()
            if len(views) == len(names) and all(named[n].grad is views.get(n) for n in names) and \
                    all(p.grad is None for n, p in named.items() if n not in views):
                eng._flat_grad["buf"].zero_()
                self.__dict__["_grads_zeroed"] = eng._flat_grad      # the buffer object this promise holds for (checked again at backward)
                return
        super().zero_grad(set_to_none=set_to_none)

    def _load_backbone(self) -> None:
        """The constructor step of vision_transformer.py:140-145 (and its five copies): converted timm weights loaded with
        strict=False.  Offline: the raw timm state dict is read from ./pretrained/<timm model name> -- where the reference leaves it
        (load_pretrained.py:26-28) -- or from $GAVIKO_PRETRAINED_DIR; when the file is absent the random initialisation stays."""
        from ..utils import load_pretrained as lp
        backbone = self._cfg.get("backbone")
        if backbone is None:
            return
        save_dir = os.environ.get("GAVIKO_PRETRAINED_DIR", "./pretrained")
        path = lp.pretrained_path(backbone, save_dir)
        if path is None or not os.path.exists(path):
            logging.info(f"no pretrained file for {backbone} under {save_dir}: keeping the random initialisation")
            return
        logging.info(f"Loading pretrained {backbone}...")
        new_dict = lp.load_pretrain(backbone, self.num_patches, self._cfg["frame_patch_size"], save_dir)
        self.load_state_dict(new_dict, strict=False)
        logging.info(f"Load pretrained {backbone} sucessfully!")

    def set_precision(self, precision):
        """'bf16' (default): MFMA bf16 operands with fp32 accumulation / residual stream / statistics.  'fp32': exact fp32
        arithmetic throughout -- what the reference computes when config['train']['fp16'] is false (train.py:157).
        Parameters stay fp32 either way.  Also settable with a `precision=` constructor kwarg or GAVIKO_HIP_PRECISION."""
        if precision is not None and str(precision).lower() not in ("bf16", "fp32", "float32"):
            raise L.GavikoHipError(f"precision={precision!r}: expected 'bf16' or 'fp32'")
        self.__dict__["_precision"] = None if precision is None else str(precision).lower()
        self.__dict__.pop("_eng", None)
        return self

    def attach_reducer(self, reducer) -> None:
        """Data parallelism: `reducer` (gaviko_amd.distributed.GradReducer) all-reduces the flat gradient buffer during backward."""
        self.__dict__["_reducer"] = reducer

    def make_reducer(self, layers_per_bucket: int = None, group=None, mode: str = "events"):
        """Attach the data-parallel gradient reducer.  mode 'events' (default): the backward stays one launch plan; every bucket of
        `layers_per_bucket` layers (default 4) is all-reduced behind the event recorded on the stream that finalises it (MWSA chain,
        GPA chain or main stream) -- overlapped with the rest of the backward, no stream joins.  mode 'segments': the backward is cut
        into one plan per bucket (every cut joins the three streams: +0.7 ms per step at ViT-B gaviko, tools/bench_reducer.py)."""
        from ..distributed import GradReducer
        named = dict(self.named_parameters())
        depth = mapping_vit(self._cfg["backbone"])[0]
        if layers_per_bucket is None:
            layers_per_bucket = 4 if mode == "events" else depth
        eng = self._engine()
        eng.set_bucket_layers(layers_per_bucket)
        names = eng.flat_names()                      # the flat buffer's own order: every ready group of buckets is one contiguous slice
        r = GradReducer(names, [named[n].numel() for n in names], depth, self._cfg.get("share_factor", 1), layers_per_bucket, group, mode)
        self.attach_reducer(r)
        return r

    def _run(self, img: torch.Tensor) -> torch.Tensor:
        if not isinstance(img, torch.Tensor) or not img.is_cuda:
            raise L.GavikoHipError("gaviko_amd runs on MI355X only: move the model and the input to the HIP device "
                                   "(there is no CPU fallback)")
        if torch.is_grad_enabled():
            # ONE trainable tensor ties the node into the autograd graph (the engine writes every parameter gradient itself and the node
            # returns None for its inputs): handing all ~200 of them to Function.apply cost the host 0.3 ms per step in argument
            # processing and as much again in the backward's bookkeeping (tools/host_split.py)
            anchor = next((p for _, p in self._named_cache()[0] if p.requires_grad), None)
            if anchor is not None:
                return _HotPathFn.apply(self, img, self._drop_config(), anchor)
        # no autograd: nothing is saved for a backward, but modules left in training mode still drop (nn.Dropout follows .training)
        return self._engine().forward(img, train=False, drop=self._drop_config())


class VisionTransformer(HotPathModule):
    """vision_transformer.py:91-164.  `linear` / `bitfit` / `fft` select which parameters train (train.py:117-137)."""

    _kind = "vit"

    def __init__(self, *, image_size, image_patch_size, frames, frame_patch_size, num_classes, pool="cls", channels=3,
                 dim_head=64, dropout=0.0, emb_dropout=0.0, backbone=None, **kwargs):
        super().__init__()
        depth, heads, dim, mlp_dim = mapping_vit(backbone)
        ih, iw = pair(image_size)
        ph, pw = pair(image_patch_size)
        assert ih % ph == 0 and iw % pw == 0, "Image dimensions must be divisible by the patch size."
        assert frames % frame_patch_size == 0, "Frames must be divisible by frame patch size"
        assert pool in {"cls", "mean"}, "pool type must be either cls (cls token) or mean (mean pooling)"
        self.num_patches = (ih // ph) * (iw // pw) * (frames // frame_patch_size)
        self.image_size, self.image_patch_size = image_size, image_patch_size
        self.frames, self.frame_patch_size, self.depth = frames, frame_patch_size, depth
        self.conv_proj = nn.Sequential(nn.Conv3d(channels, dim, kernel_size=(frame_patch_size, image_patch_size, image_patch_size),
                                                 stride=(frame_patch_size, image_patch_size, image_patch_size)))
        self.pos_embedding = nn.Parameter(torch.randn(1, self.num_patches + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Linear(dim, num_classes)
        # (the reference fetches timm weights here, vision_transformer.py:140-145: outside the hot path, needs network)
        self.__dict__["_kw_precision"] = kwargs.get("precision")      # "bf16" (default) | "fp32": see HotPathModule.set_precision
        self._cfg = dict(image_size=ih, image_patch_size=ph, frames=frames, frame_patch_size=frame_patch_size, num_classes=num_classes,
                         pool=pool, channels=channels, dim_head=dim_head, backbone=backbone, dropout=dropout, emb_dropout=emb_dropout)
        self._load_backbone()

    def _drop_config(self):
        # no train() override (vision_transformer.py:91-164): every nn.Dropout follows module.training
        return {"dropout": self._cfg["dropout"] if self.transformer.layers[0][0].dropout.training else 0.0,
                "emb_dropout": self._cfg["emb_dropout"] if self.dropout.training else 0.0}

    def forward(self, img):
        return self._run(img)
