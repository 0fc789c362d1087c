"""Loss seeds of the training step as ONE launch each (forward + gradient), with the constructor surface of the reference.

* `FocalLoss` mirrors `losses/focal_loss.py:15-118` -- same arguments, same defaults, and the behaviour the class actually
  has: the live `_process_preds` (84-91) clamps to [eps, 1-eps] and softmaxes, and `forward` (92-110) calls it twice.
* `CrossEntropyLoss` is the `train.py:178-179` alternative (torch.nn.CrossEntropyLoss, mean reduction, ignore_index -100).
* `StepMeter` replaces the two per-step host reads of `train.py:327-328` (`running_loss += loss.item() * B`,
  `num_acc += (argmax == labels).sum().item()`): the loss kernel accumulates both on the device; read once per epoch.

The kernels live in `csrc/loss.hip` (`gvk_loss_fwd_bwd`); there is no torch fallback.
"""
from __future__ import annotations

from typing import Optional, Union

import torch
from torch import Tensor, nn

from . import ops


class StepMeter:
    """Device-side running sums of an epoch: loss * batch, correct argmax predictions, samples."""

    def __init__(self, device):
        self.buf = torch.zeros(3, dtype=torch.float32, device=device)

    def reset(self) -> None:
        self.buf.zero_()

    def read(self):
        """(mean loss, accuracy, samples) -- the one host read (train.py:330-331 do this division per epoch)."""
        loss_sum, correct, n = self.buf.tolist()
        n = max(n, 1.0)
        return loss_sum / n, correct / n, int(n)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, mod):
        x = logits.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        none = mod.reduction == "none"
        loss = torch.empty(x.shape[0] if none else 1, dtype=torch.float32, device=x.device)
        dlogits = torch.empty_like(x)
        ops.loss_fwd_bwd(x, target.contiguous().view(-1), loss, dlogits, mod._kind, gamma=float(mod._gamma), eps=float(mod.eps),
                         ignore_index=int(mod.ignore_index), weights=mod._weights_on(x.device), meter=mod.meter.buf if mod.meter is not None else None,
                         reduction=mod.reduction)
        ctx.save_for_backward(dlogits)
        ctx.in_dtype = logits.dtype
        ctx.none = none
        return loss if none else loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        (dlogits,) = ctx.saved_tensors
        if ctx.none:                                     # reduction='none': row i of dlogits is d loss_i / d logits_i
            return (dlogits * grad_out.reshape(-1, 1)).to(ctx.in_dtype), None, None
        return (dlogits * grad_out).to(ctx.in_dtype), None, None


class _FusedLoss(nn.Module):
    _kind = ops.LOSS_CE
    _gamma = 0.0
    eps = 1e-16
    ignore_index = -100
    reduction = "mean"
    weights: Optional[Tensor] = None
    meter: Optional[StepMeter] = None

    def attach_meter(self, meter: Optional[StepMeter]):
        self.meter = meter
        return self

    def _weights_on(self, device):
        if self.weights is None:
            return None
        if self.weights.device != device or self.weights.dtype != torch.float32:
            self.weights = self.weights.to(device=device, dtype=torch.float32).contiguous()
        return self.weights

    def forward(self, x: Tensor, target: Tensor) -> Tensor:
        if x.dim() != 2 or x.shape[-1] < 2:
            raise NotImplementedError("the fused loss takes [batch, classes >= 2] logits (train.py feeds [B, num_classes])")
        return _LossFn.apply(x, target, self)


class FocalLoss(_FusedLoss):
    _kind = ops.LOSS_FOCAL

    def __init__(self, gamma, weights: Union[None, Tensor] = None, reduction: str = "mean", ignore_index=-100, eps=1e-16, fp16: bool = False) -> None:
        super().__init__()
        if reduction not in ["mean", "none", "sum"]:
            raise NotImplementedError("Reduction {} not implemented.".format(reduction))
        assert weights is None or isinstance(weights, Tensor), "weights should be of type Tensor or None, but {} given".format(type(weights))
        self.dtype = torch.float16 if fp16 else torch.float32
        self.reduction = reduction
        self.gamma = gamma
        self._gamma = gamma
        self.ignore_index = ignore_index
        self.eps = eps
        self.weights = weights


class CrossEntropyLoss(_FusedLoss):
    _kind = ops.LOSS_CE

    def __init__(self, weight: Optional[Tensor] = None, ignore_index: int = -100, reduction: str = "mean") -> None:
        super().__init__()
        if reduction not in ("mean", "sum", "none"):
            raise NotImplementedError("Reduction {} not implemented.".format(reduction))
        self.weights = weight
        self.ignore_index = ignore_index
        self.reduction = reduction
