"""ORACLE -- test infrastructure only.  Loss seeds of the training step (caller side of the hot path)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def focal_loss(logits: torch.Tensor, target: torch.Tensor, gamma: float = 1.2, eps: float = 1e-16, weights=None, ignore_index: int = -100,
               reduction: str = "mean") -> torch.Tensor:
    """FocalLoss.forward as it actually executes (losses/focal_loss.py:84-115): the live
    _process_preds (84-91) clamps to [eps, 1-eps] THEN softmaxes, and forward calls it twice (94, 102);
    pt = prob of the target class (0 on ignored rows, 77-83); loss_i = w_i (1-pt)^gamma * -log(eps+pt), zero on ignored rows (107);
    w_i = weights[target_i] or 1 (54-58); 'mean' divides by the weight sum of the rows that are not ignored (112-116); 'none' returns the vector (117-118)."""
    x = torch.softmax(torch.clamp(logits, eps, 1 - eps), dim=-1)
    x = torch.softmax(torch.clamp(x, eps, 1 - eps), dim=-1)
    mask = target.view(-1) == ignore_index
    tgt = torch.where(mask, torch.zeros_like(target.view(-1)), target.view(-1))
    pt = x.gather(-1, tgt.view(-1, 1)).squeeze(-1) * (~mask)
    w = torch.ones(tgt.shape[0]) if weights is None else weights[tgt]
    nll = (-torch.log(eps + pt)).masked_fill(mask, 0)
    loss = w * (1 - pt) ** gamma * nll
    if reduction == "sum":
        return loss.sum()
    if reduction == "none":                                   # focal_loss.py:117-118: the per-sample vector (ignored rows: 0)
        return loss
    return loss.sum() / ((~mask) * w).sum()


def cross_entropy(logits: torch.Tensor, target: torch.Tensor, weight=None, ignore_index: int = -100, reduction: str = "mean") -> torch.Tensor:
    """train.py:178-179 alternative."""
    return F.cross_entropy(logits, target, weight=weight, ignore_index=ignore_index, reduction=reduction)
