"""ORACLE -- test infrastructure only.  Loss seeds of the training step (caller side of the hot path)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def focal_loss(logits: torch.Tensor, target: torch.Tensor, gamma: float = 1.2, eps: float = 1e-16) -> torch.Tensor:
    """FocalLoss.forward as it actually executes (losses/focal_loss.py:84-115): the live
    _process_preds (84-91) clamps to [eps, 1-eps] THEN softmaxes, and forward calls it twice (94, 102);
    pt = prob of the target class; loss = mean((1-pt)^gamma * -log(eps+pt)); weights=None, no ignored rows."""
    x = torch.softmax(torch.clamp(logits, eps, 1 - eps), dim=-1)
    x = torch.softmax(torch.clamp(x, eps, 1 - eps), dim=-1)
    pt = x.gather(-1, target.view(-1, 1)).squeeze(-1)
    loss = (1 - pt) ** gamma * -torch.log(eps + pt)
    return loss.sum() / target.numel()


def cross_entropy(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """train.py:178-179 alternative."""
    return F.cross_entropy(logits, target)
