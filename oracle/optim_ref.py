"""ORACLE (test infrastructure only -- never imported by the product path).

The reference's optimisation step, restated as the exact torch calls it makes (train.py:185-206, 315-319):
Adam over the trainable tensors, OneCycleLR with the config's scheduler block, clip_grad_norm_(params, 1.0) before each step.
torch.optim is a third-party dependency of the reference (requirements.txt: torch==2.7.1; this container 2.10.0) -- its CPU
implementation is the arithmetic being matched.
"""
from __future__ import annotations

import torch


def make(params, *, lr, eps, max_lr, total_steps, pct_start, div_factor, final_div_factor, anneal_strategy="cos", three_phase=False):
    """-> (optimizer, scheduler) as train.py:185-206 builds them."""
    opt = torch.optim.Adam(params, lr=lr, eps=eps)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr, total_steps=total_steps, pct_start=pct_start, div_factor=div_factor,
                                              final_div_factor=final_div_factor, anneal_strategy=anneal_strategy, three_phase=three_phase)
    return opt, sch


def step(params, opt, sch, max_norm=1.0):
    """train.py:315-319; returns the pre-clip gradient norm."""
    n = torch.nn.utils.clip_grad_norm_(params, max_norm)
    opt.step()
    sch.step()
    return n
