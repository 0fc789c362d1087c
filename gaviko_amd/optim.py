"""Fused optimisation step around the hot path.

The reference's step after `loss.backward()` is (train.py:185-206, 315-319)

    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    optimizer.step()          # torch.optim.Adam(trainable, lr, eps=1e-8 fp32 / 1e-4 half)
    scheduler.step()          # OneCycleLR(max_lr, total_steps, pct_start, div_factor, final_div_factor, 'cos', three_phase)

i.e. ~300 per-tensor norm kernels and a host sync, a foreach Adam chain over 304 tensors and a Python scheduler.  Here the
gradients already sit in the engine's flat fp32 buffer, so the whole step is three launches (`gvk_sumsq` x2 stages,
`gvk_adam_step`) with no host synchronisation; the OneCycle schedule (learning rate AND the beta1 cycling that OneCycleLR
applies to Adam by default) is mirrored on the host and handed to the kernel as plain arguments.

`FusedAdamOneCycle` is used like the optimizer + scheduler pair it replaces:

    opt = FusedAdamOneCycle(model, lr=cfg.lr, eps=1e-8, max_lr=3e-4, total_steps=N, pct_start=.3, div_factor=10,
                            final_div_factor=1000)
    loss.backward(); opt.step(); opt.zero_grad()
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import torch

from . import lib as L

BLOCK = 1024          # elements per workgroup of gvk_adam_step (csrc/optim.hip)


class OneCycle:
    """The values torch.optim.lr_scheduler.OneCycleLR sets on an Adam optimizer (lr and, with cycle_momentum, beta1) at
    scheduler step `t` (t = number of scheduler.step() calls so far), restated from its documented two/three-phase schedule."""

    def __init__(self, max_lr, total_steps, pct_start=0.3, div_factor=25.0, final_div_factor=1e4, anneal_strategy="cos",
                 three_phase=False, cycle_momentum=True, base_momentum=0.85, max_momentum=0.95):
        if total_steps <= 0 or not 0 <= pct_start <= 1:
            raise ValueError("OneCycle: total_steps must be positive and pct_start within [0, 1]")
        if anneal_strategy not in ("cos", "linear"):
            raise ValueError("OneCycle: anneal_strategy must be 'cos' or 'linear'")
        self.total_steps, self.cos = int(total_steps), anneal_strategy == "cos"
        self.cycle_momentum = cycle_momentum
        initial, minimum = max_lr / div_factor, max_lr / div_factor / final_div_factor
        if three_phase:
            self.phases = [(float(pct_start * total_steps) - 1, initial, max_lr, max_momentum, base_momentum),
                           (float(2 * pct_start * total_steps) - 2, max_lr, initial, base_momentum, max_momentum),
                           (total_steps - 1, initial, minimum, max_momentum, max_momentum)]
        else:
            self.phases = [(float(pct_start * total_steps) - 1, initial, max_lr, max_momentum, base_momentum),
                           (total_steps - 1, max_lr, minimum, base_momentum, max_momentum)]

    def _anneal(self, start, end, pct):
        if self.cos:
            return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1)
        return (end - start) * pct + start

    def at(self, t: int):
        """-> (lr, beta1 or None) in effect after t scheduler steps."""
        if t >= self.total_steps:
            raise ValueError(f"OneCycle: step {t} beyond total_steps={self.total_steps}")
        start_step = 0.0
        lr = mom = None
        for i, (end_step, lr0, lr1, m0, m1) in enumerate(self.phases):
            if t <= end_step or i == len(self.phases) - 1:
                pct = (t - start_step) / (end_step - start_step)
                lr, mom = self._anneal(lr0, lr1, pct), self._anneal(m0, m1, pct)
                break
            start_step = end_step
        return lr, (mom if self.cycle_momentum else None)


class FusedAdamOneCycle:
    """clip_grad_norm_(max_norm) + Adam + OneCycleLR for the trainable tensors of a gaviko_amd model, or for an explicit
    (params, flat_grad) pair whose gradient layout is the concatenation of the params in order.  In the (params, flat_grad) form the
    caller must call `engine.invalidate_weights()` after step() if any of the params is a backbone weight of a gaviko_amd model."""

    def __init__(self, model_or_params, *, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm: Optional[float] = 1.0,
                 max_lr=None, total_steps=None, pct_start=0.3, div_factor=25.0, final_div_factor=1e4, anneal_strategy="cos",
                 three_phase=False, cycle_momentum=True, base_momentum=0.85, max_momentum=0.95, flat_grad: torch.Tensor = None):
        self.model = None
        if hasattr(model_or_params, "_engine"):
            self.model = model_or_params
            self._params = None
        else:
            self._params = list(model_or_params)
            if flat_grad is None:
                raise L.GavikoHipError("FusedAdamOneCycle(params, ...): pass flat_grad=, the flat fp32 gradient buffer of those params")
        self._flat = flat_grad
        self.beta1, self.beta2, self.eps, self.lr0 = float(betas[0]), float(betas[1]), float(eps), float(lr)
        self.max_norm = max_norm
        self.schedule = None if max_lr is None else OneCycle(max_lr, total_steps, pct_start, div_factor, final_div_factor, anneal_strategy,
                                                             three_phase, cycle_momentum, base_momentum, max_momentum)
        self.t = 0                      # optimizer steps taken == scheduler steps taken
        self._tabs = None
        self.m = self.v = None
        self._layout = None             # [(name, numel)] in the order m / v are stored in (= the flat gradient buffer's at the last bind)

    # ---- tables
    def _bind(self):
        if self.model is not None:
            eng = self.model._engine()
            if eng.flat_grad is None:
                raise L.GavikoHipError("FusedAdamOneCycle.step(): run a backward first (the engine owns the flat gradient buffer)")
            named = dict(self.model.named_parameters())
            names = list(eng.flat_names())                             # the flat gradient buffer's layout
            params = [named[n] for n in names]
            flat = eng.flat_grad
        else:
            params, flat = self._params, self._flat
            names = [f"#{i}" for i in range(len(params))]
        sig = (flat.data_ptr(),) + tuple(p.data_ptr() for p in params)
        if self._tabs is not None and self._tabs["sig"] == sig:
            return self._tabs
        if any(p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() for p in params) or flat.dtype != torch.float32:
            raise L.GavikoHipError("FusedAdamOneCycle: parameters and the flat gradient must be contiguous fp32 tensors on the HIP device")
        total = sum(p.numel() for p in params)
        if flat.numel() < total:
            raise L.GavikoHipError("FusedAdamOneCycle: flat gradient buffer smaller than the parameters")
        rows, off = [], 0
        for tid, p in enumerate(params):
            n = p.numel()
            for o in range(0, n, BLOCK):
                rows.append((tid, o, off + o, min(BLOCK, n - o)))
            off += n
        dev = flat.device
        tabs = dict(sig=sig, total=total, nblocks=len(rows), flat=flat,
                    ptr=torch.tensor([p.data_ptr() for p in params], dtype=torch.int64, device=dev),
                    blk=torch.tensor(rows, dtype=torch.int32, device=dev).contiguous(),
                    scratch=torch.zeros(256, device=dev), norm_sq=torch.zeros(1, device=dev))
        layout = [(n, p.numel()) for n, p in zip(names, params)]
        if self.m is None:
            self.m, self.v = torch.zeros(total, device=dev), torch.zeros(total, device=dev)
        else:
            # The moments are positional in the flat layout, and that layout moves: make_reducer(mode=...) / set_bucket_layers regroup the
            # buffer by completion bucket, a checkpoint may come from another reducer mode.  Follow the names, never the positions.
            self.m, self.v = (self._relayout(t.to(dev), self._layout, layout) for t in (self.m, self.v))
        self._layout = layout
        self._tabs = tabs
        return tabs

    @staticmethod
    def _relayout(t: torch.Tensor, old, new) -> torch.Tensor:
        """Moment vector stored in layout `old` ([(name, numel)]) -> layout `new`.  Same tensors in another order are permuted; anything
        else (a tensor missing, added or resized) is an error -- silently keeping positions would pair moments with the wrong parameters."""
        if old is None:
            raise L.GavikoHipError("FusedAdamOneCycle: optimizer state without a recorded layout")
        if old == new:
            return t
        where, off = {}, 0
        for n, k in old:
            where[n] = (off, k)
            off += k
        if off != t.numel() or sorted(old) != sorted(new):
            only_old = sorted(set(dict(old)) - set(dict(new)))[:3]
            only_new = sorted(set(dict(new)) - set(dict(old)))[:3]
            raise L.GavikoHipError(f"FusedAdamOneCycle: the optimizer state covers other tensors than the model trains now "
                                   f"(state only: {only_old}, model only: {only_new}, or sizes differ): cannot carry Adam moments over")
        return torch.cat([t[where[n][0]: where[n][0] + k] for n, k in new]) if new else t

    # ---- the step
    def current(self):
        """(lr, beta1) the next step() will use."""
        if self.schedule is None:
            return self.lr0, self.beta1
        lr, mom = self.schedule.at(self.t)
        return lr, (self.beta1 if mom is None else mom)

    def step(self) -> None:
        L.require_device()
        tb = self._bind()
        lib, st = L.load(), L.stream_ptr()
        lr, b1 = self.current()
        self.t += 1
        norm_ptr = None
        if self.max_norm is not None:
            L.check(lib.gvk_sumsq(L.ptr(tb["flat"]), tb["total"], L.ptr(tb["scratch"]), L.ptr(tb["norm_sq"]), st), "gvk_sumsq")
            norm_ptr = L.ptr(tb["norm_sq"])
        d = L.AdamDesc(ptr_tab=L.ptr(tb["ptr"]), blk_tab=L.ptr(tb["blk"]), grad=L.ptr(tb["flat"]), m=L.ptr(self.m), v=L.ptr(self.v),
                       norm_sq=norm_ptr, nblocks=tb["nblocks"], lr=lr, beta1=b1, beta2=self.beta2, eps=self.eps,
                       # torch.optim.Adam: bias_correction1 = 1 - beta1 ** step with the group's CURRENT beta1 (which OneCycleLR cycles)
                       bias_c1=1.0 - b1 ** self.t, bias_c2=1.0 - self.beta2 ** self.t,
                       max_norm=float(self.max_norm if self.max_norm is not None else 0.0))
        L.check(lib.gvk_adam_step(C.byref(d), st), "gvk_adam_step")
        if self.model is not None:
            # the kernel writes parameters through raw pointers, so torch's version counters do not move: tell the engine, which keeps
            # bf16 / transposed operand shadows of the backbone weights (`--method fft` trains them)
            eng = self.model._engine()
            eng.invalidate_weights(eng.trainable_names())

    def grad_norm(self) -> torch.Tensor:
        """Device scalar: the pre-clip total gradient norm of the last step (what clip_grad_norm_ returns)."""
        return self._tabs["norm_sq"].sqrt()

    def get_last_lr(self) -> List[float]:
        return [self.current()[0]]

    def zero_grad(self, set_to_none: bool = True) -> None:
        """torch.optim.Optimizer.zero_grad.  Bound to a model it is the model's own zero_grad, so that set_to_none=False is the ONE memset of
        the flat gradient buffer (HotPathModule.zero_grad) and the following backward skips the per-tensor bookkeeping."""
        if self.model is not None:
            self.model.zero_grad(set_to_none=set_to_none)
            return
        for p in self._params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def state_dict(self):
        """`names` / `numels` say which tensor every stretch of exp_avg / exp_avg_sq belongs to (the flat layout at save time)."""
        return {"t": self.t, "exp_avg": None if self.m is None else self.m.clone(),
                "exp_avg_sq": None if self.v is None else self.v.clone(),
                "names": None if self._layout is None else [n for n, _ in self._layout],
                "numels": None if self._layout is None else [k for _, k in self._layout]}

    def load_state_dict(self, sd):
        """The moments are re-ordered to the current flat layout at the next step (by name); a state saved before names were recorded
        (round <= 4) is taken to be in `named_parameters()` order of the trainable tensors, which is what those rounds stored."""
        self.t = int(sd["t"])
        self._tabs = None
        if sd["exp_avg"] is None:
            self.m = self.v = self._layout = None
            return
        self.m, self.v = sd["exp_avg"].clone(), sd["exp_avg_sq"].clone()
        if sd.get("names") is not None:
            self._layout = list(zip(sd["names"], (int(k) for k in sd["numels"])))
        elif self.model is not None:
            eng = self.model._engine()
            self._layout = [(n, eng.p[n].numel()) for n in eng.trainable_names()]
        else:
            self._layout = [(f"#{i}", p.numel()) for i, p in enumerate(self._params)]
        if sum(k for _, k in self._layout) != self.m.numel():
            raise L.GavikoHipError("FusedAdamOneCycle.load_state_dict: exp_avg does not have the size its layout says")
