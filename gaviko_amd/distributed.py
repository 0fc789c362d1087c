"""Data parallelism for the hot path: one process per GPU, the minibatch sharded by sample index, and ONE exchange
per step -- a mean all-reduce of the flat fp32 buffer of trainable gradients (3.6 MB for ViT-B gaviko) over
RCCL/xGMI.  The reference has no data parallelism of its own; equivalence is pinned by the cfg3
fixture (8 shards x 4 == 1 x 32, mean-reduced).

The buffer is cut into a few layer-aligned buckets.  A bucket is reduced on the collective's own stream as soon as the
backward sweep has written its last gradient (the sweep runs from the last layer to the first, so buckets complete in
reverse order), overlapping the collectives with the remaining backward kernels; the job is latency-bound, not
bandwidth-bound, hence few buckets.

Ordering (default, `mode="events"`): the backward stays ONE recorded launch plan.  The engine records an event on the
stream that finalises a bucket -- the MWSA chain for `local_attns.*`, the GPA chain for `prompt_projs.*`, the main
stream for everything else -- and the collective stream waits for exactly that event (gvk_plan_event_stream_wait), so
no bucket boundary joins the three streams.  `mode="segments"` is the older form (the backward cut into one plan per
bucket, every cut a three-stream join: +0.7 ms per step at ViT-B), kept for A/B.  The bucket planner and the reduction
are device-agnostic so the N>1 path is covered by world_size-2 `gloo` tests on CPU.
"""
from __future__ import annotations

import os
import re
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def stream_kind(name: str) -> str:
    """Which of the engine's three streams writes the gradient of `name` last (engine.py: MWSA chain / GPA chain / main)."""
    if ".local_attns." in "." + name:
        return "loc"
    if ".prompt_projs." in "." + name:
        return "gpa"
    return "main"


_LAYER_PAT = re.compile(r"\.(?:local_attns|prompt_projs|layers)\.(\d+)\.")


def ready_layer(name: str, share_factor: int = 1, layers_per_bucket: int = 4) -> int:
    """The layer whose backward completes the bucket `name` belongs to (-1: tensors without a layer index -- prompts, head -- final only
    at the end of the sweep)."""
    m = _LAYER_PAT.search("." + name)
    if m is None:
        return -1
    first_layer = int(m.group(1)) * (share_factor if ("local_attns" in name or "prompt_projs" in name) else 1)
    return (first_layer // layers_per_bucket) * layers_per_bucket


def flat_order(names: Sequence[str], share_factor: int = 1, layers_per_bucket: int = 4) -> List[str]:
    """`names` in the order the engine lays the flat gradient buffer out: grouped by completion order of the backward sweep (highest
    ready layer first, unindexed tensors last), the given order kept inside a group (stable).  Every group of buckets that becomes final
    together is then ONE contiguous slice of the buffer -- one plain all_reduce per group, public API only."""
    key = lambda n: (lambda r: (1, 0) if r < 0 else (0, -r))(ready_layer(n, share_factor, layers_per_bucket))
    return sorted(names, key=key)


def plan_buckets(names: Sequence[str], numels: Sequence[int], depth: int, share_factor: int = 1, layers_per_bucket: int = 4, kinds: bool = False):
    """-> list of (ready_layer, start, end) element ranges of the flat buffer, sorted by the order they become ready.

    A tensor of module index s (…local_attns.s… / …prompt_projs.s… / …layers.i…) is final once the backward sweep has
    finished layer  s * share_factor  (the lowest layer using it).  Tensors without a layer index (prompts, head) are
    final at the very end (ready_layer = -1).  Consecutive tensors with the same bucket id are merged into one range.
    """
    offs, o = [], 0
    for n in numels:
        offs.append(o)
        o += n
    ranges: List[Tuple[int, int, int]] = []
    rkinds: List[str] = []
    for name, off, n in zip(names, offs, numels):
        ready = ready_layer(name, share_factor, layers_per_bucket)              # bucket completes at its lowest layer
        kind = stream_kind(name) if ready >= 0 else "main"
        if ranges and ranges[-1][0] == ready and ranges[-1][2] == off and rkinds[-1] == kind:
            ranges[-1] = (ready, ranges[-1][1], off + n)
        else:
            ranges.append((ready, off, off + n))
            rkinds.append(kind)
    order = sorted(range(len(ranges)), key=lambda i: (-ranges[i][0] if ranges[i][0] >= 0 else 1, ranges[i][1]))   # high layers first, unindexed last
    ranges, rkinds = [ranges[i] for i in order], [rkinds[i] for i in order]
    assert sum(e - s for _, s, e in ranges) == o
    return (ranges, rkinds) if kinds else ranges


class GradReducer:
    """Bucketed mean all-reduce of a flat gradient buffer."""

    def __init__(self, names: Sequence[str], numels: Sequence[int], depth: int, share_factor: int = 1, layers_per_bucket: int = 4,
                 group=None, mode: str = "events"):
        if mode not in ("events", "segments"):
            raise ValueError("GradReducer: mode must be 'events' or 'segments'")
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a single rank has nothing to exchange; GAVIKO_DP_FORCE_COLLECTIVES=1 issues the collectives anyway (identity over one rank) so that the
        # whole ordering machinery -- and RCCL itself -- can be exercised on a one-GPU box (tests/test_distributed_gpu.py)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("GAVIKO_DP_FORCE_COLLECTIVES", "0") == "1")
        self.group = group
        self.mode = mode
        self.ranges, self.kinds = plan_buckets(names, numels, depth, share_factor, layers_per_bucket, kinds=True)
        self._pending: List = []
        self._stream: Optional[torch.cuda.Stream] = None
        self._done_upto = None

    def _backend(self) -> str:
        try:
            return str(dist.get_backend(self.group))
        except Exception:
            return "none"

    @staticmethod
    def _merge_adjacent(pieces):
        """Slices of one buffer that sit back to back become one slice (the engine lays the flat buffer out by completion group --
        flat_order -- so a group of buckets that is final together is ONE slice; any other layout still works, with more collectives)."""
        out = []
        for t in sorted(pieces, key=lambda t: t.data_ptr()):
            if out and out[-1].data_ptr() + out[-1].numel() * out[-1].element_size() == t.data_ptr() and out[-1].dtype == t.dtype:
                prev = out[-1]
                out[-1] = torch.as_strided(prev, (prev.numel() + t.numel(),), (1,))     # same storage, same offset, both slices
            else:
                out.append(t)
        return out

    def _mean_pieces(self, pieces):
        """Mean all-reduce of a group of slices of the flat buffer: adjacent slices are merged first, so with the engine's layout this is
        ONE plain all_reduce.  RCCL applies the 1/world factor inside the collective kernel (ReduceOp.AVG: no separate scaling launch
        on the collective stream); gloo (CPU tests, one-GPU rehearsals) has no AVG, so it sums and scales.  For a power-of-two world both
        forms give the same bits (x / 2^k is exact)."""
        if not pieces:
            return
        pieces = self._merge_adjacent(pieces)
        if pieces[0].is_cuda and self._backend() == "nccl":
            for t in pieces:
                dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
            return
        for t in pieces:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.mul_(1.0 / self.world)

    def _reduce(self, flat: torch.Tensor, s: int, e: int):
        if not self.active:
            return
        piece = flat[s:e]
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._stream):
                self._stream.wait_event(ev)
                self._mean_pieces([piece])
        else:
            self._mean_pieces([piece])

    def ready_groups(self):
        """Buckets that become final together, as [(ready_layer, [bucket indices])] in completion order.  Every stream kind's bucket of
        one ready layer goes into one group, and the buckets of the LOWEST ready layer are merged with the unindexed tensors (prompts,
        head: ready = -1): both are final only at the very end of the sweep, so one collective serves them (3 launches per step at depth 12
        with 4 layers per bucket instead of 7)."""
        by = {}
        for idx, (ready, _, _) in enumerate(self.ranges):
            by.setdefault(ready, []).append(idx)
        layers = sorted((r for r in by if r >= 0), reverse=True)
        groups = [(r, by[r]) for r in layers]
        if -1 in by:
            if groups and groups[-1][0] == min(layers):
                groups[-1] = (-1, groups[-1][1] + by[-1])
            else:
                groups.append((-1, by[-1]))
        return groups

    def reduce_marked(self, flat: torch.Tensor, marks: dict, waiter) -> None:
        """mode 'events': every group of buckets is all-reduced on the collective stream behind the events the engine recorded for its
        members.  marks: {(kind, ready_layer): event handle}; waiter(stream, handle) makes `stream` wait for that event.  Called once per
        step, after the whole backward has been ENQUEUED -- the GPU still runs it, and each collective starts when its buckets are final."""
        if not self.active:
            return
        if flat.is_cuda and self._stream is None:
            self._stream = torch.cuda.Stream(device=flat.device)
        for _, members in self.ready_groups():
            pieces = [flat[self.ranges[i][1]: self.ranges[i][2]] for i in members]
            if not flat.is_cuda:
                self._mean_pieces(pieces)
                continue
            keys = []
            for i in members:
                key = (self.kinds[i], self.ranges[i][0])
                key = key if key in marks else ("main", -1)                        # fall back to the end-of-backward event
                if key not in keys:
                    keys.append(key)
            with torch.cuda.stream(self._stream):
                for key in keys:
                    waiter(self._stream, marks[key])
                self._mean_pieces(pieces)
        if flat.is_cuda and self._stream is not None:
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)

    def begin(self):
        self._next = 0

    def layer_done(self, flat: torch.Tensor, layer: int):
        """Call after the backward sweep finished `layer` (counting down): reduces every bucket that is now final."""
        while self._next < len(self.ranges) and self.ranges[self._next][0] >= 0 and self.ranges[self._next][0] >= layer:
            _, s, e = self.ranges[self._next]
            self._reduce(flat, s, e)
            self._next += 1

    def finish(self, flat: torch.Tensor):
        """Reduce what is left (prompts, head) and make the compute stream wait for the side stream."""
        while self._next < len(self.ranges):
            _, s, e = self.ranges[self._next]
            self._reduce(flat, s, e)
            self._next += 1
        if flat.is_cuda and self._stream is not None:
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Samples [lo, hi) of the global batch owned by `rank` (even split)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    b = global_batch // world
    return rank * b, (rank + 1) * b
