"""Launch plans (forward + backward) of the 3D-ViT hot path on MI355X.

One `Engine` per model instance.  It owns
  * bf16 MFMA-operand shadows of the frozen fp32 weights, W and W^T (the dgrad of a frozen Linear is an NT GEMM against
    the pre-transposed weight; frozen weights need no wgrad and their GEMM inputs are never saved),
  * the per-batch-size workspace (fp32 residual streams per layer, bf16 GEMM operands, saved statistics),
  * the ordered list of C-ABI launches that make up forward() and backward().
Everything is enqueued on torch's current HIP stream; nothing synchronises, so a whole step can be captured in a HIP
graph.  There is no CPU / eager fallback.

Data layout in HBM (B samples, T tokens, C channels, M = B*T):
  G[i], G1[i]   fp32 [pad128(M)][C]   global residual stream entering layer i / after its attention block
  Lc[i]         fp32 [B*N][C]         GAViKO local stream entering layer i
  xn / act / dpre ... bf16 [pad128(M)][*]  MFMA operands (transient, reused by every layer)
  qkv[i], ctx[i], pre[i]  bf16        saved for the backward (flash attention recompute, GELU')
Reference call structure mirrored here: gaviko.py:291-306 (layer loop), 531-552 (embedding + head).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import lib as L
from . import ops

import os

# How a step is issued after the GRAPH_WARMUP eager passes (env GAVIKO_HIP_GRAPHS):
#   "plan"  (default, also "1") -- the library's launch plan: recorded once, replayed from one C loop on the real streams
#   "graph"                     -- one captured hipGraph per pass (kept for comparison: its executor serialises three forked
#                                  branches on this runtime, tools/probe/probe_streams.hip: 7.6 ms vs 3.9 ms eager)
#   "0" / "eager"               -- every launch from Python
_MODE = os.environ.get("GAVIKO_HIP_GRAPHS", "plan")
STEP_MODE = {"1": "plan", "0": "eager"}.get(_MODE, _MODE)
USE_GRAPHS = STEP_MODE != "eager"
GRAPH_WARMUP = 2
PLAN_TIMING = os.environ.get("GAVIKO_HIP_PLAN_TIMING") is not None
# Timing ablations (tools/ablate_streams.py) -- the RESULTS ARE WRONG with either switch; they only answer "where does the step go":
#   nowait: the main stream skips its waits on the side chains;  noside: the MWSA / GPA chains are not launched at all.
_ABLATE = set(filter(None, os.environ.get("GAVIKO_HIP_ABLATE", "").split(",")))


def _on(tag: str) -> bool:
    """False when the timing ablation `tag` is switched on (GAVIKO_HIP_ABLATE, diagnostics only: bench.py refuses it without --allow-ablate)."""
    return tag not in _ABLATE
# Site seeds of the backbone's own nn.Dropout modules (added to the device epoch word): embedding, VPT prompts of layer i, and per layer
# {+0 attention probabilities, +1 to_out, +2 after GELU, +3 after fc2}.  The MWSA sites use 2*i and 2*i + 1.
SEED_EMB, SEED_PROMPT, SEED_LAYER = 900, 950, 1000

# bench.py instrumentation: when a dict, every GEMM launch is bracketed by HIP events recorded on the launch stream
# bench.py instrumentation: when set to a dict, plans recorded from then on bracket every GEMM launch (and the patch-embed
# stage) with timestamped plan events, so the kernels are timed inside the real three-stream schedule of a replayed step.
GEMM_MARKS = None
# Dispatch priority of the side-chain streams (negative = higher).  Measured: -1 drops the step rate from 448 to 274 volumes/s
# (priority queues serialise against the captured graph's main queue on this runtime), so the default stays 0.
SIDE_STREAM_PRIORITY = int(os.environ.get("GAVIKO_HIP_SIDE_PRIORITY", "0"))
# Patch embedding as one implicit GEMM (csrc/patch_gemm.hip) instead of the im2col kernel + GEMM: correct and bit-identical, but 74-79 us
# against 54 us for the pair -- opt-in
_PATCH_IMPLICIT = os.environ.get("GAVIKO_HIP_PATCH_IMPLICIT", "0") == "1"
# GPA prompt fix inside the next layer's first LayerNorm (gvk_layernorm_fwd_fix) instead of its own 128-row launch: measured 709-711 vs
# 719-721 volumes/s -- the 128 prompt rows' waves become the tail of a 4132-row kernel; opt-in
_FIX_IN_LN = os.environ.get("GAVIKO_HIP_FIX_IN_LN", "0") == "1"
_SIDE_STREAMS = {}                       # (device index, kind) -> the process-wide side stream of that kind
# MWSA backward chain held behind the layer's attention backward: measured 669 vs 688 volumes/s -- the chain then slows the dgrad GEMMs
# of the next layer by as much as it slowed the attention kernels before (start->fc1d 82 -> 98 us); opt-in only
_LOC_SHIFT = os.environ.get("GAVIKO_HIP_LOC_SHIFT", "0") == "1"
_EPI_NAMES = {0: "store_bf16", 1: "bias_res_f32", 2: "bias_gelu_bf16", 3: "patch_f32", 4: "gelu_bwd_bf16", 5: "store_f32", 6: "bias_res_f32_bf16",
              7: "bias_relu_bf16", 8: "relu_bwd_bf16"}


def evp_highpass_operator(D: int, H: int, W: int, rate: float):
    """The linear operator behind PromptGenerator.fft (evp.py:126-147) as it executes on a [B, C, D, H, W] volume.
    fft2 / ifft2 run over (H, W); fftshift / ifftshift run over EVERY axis; the mask `mask[:, :, w//2-line:w//2+line, h//2-line:h//2+line]`
    (w, h = the last two sizes) is indexed on axes 2 and 3 = (D, H).  Net effect: on the depth slices whose shifted index falls in the
    first range, the H-frequencies whose shifted index falls in the second range are zeroed for every W-frequency; all other slices pass.
    Returns (Hp [H][H] float32 with Hp = I - Re(F^-1 diag(band) F), depth mask int32 [D]): out[b, d] = |Hp . x[b, d]| or |x[b, d]|."""
    import numpy as np
    w_, h_ = H, W                                            # the reference's names for x.shape[-2:]
    line = int((w_ * h_ * rate) ** 0.5 // 2)
    dlo, dhi = max(0, w_ // 2 - line), min(D, w_ // 2 + line)              # slice of axis 2 (depth), clipped like Python slicing
    hlo, hhi = max(0, h_ // 2 - line), min(H, h_ // 2 + line)              # slice of axis 3 (H)
    d_shift = (np.arange(D) + D // 2) % D                    # fftshift: original index d sits at shifted index (d + D//2) % D
    dmask = ((d_shift >= dlo) & (d_shift < dhi)).astype(np.int32)
    k_shift = (np.arange(H) + H // 2) % H
    band = ((k_shift >= hlo) & (k_shift < hhi)).astype(np.float64)
    idx = np.arange(H)
    ph = np.exp(2j * np.pi * np.outer(idx, idx) / H)          # ph[i][k] = e^{2 pi i k i / H}
    A = (ph * band[None, :]) @ ph.conj().T / H                # A[i][j] = 1/H sum_k band[k] e^{2 pi i k (i - j) / H}
    return (np.eye(H) - A.real).astype(np.float32), dmask


class Names:
    """Maps logical backbone tensors to the state_dict names of each reference class."""

    def __init__(self, kind: str):
        self.kind = kind
        self.root = {"vpt": "vision_transformer.", "melo": "lora_vit."}.get(kind, "")

    def attn(self, i):
        if self.kind == "gaviko":
            return f"transformer.attns.{i}"
        if self.kind == "dvpt":
            return f"transformer.layers.{i}.0.attn"
        return f"{self.root}transformer.layers.{i}.0"

    def mlp(self, i):
        if self.kind == "gaviko":
            return f"transformer.mlps.{i}"
        if self.kind == "dvpt":
            return f"transformer.layers.{i}.0.mlp"
        return f"{self.root}transformer.layers.{i}." + ("2" if self.kind == "adaptformer" else "1")

    def conv(self):
        return "conv_proj.proj" if self.kind == "evp" else f"{self.root}conv_proj.0"      # evp.py:292: a PatchEmbed, not a Sequential

    def qkv_weight(self, i):
        return self.attn(i) + (".to_qkv.qkv.weight" if self.kind == "melo" else ".to_qkv.weight")

    def head(self):
        return "mlp_head.head" if self.kind == "gaviko" else f"{self.root}mlp_head"


class Engine:
    def __init__(self, kind: str, cfg: dict, params: Dict[str, torch.nn.Parameter], depth, heads, dim, mlp_dim):
        self.kind, self.cfg, self.p = kind, cfg, params
        self.depth, self.heads, self.C, self.mlp = depth, heads, dim, mlp_dim
        self.names = Names(kind)
        fp, ip = cfg["frame_patch_size"], cfg["image_patch_size"]
        self.patch = (fp, ip, ip)
        self.grid = (cfg["frames"] // fp, cfg["image_size"] // ip, cfg["image_size"] // ip)
        self.N = self.grid[0] * self.grid[1] * self.grid[2]
        self.Kp = fp * ip * ip
        self.K = cfg["num_classes"]
        # Operand precision of the backbone GEMMs / attention: "bf16" = MFMA bf16 operands with fp32 accumulation (the headline
        # path); "fp32" = exact fp32 arithmetic (gemm_f32.hip / attention_f32.hip) for the reference's fp32 configurations
        # (train.py:157 keeps the model in float32 unless config['train']['fp16']; BASELINE cfg4 is pinned at 1e-5).
        prec = str(cfg.get("precision") or os.environ.get("GAVIKO_HIP_PRECISION", "bf16")).lower()
        if prec not in ("bf16", "fp32", "float32"):
            raise L.GavikoHipError(f"precision={prec!r}: expected 'bf16' or 'fp32'")
        self.fp32 = prec != "bf16"
        self.q_scale = 64 ** -0.5 * ops.LOG2E     # what the saved q block carries on the bf16 path (attention kernels' operand form)
        self.adt = torch.float32 if self.fp32 else torch.bfloat16
        if cfg.get("dim_head", 64) != 64:
            raise L.GavikoHipError("the attention kernels are built for dim_head = 64")
        if cfg.get("channels", 1) != 1:
            raise L.GavikoHipError("patch embedding is built for single-channel volumes (channels = 1)")
        self.pool = cfg.get("pool", "cls")
        if kind == "gaviko":
            self.P = cfg["num_prompts"]
            self.Lat = cfg.get("prompt_latent_dim", 20)
            if cfg.get("local_dim", 20) != self.Lat:
                raise L.GavikoHipError("local_dim and prompt_latent_dim must match (one latent width per build)")
            self.share = cfg.get("share_factor", 1)
            self.T = self.P + 1 + self.N
            self.row_off = self.P + 1
            dhw = cfg.get("DHW", (10, 10, 10))
            if dhw is None:
                self.win = tuple(2 * g + 1 for g in self.grid)     # no mask == a window that always covers the grid
            else:
                if tuple(dhw) != self.grid:
                    raise L.GavikoHipError(f"DHW={tuple(dhw)} does not match the patch grid {self.grid}")
                self.win = tuple(cfg.get("local_k", (3, 6, 6)))
        elif kind == "evp":
            self.P, self.T, self.row_off = 0, 1 + self.N, 1
            self.r = dim // int(cfg.get("scale_factor", 32))                              # rank of the prompt latents (evp.py:41-42)
            widths = [w_ for w_ in (4, 8, 16, 20, 24, 32) if w_ >= self.r]
            if not widths or self.r < 1:
                raise L.GavikoHipError(f"EVP: prompt rank dim/scale_factor = {self.r} is outside the rank-L kernels' range (1..32)")
            self.Lp = widths[0]                                                            # latents are zero-padded to this width
            self.freq = float(cfg.get("freq_nums", 0.25))
        elif kind == "dvpt":
            self.P = cfg.get("num_prompts", 50)
            self.Lat = 20                                                                  # share_MLP.latent_dim (dvpt.py:27)
            self.T, self.row_off = self.P + 1 + self.N, self.P + 1
        elif kind == "vpt":
            self.P = cfg.get("num_prompts", 8)
            self.pd = cfg.get("prompt_dim", 64)
            self.deep = bool(cfg.get("deep_prompt", True))
            self.T, self.row_off = 1 + self.P + self.N, 1 + self.P
        else:
            self.P, self.T, self.row_off = 0, 1 + self.N, 1
        if kind == "melo":
            self.r, self.lora_s = int(cfg["r"]), int(cfg["alpha"]) // int(cfg["r"])      # integer alpha // r (melo.py:45-46)
        if kind == "adaptformer":
            self.adim = 64                                                                # Adapter(down_dim=64), adaptformer.py:25
        # tokens entering layer i.  Deep VPT rebuilds the sequence before every layer > 0 as [cls | P prompts | x[:, 1+prompt_dim:]]
        # (vpt.py:147-153: the slice uses deep_prompt_embeddings[i].shape[1] == prompt_dim), so it shrinks by prompt_dim - P per layer.
        self.Ts = [self.T] * depth
        if kind == "vpt" and self.deep:
            for i in range(1, depth):
                self.Ts[i] = self.Ts[i - 1] - self.pd + self.P
            if self.Ts[-1] <= 1 + self.pd:
                raise L.GavikoHipError("deep VPT: the shrinking sequence runs out of tokens for this depth / prompt_dim")
        self._w16: Dict[str, torch.Tensor] = {}
        self._w16_version = None
        self._shadowed = frozenset(self._backbone_weight_names())
        self._fwd_gen = 0                       # counts training-mode forwards: the autograd node checks it owns the saved state
        # static_io = True: forward() returns the workspace's logits buffer itself (valid until the next forward) instead of a copy
        self.static_io = False
        self._have_dgrad = False
        self._graphs = {}
        self._calls = {}
        self._wss = {}
        self._streams = {}
        self._recording = False
        self._keep_inputs = False               # unfrozen backbone weights: the forward keeps the GEMM inputs for their wgrads
        self._eff: Dict[str, torch.Tensor] = {}
        # GPA projections of backbone rows ride along in the backbone's LayerNorm kernels (gvk_layernorm_*_proj)
        self._fuse_proj = (kind == "gaviko" and not self.fp32 and ops.rowproj_supported(self.Lat, dim)
                           and os.environ.get("GAVIKO_HIP_FUSE_PROJ", "1") != "0")
        self._fuse_local = kind == "gaviko" and ops.side_tile_supported(self.Lat, dim)
        self._fuse_bnd = self._fuse_local and os.environ.get("GAVIKO_HIP_FUSE_BOUNDARY", "1") != "0"
        self._fuse_next = self._fuse_local and os.environ.get("GAVIKO_HIP_FUSE_NEXT", "1") != "0"
        self._mwsa_pending = None
        # GPA up-projection as K-concatenation of the MLP's second Linear: 64 spare K columns carry the rank-L product in split-bf16 form,
        # A' = [act | lat_hi | lat_lo | lat_hi | 1 | 1], W' = [W_fc2 | Wup_hi | Wup_hi | Wup_lo | b_hi | b_lo] (fp32-grade: the dropped
        # lo.lo term is 2^-16 relative), so x + ff(x) + proj_up(.) (gaviko.py:187 after vision_transformer.py:34) is ONE GEMM and the
        # main stream loses a full read-modify-write pass over the token stream per layer
        self._fuse_up = (kind == "gaviko" and not self.fp32 and 3 * self.Lat + 2 <= 64 and os.environ.get("GAVIKO_HIP_FUSE_UP", "1") != "0")
        self.ldx = self.mlp + 64 if self._fuse_up else self.mlp      # row stride of the MLP hidden buffers
        self._marks = []
        self.plan_marks = {}                # plan id -> [(name, event id)]
        self._bucket_marks = {}             # (stream kind, layer) -> event of the pass being issued / recorded: gradients of that layer final
        self.plan_bucket_marks = {}         # plan id -> that dict, for replays
        self._want_bucket_marks = False
        self._gemm_marks = []
        self.plan_gemm_marks = {}           # plan id -> [(class, flops, shape, bytes, e0, e1)]
        self._ws = None
        self._step = 0
        self._flat_grad = None
        self._saved = None

    def _gemm(self, a, w, M, out0, alg_k=None, **kw):
        """One NT GEMM launch.  alg_k: the ALGORITHMIC contraction length when the operand carries padding columns (fc2 with the
        K-concatenated GPA up-projection: 3072 + 20 of 3136 columns are products the reference computes, the split-bf16 copies and the
        zero padding are not) -- only the bench instrumentation reads it, for flop_per_launch."""
        if GEMM_MARKS is None or not self._recording:
            return ops.gemm_nt(a, w, M, out0, **kw)
        N, K = w.shape[0], int(kw.get("K") or w.shape[1])
        ka = int(alg_k) if alg_k is not None else K
        key = f"gemm_nt_bf16[{_EPI_NAMES[kw['epilogue']]}] M={M} N={N} K={K}"
        cur = torch.cuda.current_stream()
        e0 = self._ev_record(cur)
        ops.gemm_nt(a, w, M, out0, **kw)
        e1 = self._ev_record(cur)
        self._gemm_marks.append((key, 2.0 * M * N * ka, [M, N, K], None, e0, e1))

    def collect_gemm_marks(self, acc=None):
        """After a sync: add the event-pair durations of the last replay of every instrumented plan to `acc`
        ({class: {"ms": [...], "flops", "shape", "bytes", "overhead_ms"}}).  Durations are RAW event-pair times (they agree with the
        rocprofv3 kernel durations of the same launches, profiles/r02_bench_kernel_stats.csv); the empty event pair recorded at the head
        of each plan is reported beside them as `overhead_ms`, never subtracted."""
        import ctypes
        acc = {} if acc is None else acc
        lib, ms = L.load(), ctypes.c_float()
        for pid, marks in self.plan_gemm_marks.items():
            overhead = 0.0
            for key, flops, shape, nbytes, e0, e1 in marks:
                L.check(lib.gvk_plan_event_elapsed(pid, e0, e1, ctypes.byref(ms)), "gvk_plan_event_elapsed")
                if key == "__empty__":
                    overhead = ms.value
                    continue
                rec = acc.setdefault(key, {"ms": [], "flops": flops, "shape": shape, "bytes": nbytes, "overhead_ms": overhead})
                rec["ms"].append(ms.value)
        return acc

    # ------------------------------------------------------------------ weights
    def _d(self, name) -> torch.Tensor:
        eff = self._eff.get(name)                 # SSF: LayerNorm affines and biases are read in their effective (folded) form
        return self.p[name].detach() if eff is None else eff

    def _backbone_weight_names(self) -> List[str]:
        n = [self.names.conv() + ".weight"]
        for i in range(self.depth):
            n += [self.names.qkv_weight(i), self.names.attn(i) + ".to_out.0.weight", self.names.mlp(i) + ".net.1.weight",
                  self.names.mlp(i) + ".net.4.weight"]
        return n

    def refresh_weights(self, need_dgrad: bool) -> None:
        """(Re)build the bf16 shadows when a source weight changed (load_state_dict, optimizer step on an unfrozen tensor).
        Shadows are rewritten IN PLACE so that captured HIP graphs keep pointing at valid operands."""
        if self.kind == "ssf":
            return                                  # every operand is re-folded from (W, scale) inside the recorded step (_ssf_fold)
        names = self._backbone_weight_names()
        version = tuple(self.p[n]._version for n in names) + tuple(self.p[n].data_ptr() for n in names)
        stale = version != self._w16_version
        w = self._w16
        if stale:
            conv = self._d(names[0]).reshape(self.C, self.Kp).contiguous()
            w["conv"] = ops.to_operand(conv, None if self.fp32 else w.get("conv"), self.adt)
        if not stale and (not need_dgrad or self._have_dgrad):
            return
        for i in range(self.depth):
            for tag, nm in (("qkv", self.names.qkv_weight(i)), ("out", self.names.attn(i) + ".to_out.0.weight"),
                            ("fc1", self.names.mlp(i) + ".net.1.weight"), ("fc2", self.names.mlp(i) + ".net.4.weight")):
                src = self._d(nm).contiguous()
                if stale and tag == "fc2" and self._fuse_up:
                    buf = w.get(f"fc2{i}")
                    if buf is None:
                        buf = w[f"fc2{i}"] = torch.zeros((self.C, self.ldx), dtype=self.adt, device=src.device)
                    buf[:, :self.mlp].copy_(src)                 # columns mlp.. are packed from the trainable proj_up inside every step
                elif stale:
                    w[f"{tag}{i}"] = ops.to_operand(src, None if self.fp32 else w.get(f"{tag}{i}"), self.adt)   # fp32: the parameter itself
                if need_dgrad and (stale or not self._have_dgrad):
                    w[f"{tag}{i}_t"] = ops.transpose_operand(src, w.get(f"{tag}{i}_t"), self.adt)
        self._have_dgrad = self._have_dgrad and not stale or need_dgrad
        self._w16_version = version

    # ------------------------------------------------------------------ workspace
    def _seed_base(self, train: bool) -> int:
        """Initial value of the device-side dropout epoch word.  Mixes torch's base seed (torch.manual_seed = the user's knob), the
        data-parallel rank and the mode, so that ranks -- and the train / eval workspaces of one rank -- draw independent masks, as
        the reference's per-process torch RNG streams do (train.py has no seeding; every process draws its own)."""
        rank = 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank = dist.get_rank()
        except Exception:
            rank = 0
        x = (torch.initial_seed() & 0xFFFFFFFFFFFF) * 0x9E3779B97F4A7C15 + (rank + 1) * 0xBF58476D1CE4E5B9 + (0x94D049BB133111EB if train else 0)
        x &= (1 << 64) - 1
        x ^= x >> 31
        return 0x5EED0000 ^ (x & 0x3FFFFFFFFFFF0000)          # stays positive in int64; the low 16 bits are left to the per-site offsets

    def invalidate_weights(self, names=None) -> None:
        """Tell the engine that parameters were written behind torch's back (a raw-pointer optimizer step: `p._version` does not move).
        With `names`, only if one of them is a backbone weight the engine keeps operand shadows of."""
        if names is None or any(n in self._shadowed for n in names):
            self._w16_version = None

    def workspace(self, B: int, device, train: bool):
        key = (B, train, str(device))
        if key in self._wss:
            self._ws = self._wss[key]
            return self._ws
        C, T, N, M = self.C, self.T, self.N, B * self.T
        z = lambda r, c, dt: ops.act_zeros(r, c, dt, device)
        f32, bf16 = torch.float32, self.adt                # "bf16" below = the GEMM-operand dtype (fp32 on the fp32 path)
        nsave = self.depth if train else 1
        ws = {"key": key, "B": B, "M": M}
        ws["img"] = torch.zeros((B, 1) + tuple(g * p for g, p in zip(self.grid, self.patch)), device=device)
        ws["logits"] = torch.zeros((B, self.K), device=device)
        ws["dlogits"] = torch.zeros((B, self.K), device=device)
        ws["seed"] = torch.full((1,), self._seed_base(train), dtype=torch.int64, device=device)
        ws["cols"] = z(B * N, self.Kp, bf16)
        ws["G"] = [z(M, C, f32) for _ in range(self.depth + 1)] if train else [z(M, C, f32), z(M, C, f32)]
        ws["G1"] = [z(M, C, f32) for _ in range(nsave)]
        ws["xn"] = z(M, C, bf16)
        ws["qkv"] = [z(M, 3 * C, bf16) for _ in range(nsave)]
        ws["ctx"] = [z(M, C, bf16) for _ in range(nsave)]
        ws["lse"] = [torch.zeros((B, self.heads, T), device=device) for _ in range(nsave)]
        ws["pre"] = [z(M, self.ldx, bf16) for _ in range(nsave)] if train else [None]
        ws["act"] = z(M, self.ldx, bf16)
        if self._fuse_up:
            ws["act"][:, self.mlp + 3 * self.Lat: self.mlp + 3 * self.Lat + 2] = 1.0     # the two bias columns (b_hi, b_lo); the rest of the slot stays 0
        ws["stat"] = [[torch.zeros(M, device=device) for _ in range(4)] for _ in range(nsave)]   # mean1, rstd1, mean2, rstd2
        ws["pooled"] = torch.zeros((B, C), device=device)
        if self.kind == "gaviko":
            Lt, P, BN = self.Lat, self.P, B * N
            ws["Lc"] = [torch.zeros((BN, C), device=device) for _ in range((self.depth + 1) if train else 2)]
            mk = lambda *s: torch.zeros(s, device=device)
            ws["mw"] = [dict(mean=mk(BN), rstd=mk(BN), lat=mk(BN, Lt), qkv=mk(BN, 3 * Lt), ctx=mk(BN, Lt), lse=mk(BN)) for _ in range(nsave)]
            ws["gp"] = [dict(zx=mk(M, Lt), xl=mk(M, Lt), zl=mk(BN, Lt), ll=mk(BN, Lt), imp=mk(B, P), gw=mk(B), enh=mk(B, P, Lt),
                             prm=mk(B, P, Lt), qg=mk(B, P, Lt), ql=mk(B, P, Lt), cg=mk(B, P, Lt), cl=mk(B, P, Lt), lse_g=mk(B, P),
                             lse_l=mk(B, P)) for _ in range(nsave)]
        if self.kind == "evp":
            mk = lambda *s_: torch.zeros(s_, device=device)
            BN, Lp = B * N, self.Lp
            ws["xc"] = z(BN, C, f32)
            ws["hp"] = torch.zeros_like(ws["img"])
            ws["hcols"] = z(BN, self.Kp, f32)
            ws["hc"] = z(BN, 64, f32)
            ws["ev"] = dict(e=mk(BN, Lp), s=mk(BN, Lp), pre=[mk(BN, Lp) for _ in range(nsave)], u=[mk(BN, Lp) for _ in range(nsave)],
                            tmp=z(BN, C, f32))
            if train:
                ws["evb"] = dict(du=mk(BN, Lp), dpre=mk(BN, Lp), ds_tmp=mk(BN, Lp), ds=mk(BN, Lp))
                ws["scratch"] = mk(max(ops.outer_scratch_elems(Lp, self.Kp), 128 * C))
                ws["rscratch"] = mk(32 * (Lp * Lp + Lp + 64))
        if self.kind == "dvpt":
            mk = lambda *s_: torch.zeros(s_, device=device)
            ws["dv"] = [dict(z=mk(M, self.Lat), enh=mk(B, self.P, self.Lat), lse=mk(B, self.P)) for _ in range(nsave)]
        if self.kind == "adaptformer":
            ws["xa"] = z(M, C, bf16)
            ws["ad"] = [dict(mean=torch.zeros(M, device=device), rstd=torch.zeros(M, device=device), h16=z(M, self.adim, bf16))
                        for _ in range(nsave)]
            if train:
                ws["dh16"] = z(M, self.adim, bf16)
                ws["h32"] = torch.zeros((M, self.adim), device=device)
                ws["dh32"] = torch.zeros((M, self.adim), device=device)
        if self.kind == "melo":
            ws["merge32"] = torch.zeros((3 * C, C), device=device)
            if train:
                mk = lambda *s_: torch.zeros(s_, device=device)
                ws["dq32"], ws["dv32"] = mk(M, C), mk(M, C)
                ws["lu"] = dict(uq=mk(M, self.r), uv=mk(M, self.r), duq=mk(M, self.r), duv=mk(M, self.r))
        if self.kind == "vpt":
            R = (self.depth if self.deep else 1) * self.P
            ws["vproj"] = torch.zeros((R, C), device=device)
            ws["Go"] = z(M, C, f32)                          # layer output before the deep-VPT re-pack
            if train:
                ws["dvproj"] = torch.zeros((R, C), device=device)
                ws["dGv"] = z(M, C, f32)
        if train:
            ws["dG"] = [z(M, C, f32), z(M, C, f32)]          # ping-pong gradient of the global stream
            ws["dG16"] = z(M, C, bf16)
            ws["dpre"] = z(M, self.mlp, bf16)
            ws["dx32"] = z(M, C, f32)
            ws["dctx"] = z(M, C, bf16)
            ws["dqkv"] = z(M, 3 * C, bf16)
            ws["delta"] = torch.zeros((B, self.heads, T), device=device)
            if self.kind == "gaviko":
                Lt, P, BN = self.Lat, self.P, B * N
                mk = lambda *s: torch.zeros(s, device=device)
                ng = ops.gpa_gate_param_count(Lt, P)
                ws["dL"] = [mk(BN, C), mk(BN, C)]
                ws["dGb"] = ops.act_zeros(M, C, torch.float32, device)       # second layer-boundary gradient buffer (parity ping-pong)
                ws["bw"] = dict(dcomb=mk(M, Lt), dimp=mk(B, P), dgw_part=mk(B, P), dqg=mk(B, P, Lt), dql=mk(B, P, Lt), dcg=mk(B, P, Lt),
                                dcl=mk(B, P, Lt), delta_g=mk(B, P), delta_l=mk(B, P), dprm=mk(B, P, Lt), dcls=mk(B, Lt),
                                gate_partials=mk(B, ng), dzx=mk(M, Lt), dzl=[mk(BN, Lt), mk(BN, Lt)],
                                dctx=mk(BN, Lt), dqkv=mk(BN, 3 * Lt), wdelta=mk(BN), dlat=mk(BN, Lt), Q=mk(Lt, C), S=mk(Lt))
                ws["scratch"] = mk(max(ops.outer_scratch_elems(Lt, C), 128 * C, 64 * 3 * Lt * Lt, 64 * ng))
                ws["rscratch"] = mk(32 * (ng + 2 * Lt * Lt + 3 * Lt + 3 * Lt * Lt + 64))
                ws["scratch_l"] = mk(max(ops.outer_scratch_elems(Lt, C), 128 * C))      # the MWSA chain runs on its own stream
                ws["rscratch_l"] = mk(32 * (3 * Lt * Lt + Lt + 64))
            elif self.kind == "dvpt":
                Lt, P = self.Lat, self.P
                mk = lambda *s_: torch.zeros(s_, device=device)
                ws["dvb"] = dict(dcomb=mk(M, Lt), dz=mk(M, Lt), delta=mk(B, P))
                ws["scratch"] = mk(max(ops.outer_scratch_elems(Lt, C), 128 * C))
                ws["rscratch"] = mk(32 * (Lt + 64))
            elif self.kind == "evp":
                pass                                        # scratch / rscratch sized with the EVP buffers above
            else:
                if self.kind == "ssf":
                    ws["ssf_scratch"] = torch.zeros(64 * 2 * max(self.mlp, 3 * C), device=device)
                    ws["ssf_tmp"] = torch.zeros(2 * C, device=device)
                    ws["ssf_stat"] = [torch.zeros(M, device=device), torch.zeros(M, device=device)]
                lat = {"adaptformer": 64, "melo": getattr(self, "r", 4)}.get(self.kind, 1)
                ws["scratch"] = torch.zeros(max(128 * C, ops.outer_scratch_elems(lat, C)), device=device)
        self._ws = self._wss[key] = ws       # one workspace (and one set of captured graphs) per (batch, mode)
        return ws

    # ------------------------------------------------------------------ two-stream fork / join
    # GAViKO's local branch (MWSA) and the latent-space GPA core are independent of the backbone's attention / MLP GEMMs within
    # a layer; they run on a side stream and meet the main stream only where the dataflow does (gaviko.py:301-304).  Inside
    # a HIP-graph capture these waits become graph edges, so the replayed graph has two parallel branches.
    def _stream(self, name):
        st = self._streams.get(name)
        if st is None:
            # One side stream of each kind per DEVICE, shared by every engine of the process (train + eval models, a test suite's many
            # models): engines run their steps one after another, and every stream of their own would eventually alias hardware queues
            # (tools/bench_reducer.py: the 5th model of a process ran 14.8 ms steps instead of 5.9).
            key = (torch.cuda.current_device(), name)
            st = _SIDE_STREAMS.get(key)
            if st is not None:
                self._streams[name] = st
                return st
            # (confining the side streams to a CU subset with hipExtStreamCreateWithCUMask was measured: 676 -> 170-260 volumes/s for
            #  every mask shape tried -- masked queues are far slower to dispatch on this runtime)
            prio = int(os.environ.get(f"GAVIKO_HIP_{name.upper()}_PRIORITY", SIDE_STREAM_PRIORITY))     # per-stream A/B switch
            st = self._streams[name] = _SIDE_STREAMS[key] = torch.cuda.Stream(priority=prio)
            pad = int(os.environ.get(f"GAVIKO_HIP_LDS_PAD_{name.upper()}", os.environ.get("GAVIKO_HIP_LDS_PAD", "0")))
            if pad:
                L.check(L.load().gvk_stream_set_lds_pad(st.cuda_stream, pad), "gvk_stream_set_lds_pad")
            if "sidenop" in _ABLATE or f"{name}nop" in _ABLATE:
                L.load().gvk_plan_nop_stream(st.cuda_stream)
        return st

    def _ev_record(self, stream):
        """Record an event on `stream`; while a launch plan is being recorded the event belongs to the plan."""
        if "noevents" in _ABLATE:
            return None
        if self._recording:
            rc = L.load().gvk_plan_event_record(stream.cuda_stream)
            if rc < 0:
                L.check(rc, "gvk_plan_event_record")
            return rc
        ev = torch.cuda.Event()
        ev.record(stream)
        return ev

    def _bucket_mark(self, kind, layer):
        """Data parallelism (distributed.GradReducer, mode 'events'): an event on the CURRENT stream after the last kernel that writes
        a gradient of (`kind`, `layer`); the collective stream waits for it.  Recorded with the system-scope fence: peers read the data."""
        if not self._want_bucket_marks:
            return
        cur = torch.cuda.current_stream()
        if self._recording:
            rc = L.load().gvk_plan_event_record_fenced(cur.cuda_stream)
            if rc < 0:
                L.check(rc, "gvk_plan_event_record_fenced")
            self._bucket_marks[(kind, layer)] = rc
        else:
            ev = torch.cuda.Event()
            ev.record(cur)
            self._bucket_marks[(kind, layer)] = ev

    def _mark(self, name):
        """Diagnostics (GAVIKO_HIP_PLAN_TIMING=1): a timestamped plan event on the current stream, read by tools/plan_marks.py."""
        if self._recording and PLAN_TIMING:
            self._marks.append((name, self._ev_record(torch.cuda.current_stream())))

    def _ev_wait(self, stream, ev):
        if "noevents" in _ABLATE or ("nowait" in _ABLATE and stream.cuda_stream == torch.cuda.current_stream().cuda_stream):
            return
        if self._recording:
            L.check(L.load().gvk_plan_event_wait(stream.cuda_stream, ev), "gvk_plan_event_wait")
        else:
            stream.wait_event(ev)

    def _wait(self, waiter, on):
        """stream `waiter` waits for everything enqueued so far on stream `on` (None = the current/main stream)."""
        cur = torch.cuda.current_stream()
        src = cur if on is None else self._stream(on)
        dst = cur if waiter is None else self._stream(waiter)
        self._ev_wait(dst, self._ev_record(src))

    # ------------------------------------------------------------------ graphs
    def _run(self, tag, key, fn):
        """Run `fn` eagerly the first GRAPH_WARMUP times, then capture it into a HIP graph and replay that.
        Everything `fn` launches reads/writes workspace buffers only, so a replay is exactly one more step."""
        k = (tag,) + key + (torch.cuda.current_stream().cuda_stream,)
        g = self._graphs.get(k)
        self._last_run = ("eager", None)
        if g is not None:
            if isinstance(g, int):
                L.check(L.load().gvk_plan_replay(g), "gvk_plan_replay")
                self._last_run = ("replayed", g)
            else:
                g.replay()
                self._last_run = ("graph", None)
            return
        n = self._calls.get(k, 0)
        self._calls[k] = n + 1
        if not USE_GRAPHS or n < GRAPH_WARMUP:
            fn()
            return
        if STEP_MODE == "plan":
            # this pass both executes and records; every launch inside fn goes through the library (no torch kernels)
            lib = L.load()
            L.check(lib.gvk_plan_begin(), "gvk_plan_begin")
            self._recording = True
            self._marks, self._gemm_marks = [], []
            self._bucket_marks = {}
            try:
                if GEMM_MARKS is not None:               # calibration: an empty event pair on the launch stream
                    cur = torch.cuda.current_stream()
                    self._gemm_marks.append(("__empty__", 0.0, None, None, self._ev_record(cur), self._ev_record(cur)))
                fn()
            except BaseException:
                lib.gvk_plan_abort()
                raise
            finally:
                self._recording = False
            pid = lib.gvk_plan_end()
            if pid < 0:
                L.check(pid, "gvk_plan_end")
            self._graphs[k] = pid
            self.plan_marks[pid] = (tag, self._marks)
            self.plan_bucket_marks[pid] = dict(self._bucket_marks)
            self._last_run = ("recorded", pid)
            if self._gemm_marks:
                self.plan_gemm_marks[pid] = self._gemm_marks
            return
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        self._graphs[k] = g
        g.replay()

    # ------------------------------------------------------------------ forward
    def forward(self, img: torch.Tensor, train: bool, drop: Optional[dict] = None) -> torch.Tensor:
        L.require_device()
        if not img.is_cuda:
            raise L.GavikoHipError("input volume must be on the HIP device: gaviko_amd has no CPU path")
        if img.dim() != 5 or img.shape[1] != 1 or tuple(img.shape[2:]) != tuple(g * p for g, p in zip(self.grid, self.patch)):
            raise L.GavikoHipError(f"expected img [B,1,{self.grid[0] * self.patch[0]},{self.grid[1] * self.patch[1]},"
                                   f"{self.grid[2] * self.patch[2]}], got {tuple(img.shape)}")
        B = img.shape[0]
        drop = drop or {}
        sv = {"B": B, "train": train, "attn_drop": float(drop.get("attn_drop", 0.0)), "proj_drop": float(drop.get("proj_drop", 0.0))}
        # nn.Dropout of the backbone itself (vision_transformer.py:33-34,52-54,157; vpt.py:129,148): live for the classes without a
        # train() override (linear / bitfit / fft, melo) and for VPT's prompt_dropout.  bf16 path only.
        sv["bdrop"], sv["edrop"], sv["pdrop"] = (float(drop.get(k, 0.0)) for k in ("dropout", "emb_dropout", "prompt_dropout"))
        if (sv["bdrop"] > 0 or sv["edrop"] > 0 or sv["pdrop"] > 0) and self.kind not in ("vit", "melo", "vpt"):
            raise L.GavikoHipError(f"backbone dropout > 0 in training mode is built for the vit / melo / vpt classes, not kind={self.kind!r}")
        self.refresh_weights(need_dgrad=train)
        ws = self.workspace(B, img.device, train)
        if img.data_ptr() != ws["img"].data_ptr():           # a caller that fills input_buffer() itself skips the copy-in launch
            ws["img"].copy_(img.detach())                   # static input buffer (the only per-step host-visible copy-in)
        # unfrozen backbone tensors (`fft` / `bitfit`, train.py:123-137): which ones train, and whether GEMM inputs must be kept
        bb = frozenset(n for n in self.trainable_names() if not n.startswith(self.names.head())) if (train and self.kind == "vit") else frozenset()
        sv["bb"] = bb
        sv["wgrad"] = any(self.p[n].dim() >= 2 and n.endswith("weight") for n in bb)
        if bb:
            self._bb_buffers(ws, B, img.device, sv["wgrad"])
        key = (B, train, sv["attn_drop"], sv["proj_drop"], len(bb), sv["wgrad"], sv["bdrop"], sv["edrop"], sv["pdrop"])
        self._keep_inputs = bool(sv["wgrad"])
        self._run("fwd", key, lambda: self._forward_impl(ws, sv))
        self._saved = sv if train else None
        self._saved_key = key
        if train:
            self._fwd_gen += 1
        return ws["logits"] if self.static_io else ws["logits"].clone()

    def input_buffer(self, B: int, device, train: bool = True) -> torch.Tensor:
        """The static [B,1,D,H,W] input slot of the (B, train) workspace: a data pipeline that writes its batch here (and passes this very
        tensor to the model) saves the per-step device copy."""
        return self.workspace(B, device, train)["img"]

    def _forward_impl(self, ws, sv):
        B, C, T, N, train = sv["B"], self.C, self.T, self.N, sv["train"]
        M = B * T
        nm, w, d = self.names, self._w16, self._d
        ops.seed_advance(ws["seed"], 7919)                  # device-side dropout epoch (replay safe)
        if self.kind == "ssf":
            self._ssf_fold(train)
        # ---- embedding: patch GEMM (+bias +pos, scattered to rows row_off..) and the broadcast rows
        marking = GEMM_MARKS is not None and self._recording  # bench.py: time the whole patch-embed stage (im2col + GEMM + scatter)
        cur = torch.cuda.current_stream()
        pe0 = self._ev_record(cur) if marking else None
        pos = d(nm.root + "pos_embedding")[0]
        G0 = ws["G"][0]
        # implicit GEMM straight from the fp32 volume (csrc/patch_gemm.hip) wherever nothing else needs the im2col matrix: not EVP (reads
        # the raw embedding), not a trainable patch conv (its weight gradient contracts over the im2col rows), not the fp32 path
        conv_trains = self.p[nm.conv() + ".weight"].requires_grad
        implicit = (_PATCH_IMPLICIT and not self.fp32 and self.kind != "evp" and not conv_trains and C % 128 == 0 and self.patch[2] == 16
                    and (self.patch[1] * self.patch[2]) % 64 == 0)
        if implicit:
            ops.patch_embed(ws["img"], w["conv"], d(nm.conv() + ".bias"), pos[1:], G0, ws["Lc"][0] if self.kind == "gaviko" else None,
                            self.patch, C, T, self.row_off)
        else:
            ops.patchify(ws["img"], ws["cols"], self.patch)
        if implicit:
            pass                                              # tokens are already in place
        elif self.kind == "evp":
            # the raw patch embedding is needed on its own (embedding_generator reads it, evp.py:347-348): conv -> xc, tokens = xc + pos
            ops.gemm_nt(ws["cols"], w["conv"], B * N, ws["xc"], epilogue=ops.EPI_STORE_F32, bias=d(nm.conv() + ".bias"))
            ops.rows_patch(G0, ws["xc"], pos[1:], B, T, N, C, 1, False)
            self._evp_latents(ws, B)
        else:
            ops.gemm_nt(ws["cols"], w["conv"], B * N, G0, epilogue=ops.EPI_PATCH_F32, out1=ws["Lc"][0] if self.kind == "gaviko" else None,
                        bias=d(nm.conv() + ".bias"), pos=pos[1:], rows_in=N, rows_out=T, row_off=self.row_off)
        if marking:
            nout = 2 if self.kind == "gaviko" else 1
            self._gemm_marks.append(("__patch_embed__", 2.0 * B * N * self.Kp * C, [B * N, C, self.Kp],
                                     B * (self.Kp * N * 4 + nout * N * C * 4), pe0, self._ev_record(cur)))
        cls = d(nm.root + "cls_token")[0]
        if self.kind in ("gaviko", "dvpt"):                   # [P prompts | cls | patches] (gaviko.py:536-548, dvpt.py:196-199)
            ops.rows_broadcast(G0, d("prompt_embeddings")[0], d("prompt_positional_embedding")[0], B, T, 0, self.P, C)
            ops.rows_broadcast(G0, cls, pos[0:1], B, T, self.P, 1, C)
        else:
            ops.rows_broadcast(G0, cls, pos[0:1], B, T, 0, 1, C)
        if sv["edrop"] > 0:                                   # x = dropout(x + pos) over [cls | patches] (vision_transformer.py:157)
            ops.dropout_rows(G0, sv["edrop"], SEED_EMB, ws["seed"], out32=G0, M=B * T, N=C)
        if self.kind == "vpt":                                # prompt_proj on every layer's prompts at once (vpt.py:56,127-153)
            emb = d("deep_prompt_embeddings" if self.deep else "prompt_embeddings").reshape(-1, self.pd)
            ops.small_linear_fwd(emb, d("prompt_proj.weight"), d("prompt_proj.bias"), ws["vproj"], emb.shape[0], self.pd, C)
            ops.rows_broadcast(G0, ws["vproj"][: self.P], None, B, T, 1, self.P, C)
            self._prompt_dropout(ws, sv, 0, G0, self.Ts[0])
        if self.kind == "melo":
            self._melo_merge(ws, train)
        if self.kind == "adaptformer":
            self._adapter_shadows(train)
        # ---- layers.  main stream: attention block -> MLP block;  side stream: MWSA -> GPA latents / gates / cross-attention
        gaviko = self.kind == "gaviko"
        if gaviko:
            loc, gpa = self._stream("loc"), self._stream("gpa")
            self._wait("loc", None)                                  # Lc[0] written by the patch GEMM
            self._wait("gpa", None)
            fuse_up = self._fuse_up and sv["bdrop"] <= 0             # (dropout behind fc2 must not touch the up-projection)
            if fuse_up and "noside" not in _ABLATE:
                with torch.cuda.stream(gpa):                         # the trainable half of W': off the main stream, once per step
                    for i in range(self.depth):
                        pre, _ = self._gpa_names(i)
                        ops.pack_split_bf16(d(pre + ".proj_up.weight"), self._w16[f"fc2{i}"], self.mlp, C, b=d(pre + ".proj_up.bias"), weight_side=True)
        pending_fix = None
        for i in range(self.depth):
            si = i if train else 0
            gi, go = (i, i + 1) if train else (i & 1, (i + 1) & 1)
            Mi = B * self.Ts[i]
            repack = self.kind == "vpt" and self.deep
            gout = ws["Go"] if repack else ws["G"][go]
            if gaviko:
                if not train and i > 0:
                    self._wait("loc", "gpa")                         # eval ping-pongs Lc: the GPA of layer i-1 must be done with it
                with torch.cuda.stream(loc):
                    # with the 16-row-tile kernels the MWSA up-projection also emits GPA's proj_down of the rows it writes
                    fuse_local = self._fuse_proj and self._fuse_local
                    self._mwsa_fwd(ws, sv, i, si, ws["Lc"][gi], ws["Lc"][go], gpa_local=fuse_local)
                    if self._fuse_proj and not fuse_local:
                        self._gpa_down_local(ws, i, si, ws["Lc"][go], B)
            self._mark(f"f{i}:start")
            if self.kind == "evp":
                self._evp_add_prompt(ws, i, si, ws["G"][gi], B)               # x[:, 1:] += prompt_i (evp.py:235-238)
            if gaviko and pending_fix is not None:
                self._wait(None, "gpa")                              # the previous layer's enh
            self._attn_block_fwd(ws, i, si, ws["G"][gi], ws["G1"][si], Mi, sv["bdrop"], fix=pending_fix if gaviko and _on("noside") else None)
            pending_fix = None
            self._mark(f"f{i}:attn")
            fused = gaviko and self._fuse_proj
            if gaviko and not fused:
                self._wait("gpa", None)                              # G1 ready
                self._wait("gpa", "loc")                             # L' ready
                with torch.cuda.stream(gpa):
                    self._gpa_fwd_latents(ws, i, si, ws["G1"][si], ws["Lc"][go], M, B, True)
            if self.kind == "adaptformer":
                self._adapter_fwd_down(ws, i, si, ws["G1"][si], Mi)
            if self.kind == "dvpt":
                self._dvpt_fwd_latents(ws, i, si, ws["G1"][si], Mi, B)
            self._mlp_ln_fwd(ws, i, si, ws["G1"][si], Mi, fused)     # fused: also zx / xl = GPA proj_down(G1), same pass
            if fused:
                self._wait("gpa", None)                              # xl ready
                self._wait("gpa", "loc")                             # ll ready (the MWSA chain projects its own L')
                with torch.cuda.stream(gpa):
                    self._gpa_fwd_latents(ws, i, si, ws["G1"][si], ws["Lc"][go], M, B, False)
            up_in_fc2 = gaviko and fused and fuse_up
            # fc2 carries proj_up of the PLAIN latents of every row (ready right behind the LayerNorm); the GPA has the two GEMMs' time
            # to finish, and only the P prompt rows it replaces are fixed up afterwards
            self._mlp_block_fwd(ws, i, si, ws["G1"][si], gout, Mi, train, sv["bdrop"],
                                up_in_fc2=up_in_fc2)
            if self.kind == "adaptformer":
                self._adapter_fwd_up(ws, i, si, gout, Mi)
            if self.kind == "dvpt":
                self._dvpt_fwd_up(ws, i, si, gout, Mi)
            self._mark(f"f{i}:mlp")
            if gaviko and up_in_fc2:
                # the P prompt rows still lack (enh - xl) . Wup^T: the next layer's first LayerNorm applies it on the way in (one launch less
                # on this stream); the last layer has no successor and launches the 128-row fix itself
                pre, _ = self._gpa_names(i)
                g = ws["gp"][si]
                pending_fix = dict(enh=g["enh"], lat=g["xl"], wup=d(pre + ".proj_up.weight"))
                if i + 1 == self.depth or not _FIX_IN_LN:
                    self._wait(None, "gpa")                          # enh ready
                    if _on("noside"):
                        ops.prompt_up_fix(pending_fix["enh"], pending_fix["lat"], pending_fix["wup"], ws["G"][go], B, self.T, self.P, C, self.Lat)
                    pending_fix = None
            elif gaviko:
                self._wait(None, "gpa")                              # enh ready
                self._gpa_fwd_up(ws, i, si, ws["G"][go], M)
            self._mark(f"f{i}:end")
            if repack and i + 1 < self.depth:
                ops.vpt_repack_fwd(gout, ws["vproj"][(i + 1) * self.P: (i + 2) * self.P], ws["G"][go], B, self.Ts[i], self.Ts[i + 1],
                                   self.P, self.pd, C)
                self._prompt_dropout(ws, sv, i + 1, ws["G"][go], self.Ts[i + 1])
        if gaviko:
            self._wait(None, "loc")                                  # join the local chain (capture needs every fork joined)
        gfin = self._final_stream(ws, train)
        r0, R = self._pool_rows()
        ops.head_fwd(g=gfin, ln_gamma=d(nm.root + "transformer.norm.weight"), ln_beta=d(nm.root + "transformer.norm.bias"),
                     wh=d(nm.head() + ".weight"), bh=d(nm.head() + ".bias"), logits=ws["logits"], pooled=ws["pooled"],
                     B=B, T=self.Ts[-1], C=C, K=self.K, r0=r0, R=R)

    def _final_stream(self, ws, train):
        if self.kind == "vpt" and self.deep:
            return ws["Go"]
        return ws["G"][self.depth] if train else ws["G"][self.depth & 1]

    def _pool_rows(self):
        if self.kind == "dvpt":                       # dvpt.py:80-83,205: 'cls' reads row 0 -- the FIRST PROMPT; 'mean' = prompts + cls
            return (0, self.P + 1) if self.pool == "mean" else (0, 1)
        if self.kind == "gaviko":
            return 0, self.P + 1                      # gaviko.py:316 prompts + CLS
        return (0, self.Ts[-1]) if self.pool == "mean" else (0, 1)

    def _masked_grad(self, ws, dy, p, seed, need32, M):
        """ws['dG16'] (the dgrad GEMM operand) = dy * mask; also returns the fp32 masked gradient when bias / weight gradients need it."""
        if self.fp32:                                        # the operand IS fp32 on this path
            ops.dropout_rows(dy, p, seed, ws["seed"], out32=ws["dG16"], M=M, N=self.C)
            return ws["dG16"]
        out32 = ws["dyd"] if need32 else None
        ops.dropout_rows(dy, p, seed, ws["seed"], out32=out32, out16=ws["dG16"], M=M, N=self.C)
        return out32

    def _prompt_dropout(self, ws, sv, i, g, T):
        """prompt_dropout on the projected prompts of layer i, rows 1..P of every sample (vpt.py:129,148,152: applied after .expand(B),
        so every sample draws its own mask)."""
        if sv["pdrop"] > 0:
            ops.dropout_rows(g, sv["pdrop"], SEED_PROMPT + i, ws["seed"], out32=g, M=sv["B"] * self.P, N=self.C, rows_in=self.P, rows_out=T, row_off=1)

    def _attn_block_fwd(self, ws, i, si, gin, g1, M, pdrop=0.0, fix=None):
        nm, w, d, C = self.names, self._w16, self._d, self.C
        a = nm.attn(i)
        st = ws["stat"][si]
        if fix is not None:                                  # + the previous layer's GPA prompt fix, applied to gin in place
            ops.layernorm_fwd_fix(gin, d(a + ".norm.weight"), d(a + ".norm.bias"), M, C, y16=ws["xn"], mean=st[0], rstd=st[1], T=self.T, P=self.P,
                                  L_=self.Lat, **fix)
        else:
            ops.layernorm_fwd(gin, d(a + ".norm.weight"), d(a + ".norm.bias"), M, C, y16=ws["xn"], mean=st[0], rstd=st[1])
        if self._keep_inputs:
            ops.copy_(ws["sav"]["xn1"][si], ws["xn"])
        # bf16 path: the q block leaves the projection as q * scale * log2(e) (one rounding, in the GEMM epilogue) -- the form the flash
        # kernels take, forward and backward alike; the fp32 kernels take the raw block
        qs = {} if self.fp32 else dict(scale_cols=self.heads * 64, col_scale=self.q_scale)
        self._gemm(ws["xn"], w[f"qkv{i}"], M, ws["qkv"][si], epilogue=ops.EPI_STORE_BF16, bias=self._eff.get(a + ".to_qkv.bias"), **qs)
        ops.attention_fwd(ws["qkv"][si], ws["ctx"][si], ws["lse"][si], ws["B"], self.Ts[i], self.heads, 64 ** -0.5,
                          drop_p=pdrop, seed=SEED_LAYER + 8 * i, seed_ptr=ws["seed"], q_prescaled=True)
        self._gemm(ws["ctx"][si], w[f"out{i}"], M, g1, epilogue=ops.EPI_BIAS_RES_F32, bias=d(a + ".to_out.0.bias"), res=gin,
                   drop_p=pdrop, seed=SEED_LAYER + 8 * i + 1, seed_ptr=ws["seed"])

    def _mlp_ln_fwd(self, ws, i, si, g1, M, fused):
        nm, d, C = self.names, self._d, self.C
        m = nm.mlp(i)
        st = ws["stat"][si]
        if fused:
            pre, _ = self._gpa_names(i)
            g = ws["gp"][si]
            split = dict(y_split=ws["act"], col_split=self.mlp) if self._fuse_up else {}      # the plain latents ride fc2 (self._fuse_up)
            ops.layernorm_fwd_proj(g1, d(m + ".net.0.weight"), d(m + ".net.0.bias"), M, C, y16=ws["xn"], mean=st[2], rstd=st[3],
                                   w=d(pre + ".proj_down.0.weight"), bias=d(pre + ".proj_down.0.bias"), z=g["zx"], y=g["xl"], act=1, w_layout=0,
                                   L_=self.Lat, **split)
        else:
            ops.layernorm_fwd(g1, d(m + ".net.0.weight"), d(m + ".net.0.bias"), M, C, y16=ws["xn"], mean=st[2], rstd=st[3])

    def _mlp_block_fwd(self, ws, i, si, g1, gout, M, train, pdrop=0.0, up_in_fc2=False):
        nm, w, d, C = self.names, self._w16, self._d, self.C
        m = nm.mlp(i)
        if self._keep_inputs:
            ops.copy_(ws["sav"]["xn2"][si], ws["xn"])
        self._gemm(ws["xn"], w[f"fc1{i}"], M, ws["pre"][si] if train else None, epilogue=ops.EPI_BIAS_GELU_BF16, out1=ws["act"],
                    bias=d(m + ".net.1.bias"),           # inference keeps no pre-activation (out0 = NULL)
                    ldo=self.ldx, drop_p=pdrop, seed=SEED_LAYER + 8 * i + 2, seed_ptr=ws["seed"])
        if self._keep_inputs:
            ops.copy_(ws["sav"]["act"][si], ws["act"])
        self._gemm(ws["act"], w[f"fc2{i}"], M, gout, epilogue=ops.EPI_BIAS_RES_F32, bias=d(m + ".net.4.bias"), res=g1,
                   K=self.ldx if up_in_fc2 else self.mlp,      # the GPA latents ride this GEMM as 64 extra K columns (self._fuse_up)
                   alg_k=self.mlp + self.Lat if up_in_fc2 else None,
                   drop_p=pdrop, seed=SEED_LAYER + 8 * i + 3, seed_ptr=ws["seed"])

    # ---- GAViKO side paths --------------------------------------------------------------------------------------
    def _mwsa_fwd(self, ws, sv, i, si, lin, lout, gpa_local=False):
        """MWSA of layer i on the local stream (gaviko.py:229-244).  With self._fuse_next the up-projection kernel of layer i also runs layer
        i+1's entry (LayerNorm + proj_down + qkv of the rows it writes), so only layer 0 launches the entry kernel itself."""
        if not _on("noside"):
            return
        pre = f"transformer.local_attns.{i // self.share}"
        d, C, Lt, B = self._d, self.C, self.Lat, ws["B"]
        BN = B * self.N
        m = ws["mw"][si]
        chained = self._fuse_next and gpa_local and _on("loc_noupdown")
        if _on("loc_noupdown") and not (chained and i > 0):
            ops.skinny_down(x=lin, w=d(pre + ".proj_down.weight"), bias=d(pre + ".proj_down.bias"), ln_gamma=d(pre + ".norm.weight"),
                            ln_beta=d(pre + ".norm.bias"), mean=m["mean"], rstd=m["rstd"], y=m["lat"], w2=d(pre + ".qkv.weight"), y2=m["qkv"],
                            M=BN, C=C, L=Lt, L2=3 * Lt, act=0, w_layout=0, eps=1e-5)
        if _on("nowin"):
            ops.window_attn_fwd(qkv=m["qkv"], ctx=m["ctx"], lse=m["lse"], B=B, D=self.grid[0], H=self.grid[1], W=self.grid[2],
                                kd=self.win[0], kh=self.win[1], kw=self.win[2], L=Lt, scale=C ** -0.5, drop_p=sv["attn_drop"],
                                seed=2 * i, seed_ptr=ws["seed"])
        second = {}
        if gpa_local:                                       # ll = QuickGELU(proj_down(L')) (gaviko.py:156) of the rows this launch produces
            gpre, _ = self._gpa_names(i)
            g = ws["gp"][si]
            second = dict(w2=d(gpre + ".proj_down.0.weight"), bias2=d(gpre + ".proj_down.0.bias"), z2=g["zl"], y2=g["ll"], L2=Lt, act2=1)
        if chained and i + 1 < self.depth:                  # layer i+1's norm + proj_down + qkv of the same rows
            nx = f"transformer.local_attns.{(i + 1) // self.share}"
            mn = ws["mw"][si + 1 if sv["train"] else 0]
            second.update(nx_w=d(nx + ".proj_down.weight"), nx_bias=d(nx + ".proj_down.bias"), nx_ln_gamma=d(nx + ".norm.weight"),
                          nx_ln_beta=d(nx + ".norm.bias"), nx_mean=mn["mean"], nx_rstd=mn["rstd"], nx_lat=mn["lat"], nx_w2=d(nx + ".qkv.weight"),
                          nx_y2=mn["qkv"], nx_L2=3 * Lt, nx_eps=1e-5)
        if _on("loc_noupdown"):
            ops.skinny_up(lat=m["ctx"], w=d(pre + ".proj_up.weight"), bias=d(pre + ".proj_up.bias"), res=lin, out=lout, M=BN, C=C, L=Lt,
                          w_layout=0, drop_p=sv["proj_drop"], seed=2 * i + 1, seed_ptr=ws["seed"], **second)

    def _gpa_names(self, i):
        s = i // self.share
        pre = f"transformer.prompt_projs.{s}"
        ca, gl = pre + ".cls_analyzer.cls_analyzer_", pre + ".gl_balancer.gl_balancer_"
        return pre, dict(ca0_g=ca + ".0.weight", ca0_b=ca + ".0.bias", ca1_w=ca + ".1.weight", ca1_b=ca + ".1.bias", ca3_w=ca + ".3.weight",
                         ca3_b=ca + ".3.bias", gl0_g=gl + ".0.weight", gl0_b=gl + ".0.bias", gl1_w=gl + ".1.weight", gl1_b=gl + ".1.bias",
                         wgq=pre + ".global_attention.query_proj.weight", bgq=pre + ".global_attention.query_proj.bias",
                         wlq=pre + ".local_attention.query_proj.weight", blq=pre + ".local_attention.query_proj.bias")

    def _gpa_down_local(self, ws, i, si, lnew, B):
        """ll = QuickGELU(proj_down(L')) (gaviko.py:156): depends on the MWSA chain only, so it runs at its tail."""
        if "noside" in _ABLATE:
            return
        pre, _ = self._gpa_names(i)
        d, g = self._d, ws["gp"][si]
        ops.skinny_down(x=lnew, w=d(pre + ".proj_down.0.weight"), bias=d(pre + ".proj_down.0.bias"), z=g["zl"], y=g["ll"], M=B * self.N,
                        C=self.C, L=self.Lat, act=1, w_layout=0)

    def _gpa_fwd_latents(self, ws, i, si, g1, lnew, M, B, project, enh16=None):
        if "noside" in _ABLATE:
            return
        pre, names = self._gpa_names(i)
        d, C, Lt = self._d, self.C, self.Lat
        g = ws["gp"][si]
        if project:
            ops.skinny_down(x=g1, w=d(pre + ".proj_down.0.weight"), bias=d(pre + ".proj_down.0.bias"), z=g["zx"], y=g["xl"], M=M, C=C, L=Lt,
                            act=1, w_layout=0)
            self._gpa_down_local(ws, i, si, lnew, B)
        slot = {} if enh16 is None else dict(enh16=enh16, ld16=enh16.shape[-1], col16=self.mlp)
        ops.gpa_fwd(xl=g["xl"], ll=g["ll"], B=B, T=self.T, N=self.N, P=self.P, L=Lt, scale=Lt ** -0.5,
                    imp=g["imp"], gw=g["gw"], enh=g["enh"], prm=g["prm"], qg=g["qg"], ql=g["ql"], cg=g["cg"], cl=g["cl"],
                    lse_g=g["lse_g"], lse_l=g["lse_l"], **slot, **{k: d(v) for k, v in names.items()})

    def _gpa_fwd_up(self, ws, i, si, gout, M):
        pre, _ = self._gpa_names(i)
        d, g = self._d, ws["gp"][si]
        ops.skinny_up(lat=g["xl"], w=d(pre + ".proj_up.weight"), bias=d(pre + ".proj_up.bias"), out=gout, lat_override=g["enh"],
                      M=M, C=self.C, L=self.Lat, T=self.T, P=self.P, w_layout=0, accumulate=1)

    # ------------------------------------------------------------------ backward
    def trainable_names(self) -> List[str]:
        return [k for k, p in self.p.items() if p.requires_grad]

    def _grad_views(self, device) -> Dict[str, torch.Tensor]:
        names = self.trainable_names()
        sig = tuple((n, tuple(self.p[n].shape)) for n in names)
        if self._flat_grad is None or self._flat_grad["sig"] != sig or self._flat_grad["buf"].device != device:
            total = sum(self.p[n].numel() for n in names)
            buf = torch.zeros(total, device=device)
            views, off = {}, 0
            for n in names:
                k = self.p[n].numel()
                views[n] = buf[off: off + k].view(self.p[n].shape)
                off += k
            self._flat_grad = {"sig": sig, "buf": buf, "views": views}
        return self._flat_grad["views"]

    @property
    def flat_grad(self) -> Optional[torch.Tensor]:
        return None if self._flat_grad is None else self._flat_grad["buf"]

    def backward(self, dlogits: torch.Tensor, reducer=None) -> Dict[str, torch.Tensor]:
        """Fills and returns {param name: gradient view into the flat fp32 gradient buffer} for every trainable tensor.
        The sweep is cut into segments at the reducer's bucket boundaries (one segment without a reducer); each segment is a
        HIP graph after warm-up.  `reducer` (distributed.GradReducer) is told after every segment which layers are done, so
        finished buckets of the flat buffer are all-reduced on its side stream while the next segment runs."""
        sv = self._saved
        if sv is None:
            raise L.GavikoHipError("backward() without a preceding training-mode forward()")
        ws = self._ws
        gv = self._grad_views(dlogits.device)
        unsupported = [n for n in gv if not self._grad_supported(n)]
        if unsupported:
            raise NotImplementedError(f"gradients for backbone tensors are not built yet (frozen-backbone PEFT only): {unsupported[:3]}...")
        if dlogits.data_ptr() != ws["dlogits"].data_ptr():
            ws["dlogits"].copy_(dlogits.detach())
        flat = self._flat_grad["buf"]
        if reducer is not None:
            reducer.begin()
        if not self._needs_backbone_backward():
            self._run("bwd_head", self._saved_key, lambda: self._backward_head(ws, sv, gv, False))
            if reducer is not None:
                reducer.finish(flat)
            return gv
        if reducer is not None and getattr(reducer, "mode", "segments") == "events" and STEP_MODE != "graph":
            # ONE plan; every bucket is reduced behind the event recorded on the stream that finalises it (no cut, no join)
            self._want_bucket_marks = True
            self._bucket_marks = {}
            try:
                self._run("bwd0", self._saved_key + ("events",),
                          lambda: self._backward_segment(ws, sv, gv, self.depth - 1, 0, True, True))
            finally:
                self._want_bucket_marks = False
            how, pid = self._last_run
            if how in ("replayed", "recorded"):
                marks, lib = self.plan_bucket_marks[pid], L.load()
                waiter = lambda stream, ev: L.check(lib.gvk_plan_event_stream_wait(pid, ev, stream.cuda_stream), "gvk_plan_event_stream_wait")
            else:
                marks = self._bucket_marks
                waiter = lambda stream, ev: stream.wait_event(ev)
            reducer.reduce_marked(flat, marks, waiter)
            return gv
        cuts = sorted({r for r, _, _ in reducer.ranges if r >= 0}, reverse=True) if reducer is not None else []
        cuts = [c for c in cuts if 0 < c < self.depth]          # segment k ends (inclusive) at layer cuts[k]
        hi = self.depth - 1
        for seg, lo in enumerate(cuts + [0]):
            first, last = seg == 0, lo == 0
            self._run(f"bwd{seg}", self._saved_key + (tuple(cuts),),
                      lambda hi=hi, lo=lo, first=first, last=last: self._backward_segment(ws, sv, gv, hi, lo, first, last))
            if reducer is not None:
                reducer.layer_done(flat, lo)
            hi = lo - 1
        if reducer is not None:
            reducer.finish(flat)
        return gv

    def _backward_head(self, ws, sv, gv, backbone_bwd):
        nm, d = self.names, self._d
        B, C, T = sv["B"], self.C, self.T
        r0, R = self._pool_rows()
        dG = ws["dG"][0] if backbone_bwd else None
        if backbone_bwd:
            ops.memset_zero(dG)
        ops.head_bwd(g=self._final_stream(ws, True), ln_gamma=d(nm.root + "transformer.norm.weight"),
                     ln_beta=d(nm.root + "transformer.norm.bias"), wh=d(nm.head() + ".weight"), bh=d(nm.head() + ".bias"), pooled=ws["pooled"],
                     dlogits=ws["dlogits"], dg=dG, dwh=gv[nm.head() + ".weight"], dbh=gv[nm.head() + ".bias"], B=B, T=self.Ts[-1], C=C,
                     K=self.K, r0=r0, R=R, accumulate=0)
        bb = sv.get("bb") or ()
        if backbone_bwd and ("transformer.norm.weight" in bb or "transformer.norm.bias" in bb):
            g, bw = self._final_stream(ws, True), ws["bbw"]
            ops.layernorm_fwd(g, d("transformer.norm.weight"), d("transformer.norm.bias"), B * T, C, y16=ws["xn"], mean=bw["stat"][0], rstd=bw["stat"][1])
            ops.ssf_head_grad(g, bw["stat"][0], bw["stat"][1], d(nm.head() + ".weight"), ws["dlogits"], bw["ones"][:C], bw["zeros"][:C],
                              gv["transformer.norm.weight"] if "transformer.norm.weight" in bb else bw["junk"][:C],
                              gv["transformer.norm.bias"] if "transformer.norm.bias" in bb else bw["junk"][C: 2 * C], B, T, C, self.K, r0, R)
        if self.kind == "ssf" and backbone_bwd:
            # final norm + ssf (ssf.py:138): statistics of the final stream, then the pooled rows' scale / shift gradients
            g = self._final_stream(ws, True)
            st = ws["ssf_stat"]
            ops.layernorm_fwd(g, self.p["transformer.norm.weight"].detach(), self.p["transformer.norm.bias"].detach(), B * T, C, y16=ws["xn"],
                              mean=st[0], rstd=st[1])
            ops.ssf_head_grad(g, st[0], st[1], d(nm.head() + ".weight"), ws["dlogits"], self.p["transformer.norm.weight"].detach(),
                              self.p["transformer.norm.bias"].detach(), gv["transformer.ssf_scale_1"], gv["transformer.ssf_shift_1"],
                              B, T, C, self.K, r0, R)
        if backbone_bwd:
            ops.to_operand(dG, ws["dG16"], self.adt)
            if self.kind == "gaviko":
                ops.memset_zero(ws["dL"][0])

    def _backward_segment(self, ws, sv, gv, hi, lo, first, last):
        """Layers hi, hi-1, ..., lo of the backward sweep (+ the head when `first`, + the embedding rows when `last`).
        ws['dG'][0] always holds the gradient of the global stream at a layer boundary, ws['dG'][1] the mid-layer one;
        the local-stream gradient ping-pongs with the layer parity."""
        if first:
            self._mwsa_pending = None
        nm, w, d = self.names, self._w16, self._d
        B, C, T, M = sv["B"], self.C, self.T, sv["B"] * self.T
        gaviko = self.kind == "gaviko"
        if first:
            self._backward_head(ws, sv, gv, True)
        dGout, dGin = ws["dG"][0], ws["dG"][1]
        if gaviko and ((self.depth - 1 - hi) & 1):
            dGout = ws["dGb"]                                  # the boundary gradient ping-pongs with the layer parity (see below)
        vpt_deep = self.kind == "vpt" and self.deep
        if vpt_deep and ((self.depth - 1 - hi) & 1):           # the un-repack alternates two buffers with the layer parity
            dGout = ws["dGv"]
        if gaviko:
            loc, gpa = self._stream("loc"), self._stream("gpa")
            self._wait("gpa", None)
            self._wait("loc", None)
        prev_scl = None
        for i in range(hi, lo - 1, -1):
            M = B * self.Ts[i]
            T = self.Ts[i]
            par = (self.depth - 1 - i) & 1
            m, a = nm.mlp(i), nm.attn(i)
            st = ws["stat"][i]
            # GPA stream: the critical kernels (-> dzx, dzl) first, then the parameter gradients; the other streams wait only
            # for the event between the two
            if gaviko:
                with torch.cuda.stream(gpa):
                    # dcomb = dGout . W_up was produced by the LayerNorm backward that wrote dGout, except for the top layer
                    self._gpa_bwd_core(ws, sv, gv, i, dGout, M, B, par, project=not (self._fuse_proj and i < self.depth - 1))
                    dz_ready = self._ev_record(gpa)
                    self._gpa_bwd_params(ws, sv, gv, i, dGout, M, B, par)
                    self._bucket_mark("gpa", i)                              # prompt_projs.{i // share} gradients final once the lowest layer using it is done
            # main stream, MLP block: dG1 = dGout + LN'(fc1^T(GELU'(pre) * fc2^T(dGout)))
            self._mark(f"b{i}:start")
            bb = sv.get("bb") or ()
            pd_ = sv.get("bdrop", 0.0)
            dy_ff = dGout
            if pd_ > 0:                                                      # gradient of dropout(fc2(.)): the forward's mask on dGout
                dy_ff = self._masked_grad(ws, dGout, pd_, SEED_LAYER + 8 * i + 3, bool(bb), M)
            if bb:                                                           # fc2: db = colsum(dGout), dW = dGout^T . act
                self._bb_linear_grads(ws, gv, bb, m + ".net.4", dy_ff, ws["dG16"], ws["sav"]["act"][i] if sv["wgrad"] else None, M, C, self.mlp)
            dvpt = self.kind == "dvpt"
            if dvpt:
                self._dvpt_bwd_latents(ws, gv, i, dGout, M, B)
            ssf = self.kind == "ssf"
            if ssf:                                                          # fc2 + ssf_2: dy = dGout, y = G[i+1] - G1[i]
                self._ssf_linear_grad(ws, gv, m, 2, dGout, ws["G"][i + 1], M, C, y1=ws["G1"][i])
            self._gemm(ws["dG16"], w[f"fc2{i}_t"], M, ws["dpre"], epilogue=ops.EPI_GELU_BWD_BF16, aux=ws["pre"][i], ldaux=self.ldx,
                       drop_p=pd_, seed=SEED_LAYER + 8 * i + 2, seed_ptr=ws["seed"])
            if ssf:                                                          # fc1 + ssf_1: dy = d(pre-activation), y = saved pre-activation
                self._ssf_linear_grad(ws, gv, m, 1, ws["dpre"], ws["pre"][i], M, self.mlp)
            if bb:                                                           # fc1: db = colsum(dpre), dW = dpre^T . LN2(G1)
                self._bb_linear_grads(ws, gv, bb, m + ".net.1", ws["dpre"], ws["dpre"], ws["sav"]["xn2"][i] if sv["wgrad"] else None, M, self.mlp, C)
            self._mark(f"b{i}:fc2d") if False else None
            self._gemm(ws["dpre"], w[f"fc1{i}_t"], M, ws["dx32"], epilogue=ops.EPI_STORE_F32)
            self._mark(f"b{i}:fc1d")
            if bb:
                self._bb_ln_grads(ws, gv, bb, m + ".net.0", ws["dx32"], ws["G1"][i], st[2], st[3], M)
            if ssf:                                                          # LN2 + ssf_0
                self._ssf_ln_grad(ws, gv, m, ".net.0", ws["dx32"], ws["G1"][i], st[2], st[3], M)
            adapter = self.kind == "adaptformer"
            fuse_scatter = (gaviko and self._fuse_local and not self.fp32 and "noside" not in _ABLATE
                            and os.environ.get("GAVIKO_HIP_FUSE_SCATTER", "0") == "1")     # measured: 651 vs 676 volumes/s -- off
            if fuse_scatter:
                # dG1 = dGout + LN'(dx32) + dzx . W_d (+ the bf16 operand of the out-proj dgrad) in ONE pass: the GPA core of this layer
                # (started at the top of the layer on its own stream) has long finished when the two MLP dgrad GEMMs are through
                self._ev_wait(torch.cuda.current_stream(), dz_ready)
                gpre, _ = self._gpa_names(i)
                ops.layernorm_bwd_up(ws["dx32"], ws["G1"][i], st[2], st[3], d(m + ".net.0.weight"), M, C, dx=dGin, dres=dGout, dx16=ws["dG16"],
                                     lat=ws["bw"]["dzx"], w=d(gpre + ".proj_down.0.weight"), L_=self.Lat, w_layout=1)
            else:
                ops.layernorm_bwd(ws["dx32"], ws["G1"][i], st[2], st[3], d(m + ".net.0.weight"), M, C, dx=dGin, dres=dGout,
                                  dx16=None if (gaviko or adapter or dvpt) else ws["dG16"])
            if dvpt:
                self._dvpt_bwd_scatter(ws, i, dGin, M)                       # dG1 += (dz . Wd) * QuickGELU'(G1)  (+ operand copy)
            if adapter:
                self._adapter_bwd(ws, gv, i, dGout, dGin, M)                 # adds LN_a'(...) into dG1 and refreshes dG16
            self._mark(f"b{i}:ln2")
            if gaviko:
                if not fuse_scatter:
                    self._ev_wait(torch.cuda.current_stream(), dz_ready)
                    self._gpa_bwd_scatter_g(ws, i, dGin, M)                  # dG1 += dzx.Wd (+ bf16 copy)
                self._mark(f"b{i}:scatter")
                # The MWSA chain of this layer (~170 us of kernels against ~290 us of backbone work per layer) runs beside the attention
                # backward, which it slows by 21 % (tools/plan_marks.py, locnop ablation); GAVIKO_HIP_LOC_SHIFT=1 holds it back until that
                # is through (not a gain, see _LOC_SHIFT)
                shift = _LOC_SHIFT and i > lo
                if not shift:
                    self._mwsa_chain_bwd(ws, sv, gv, i, par, B, loc, dz_ready)
            # main stream, attention block: dG0 = dG1 + LN'(qkv^T(attn'(out^T(dG1))))
            if ssf:                                                          # to_out + ssf_2: dy = dG1, y = G1[i] - G[i]
                self._ssf_linear_grad(ws, gv, a, 2, dGin, ws["G1"][i], M, C, y1=ws["G"][i])
            dy_at = dGin
            if pd_ > 0:                                                      # gradient of dropout(to_out(.))
                dy_at = self._masked_grad(ws, dGin, pd_, SEED_LAYER + 8 * i + 1, bool(bb), M)
            if bb:                                                           # to_out: db = colsum(dG1), dW = dG1^T . ctx
                self._bb_linear_grads(ws, gv, bb, a + ".to_out.0", dy_at, ws["dG16"], ws["ctx"][i], M, C, C)
            self._gemm(ws["dG16"], w[f"out{i}_t"], M, ws["dctx"], epilogue=ops.EPI_STORE_BF16)
            self._mark(f"b{i}:outd")
            ops.attention_bwd(ws["qkv"][i], ws["ctx"][i], ws["dctx"], ws["lse"][i], ws["delta"], ws["dqkv"], B, T, self.heads, 64 ** -0.5,
                              drop_p=pd_, seed=SEED_LAYER + 8 * i, seed_ptr=ws["seed"], q_prescaled=True)
            self._mark(f"b{i}:attnb")
            if gaviko and shift:
                self._mwsa_chain_bwd(ws, sv, gv, i, par, B, loc, self._ev_record(torch.cuda.current_stream()))
            if self.kind == "melo":
                self._melo_bwd(ws, gv, i, M)
            if ssf:                                                          # to_qkv + ssf_1: dy = dqkv, y = saved qkv
                uq = {} if self.fp32 else dict(y0_cols=self.heads * 64, y0_mul=1.0 / self.q_scale)     # the saved q block is pre-scaled
                self._ssf_linear_grad(ws, gv, a, 1, ws["dqkv"], ws["qkv"][i], M, 3 * C, **uq)
            if bb:                                                           # to_qkv (bias-free): dW = dqkv^T . LN1(G)
                self._bb_linear_grads(ws, gv, bb, a + ".to_qkv", None, ws["dqkv"], ws["sav"]["xn1"][i] if sv["wgrad"] else None, M, 3 * C, C)
            self._gemm(ws["dqkv"], w[f"qkv{i}_t"], M, ws["dx32"], epilogue=ops.EPI_STORE_F32)
            if bb:
                self._bb_ln_grads(ws, gv, bb, a + ".norm", ws["dx32"], ws["G"][i], st[0], st[1], M)
            if ssf:                                                          # LN1 + ssf_0
                self._ssf_ln_grad(ws, gv, a, ".norm", ws["dx32"], ws["G"][i], st[0], st[1], M)
            self._mark(f"b{i}:qkvd")
            if gaviko:
                # The new boundary gradient goes to the OTHER parity buffer: the GPA parameter kernels of this layer keep reading
                # dGout off the critical path.  The buffer being overwritten was last read by layer i+1's parameter kernels, which
                # precede this layer's dz_ready in the GPA stream -- and the main stream has already waited for that above.
                dGnext = ws["dGb"] if dGout is ws["dG"][0] else ws["dG"][0]
                if self._fuse_proj and i > 0:
                    pre_lo, _ = self._gpa_names(i - 1)
                    ops.layernorm_bwd_proj(ws["dx32"], ws["G"][i], st[0], st[1], d(a + ".norm.weight"), M, C, dx=dGnext, dres=dGin,
                                           dx16=ws["dG16"], w=d(pre_lo + ".proj_up.weight"), y=ws["bw"]["dcomb"], w_layout=1, L_=self.Lat)
                else:
                    ops.layernorm_bwd(ws["dx32"], ws["G"][i], st[0], st[1], d(a + ".norm.weight"), M, C, dx=dGnext, dres=dGin,
                                      dx16=ws["dG16"])
                dGout = dGnext
            else:
                ops.layernorm_bwd(ws["dx32"], ws["G"][i], st[0], st[1], d(a + ".norm.weight"), M, C, dx=dGout, dres=dGin, dx16=ws["dG16"])
            if gaviko:
                # The MWSA chain never feeds the global stream in the backward, so the main stream does not join it per layer: dzl
                # is double-buffered by layer parity and the only cross-stream hazard left is layer i-1's GPA rewriting the buffer
                # layer i+1's scatter read -- ordered by making the GPA stream (not the main one) wait for THAT long finished kernel.
                if prev_scl is not None:
                    self._ev_wait(gpa, prev_scl)
                prev_scl = self._scl_done
                self._wait("gpa", None)                                      # the next layer's GPA backward needs this dG[i]
            if self.kind == "evp":
                self._evp_bwd_layer(ws, gv, i, dGout, B)
            self._mark(f"b{i}:end")
            if not gaviko:
                self._bucket_mark("main", i)                                 # transformer.layers.{i}.* gradients (adapters, LoRA, ...) are final
            if self.kind == "vpt" and (i == 0 or self.deep):
                # prompt rows 1..P of this layer's input are this layer's projected prompts (vpt.py:127-131,147-153)
                if sv.get("pdrop", 0.0) > 0:                                 # through prompt_dropout: same mask, in place (these rows end here)
                    ops.dropout_rows(dGout, sv["pdrop"], SEED_PROMPT + i, ws["seed"], out32=dGout, M=B * self.P, N=C, rows_in=self.P, rows_out=T, row_off=1)
                ops.rows_batch_sum(dGout, ws["dvproj"][i * self.P: (i + 1) * self.P], None, B, T, 1, self.P, C)
            if vpt_deep and i > 0:
                other = ws["dGv"] if dGout is ws["dG"][0] else ws["dG"][0]
                ops.vpt_repack_bwd(dGout, other, B, self.Ts[i - 1], T, self.P, self.pd, C)
                dGout = other
                ops.to_operand(dGout, ws["dG16"], self.adt)
        if gaviko:
            if last:                                                         # (the deferred step crosses segment boundaries like layer boundaries)
                self._mwsa_flush(ws, B, loc)
            self._wait(None, "gpa")
            self._wait(None, "loc")
        if last and sv.get("bb"):
            if sv.get("edrop", 0.0) > 0:                                     # through emb_dropout
                ops.dropout_rows(dGout, sv["edrop"], SEED_EMB, ws["seed"], out32=dGout, M=B * self.T, N=C)
            self._bb_embed_grads(ws, gv, sv["bb"], dGout, B)
        if last and self.kind == "evp":
            self._evp_bwd_finish(ws, gv, B)
        if last and self.kind == "ssf":
            # patch embedding + ssf (ssf.py:229-232): dy = the patch rows of the input gradient, y = G[0] patch rows - pos[1:]
            pos = self.p["pos_embedding"].detach()[0]
            ops.ssf_colgrad(dGout, ws["G"][0], self.p["ssf_scale_1"].detach(), self.p["ssf_shift_1"].detach(), gv["ssf_scale_1"], gv["ssf_shift_1"],
                            ws["ssf_scratch"], B * self.N, C, pos=pos[1:], rows_in=self.N, rows_out=T, row_off=1)
        if last and self.kind == "vpt":
            emb_name = "deep_prompt_embeddings" if self.deep else "prompt_embeddings"
            emb = d(emb_name).reshape(-1, self.pd)
            ops.small_linear_bwd(emb, d("prompt_proj.weight"), ws["dvproj"], gv["prompt_proj.weight"], gv["prompt_proj.bias"],
                                 gv[emb_name].view(-1, self.pd), emb.shape[0], self.pd, C)
        if last and (gaviko or self.kind == "dvpt"):
            ops.rows_batch_sum(dGout, gv["prompt_embeddings"].view(self.P, C), gv["prompt_positional_embedding"].view(self.P, C), B, T, 0,
                               self.P, C)
        if last:
            self._bucket_mark("main", -1)                                    # everything else (prompts, head, unindexed tensors): end of the sweep

    def _grad_supported(self, name: str) -> bool:
        if name.startswith(self.names.head()) or self.kind == "vit":       # plain ViT: every tensor (`linear` / `bitfit` / `fft`)
            return True
        if self.kind == "gaviko":
            return ("local_attns" in name or "prompt_projs" in name or name in ("prompt_embeddings", "prompt_positional_embedding"))
        if self.kind == "vpt":
            return name in ("prompt_proj.weight", "prompt_proj.bias", "deep_prompt_embeddings", "prompt_embeddings")
        if self.kind == "adaptformer":
            return "adapter" in name
        if self.kind == "melo":
            return ".linear_a_" in name or ".linear_b_" in name
        if self.kind == "ssf":
            return "ssf_scale_" in name or "ssf_shift_" in name
        if self.kind == "dvpt":
            return "prompt" in name
        if self.kind == "evp":
            return "prompt_generator" in name
        return False

    def _needs_backbone_backward(self) -> bool:
        return any(not n.startswith(self.names.head()) for n in self.trainable_names())

    def _acc(self, i) -> int:
        """0 when layer i is the first (highest) layer of the sweep that touches its shared side-path module, else 1."""
        s = i // self.share
        top = min(self.depth - 1, s * self.share + self.share - 1)
        return 0 if i == top else 1

    def _gpa_bwd_core(self, ws, sv, gv, i, dGout, M, B, par, project=True):
        """Critical part of the GPA backward: dcomb = dGout . Wup and the latent-space backward -> dzx / dzl
        (what the main stream's dG1 update and the MWSA chain wait for)."""
        if "noside" in _ABLATE:
            return
        pre, names = self._gpa_names(i)
        d, C, Lt, P, T, N = self._d, self.C, self.Lat, self.P, self.T, self.N
        g, bw = ws["gp"][i], ws["bw"]
        if project:
            ops.skinny_down(x=dGout, w=d(pre + ".proj_up.weight"), y=bw["dcomb"], M=M, C=C, L=Lt, act=0, w_layout=1)
        ops.gpa_bwd(xl=g["xl"], ll=g["ll"], B=B, T=T, N=N, P=P, L=Lt, scale=Lt ** -0.5, imp=g["imp"], gw=g["gw"], enh=g["enh"], prm=g["prm"],
                    qg=g["qg"], ql=g["ql"], cg=g["cg"], cl=g["cl"], lse_g=g["lse_g"], lse_l=g["lse_l"], dcomb=bw["dcomb"], zx=g["zx"], zl=g["zl"],
                    dimp=bw["dimp"], dgw_part=bw["dgw_part"], dqg=bw["dqg"], dql=bw["dql"], dcg=bw["dcg"], dcl=bw["dcl"],
                    delta_g=bw["delta_g"], delta_l=bw["delta_l"], dprm=bw["dprm"], dcls=bw["dcls"], gate_partials=bw["gate_partials"],
                    dzx=bw["dzx"], dzl=bw["dzl"][par], **{k: d(v) for k, v in names.items()})

    def _gpa_bwd_params(self, ws, sv, gv, i, dGout, M, B, par):
        """Off the critical path: every parameter gradient of the GPA module (reads dGout, dzx, dzl, saved activations)."""
        if "noside" in _ABLATE or "noparams" in _ABLATE:
            return
        pre, names = self._gpa_names(i)
        d, C, Lt, P, T, N = self._d, self.C, self.Lat, self.P, self.T, self.N
        g, bw, sc = ws["gp"][i], ws["bw"], ws["scratch"]
        acc = self._acc(i)
        ops.outer_reduce(narrow=g["xl"], wide=dGout, lat_override=g["enh"], scratch=sc, out=gv[pre + ".proj_up.weight"],
                         colsum=gv[pre + ".proj_up.bias"], M=M, C=C, L=Lt, T=T, P=P, transposed=1, accumulate=acc)
        # gate parameters: one contiguous slice of the flat gradient buffer, in the kernel's order
        ng = ops.gpa_gate_param_count(Lt, P)
        first = gv[names["ca0_g"]]
        gate_flat = self._flat_grad["buf"][self._offset_of(names["ca0_g"]): self._offset_of(names["ca0_g"]) + ng]
        assert gate_flat.data_ptr() == first.data_ptr()
        gwd, gbd = gv[pre + ".proj_down.0.weight"], gv[pre + ".proj_down.0.bias"]
        BP = B * P
        dqg, dql, prm = bw["dqg"].view(BP, Lt), bw["dql"].view(BP, Lt), g["prm"].view(BP, Lt)
        ops.reduce_batch([(bw["gate_partials"], None, gate_flat, acc),
                          (dqg, prm, gv[names["wgq"]], acc), (dqg, None, gv[names["bgq"]], acc),
                          (dql, prm, gv[names["wlq"]], acc), (dql, None, gv[names["blq"]], acc),
                          (bw["dzx"], None, gbd, acc, bw["dzl"][par])], ws["rscratch"])      # proj_down bias: both token streams
        # proj_down (shared by both streams): dWd = dzx^T.G1 + dzl^T.Lnew
        if M + B * N <= ops.OUTER_MAX_ROWS:                                   # both token streams in one pass
            ops.outer_reduce(narrow=bw["dzx"], wide=ws["G1"][i], narrow2=bw["dzl"][par], wide2=ws["Lc"][i + 1], scratch=sc, out=gwd, M=M, M2=B * N,
                             C=C, L=Lt, transposed=0, accumulate=acc)
        else:
            ops.outer_reduce(narrow=bw["dzx"], wide=ws["G1"][i], scratch=sc, out=gwd, M=M, C=C, L=Lt, transposed=0, accumulate=acc)
            ops.outer_reduce(narrow=bw["dzl"][par], wide=ws["Lc"][i + 1], scratch=sc, out=gwd, M=B * N, C=C, L=Lt, transposed=0, accumulate=1)

    def _gpa_bwd_scatter_g(self, ws, i, dG1, M):
        """main stream: dG1 += dzx . Wd, with the bf16 copy for the out-proj dgrad."""
        pre, _ = self._gpa_names(i)
        ops.skinny_up(lat=ws["bw"]["dzx"], w=self._d(pre + ".proj_down.0.weight"), out=dG1, out_bf16=None if self.fp32 else ws["dG16"],
                      M=M, C=self.C, L=self.Lat, w_layout=1, accumulate=1)
        if self.fp32:
            ops.copy_(ws["dG16"], dG1)

    def _mwsa_chain_bwd(self, ws, sv, gv, i, par, B, loc, after):
        """Local stream: dL += dzl . Wd (GPA's share), then the MWSA backward of layer i; starts once event `after` is reached.
        The last step of a layer's MWSA backward (dL_in = dL_out + LN'(dlat . Wd)) is deferred to the start of the next-lower layer's
        chain, where ONE kernel does it together with that layer's scatter and its first down-projection (self._fuse_bnd)."""
        self._ev_wait(loc, after)
        with torch.cuda.stream(loc):
            pend, fused = self._mwsa_pending, False
            if pend is not None and self._fuse_bnd and "noside" not in _ABLATE and "loc_noupdown" not in _ABLATE:
                self._mwsa_boundary(ws, sv, pend, i, par, B)
                fused = True
            else:
                if pend is not None:
                    self._mwsa_final(ws, pend, B)
                self._gpa_bwd_scatter_l(ws, i, ws["dL"][par], B, par)        # dL[par] was written on this stream
            self._scl_done = self._ev_record(loc)
            self._mwsa_bwd(ws, sv, gv, i, ws["dL"][par], ws["dL"][par ^ 1], B, have_dctx=fused, defer_final=True)
            self._mwsa_pending = i
            self._bucket_mark("loc", i)

    def _mwsa_flush(self, ws, B, loc):
        """End of a backward plan: the deferred last step of the lowest layer of the sweep."""
        if self._mwsa_pending is not None:
            with torch.cuda.stream(loc):
                self._mwsa_final(ws, self._mwsa_pending, B)
            self._mwsa_pending = None

    def _mwsa_final(self, ws, j, B):
        """dL_in = dL_out + LN'(dlat . Wd) of layer j (gaviko.py:231): the rank-L product never touches HBM."""
        if "noside" in _ABLATE or "loc_noupdown" in _ABLATE:
            return
        pre = f"transformer.local_attns.{j // self.share}"
        d, m, par = self._d, ws["mw"][j], (self.depth - 1 - j) & 1
        ops.skinny_up(lat=ws["bw"]["dlat"], w=d(pre + ".proj_down.weight"), res=ws["dL"][par], out=ws["dL"][par ^ 1], ln_x=ws["Lc"][j],
                      ln_mean=m["mean"], ln_rstd=m["rstd"], ln_gamma=d(pre + ".norm.weight"), M=B * self.N, C=self.C, L=self.Lat, w_layout=1)

    def _mwsa_boundary(self, ws, sv, j, i, par, B):
        """Layer j = i + 1's last step, layer i's GPA scatter and layer i's dctx = proj_drop'(dL) . Wup in one pass over the local-stream
        gradient (gvk_skinny_up with lat_b): dL[par] = dL[par ^ 1] + LN'(dlat_j . Wd_j) + dzl_i . Wd_gpa_i;  dctx_i = (dL[par] o mask_i) . Wup_i."""
        pj, pi_ = f"transformer.local_attns.{j // self.share}", f"transformer.local_attns.{i // self.share}"
        gpre, _ = self._gpa_names(i)
        d, m, bw = self._d, ws["mw"][j], ws["bw"]
        ops.skinny_up(lat=bw["dlat"], w=d(pj + ".proj_down.weight"), res=ws["dL"][par ^ 1], out=ws["dL"][par], ln_x=ws["Lc"][j],
                      ln_mean=m["mean"], ln_rstd=m["rstd"], ln_gamma=d(pj + ".norm.weight"), M=B * self.N, C=self.C, L=self.Lat, w_layout=1,
                      lat_b=bw["dzl"][par], w_b=d(gpre + ".proj_down.0.weight"),
                      w2=d(pi_ + ".proj_up.weight"), z2=bw["dctx"], L2=self.Lat, act2=0, w2_layout=1,
                      drop2_p=sv["proj_drop"], seed2=2 * i + 1, seed_ptr=ws["seed"])

    def _gpa_bwd_scatter_l(self, ws, i, dLnew, B, par):
        """MWSA chain: dL += dzl . Wd."""
        if "noside" in _ABLATE:
            return
        pre, _ = self._gpa_names(i)
        if _on("loc_noupdown"):
            ops.skinny_up(lat=ws["bw"]["dzl"][par], w=self._d(pre + ".proj_down.0.weight"), out=dLnew, M=B * self.N, C=self.C, L=self.Lat,
                          w_layout=1, accumulate=1)

    # ---- AdaptFormer (adaptformer.py:58-78, 93-97): r = up(ReLU(down(LN_a(x)))), x_out = ff(x) + x + r -----------------------
    def _adapter_prefix(self, i):
        return f"transformer.layers.{i}.1"

    def _adapter_shadows(self, train):
        """bf16 MFMA operands of the TRAINABLE adapter weights, refreshed in place every step (inside the captured graph)."""
        w = self._w16
        for i in range(self.depth):
            p = self._adapter_prefix(i)
            wd, wu = self._d(p + ".down_adapter_proj.weight"), self._d(p + ".up_adapter_proj.weight")
            w[f"ad_d{i}"] = ops.to_operand(wd, None if self.fp32 else w.get(f"ad_d{i}"), self.adt)
            w[f"ad_u{i}"] = ops.to_operand(wu, None if self.fp32 else w.get(f"ad_u{i}"), self.adt)
            if train:
                w[f"ad_dT{i}"] = ops.transpose_operand(wd, w.get(f"ad_dT{i}"), self.adt)
                w[f"ad_uT{i}"] = ops.transpose_operand(wu, w.get(f"ad_uT{i}"), self.adt)

    def _adapter_fwd_down(self, ws, i, si, g1, M):
        p, d, ad = self._adapter_prefix(i), self._d, ws["ad"][si]
        ops.layernorm_fwd(g1, d(p + ".adapter_layer_norm_before.weight"), d(p + ".adapter_layer_norm_before.bias"), M, self.C, y16=ws["xa"],
                          mean=ad["mean"], rstd=ad["rstd"])
        self._gemm(ws["xa"], self._w16[f"ad_d{i}"], M, ad["h16"], epilogue=ops.EPI_BIAS_RELU_BF16, bias=d(p + ".down_adapter_proj.bias"))

    def _adapter_fwd_up(self, ws, i, si, gout, M):
        p = self._adapter_prefix(i)
        self._gemm(ws["ad"][si]["h16"], self._w16[f"ad_u{i}"], M, gout, epilogue=ops.EPI_BIAS_RES_F32, bias=self._d(p + ".up_adapter_proj.bias"),
                   res=gout)

    def _adapter_bwd(self, ws, gv, i, dGout, dG1, M):
        p, d, C, A = self._adapter_prefix(i), self._d, self.C, self.adim
        ad, w, sc = ws["ad"][i], self._w16, ws["scratch"]
        g, b = d(p + ".adapter_layer_norm_before.weight"), d(p + ".adapter_layer_norm_before.bias")
        # up-projection: dh = (dGout . Wu) * [h > 0]; dWu = dGout^T . h; dbu = colsum(dGout)
        self._gemm(ws["dG16"], w[f"ad_uT{i}"], M, ws["dh16"], epilogue=ops.EPI_RELU_BWD_BF16, aux=ad["h16"])
        ops.cast_bf16_f32_strided(ad["h16"], ws["h32"], M, A, A)
        ops.cast_bf16_f32_strided(ws["dh16"], ws["dh32"], M, A, A)
        ops.outer_reduce(narrow=ws["h32"], wide=dGout, scratch=sc, out=gv[p + ".up_adapter_proj.weight"], colsum=gv[p + ".up_adapter_proj.bias"],
                         M=M, C=C, L=A, transposed=1, accumulate=0)
        # down-projection: dxa = dh . Wd; dWd = dh^T . LN_a(G1); dbd = colsum(dh)
        self._gemm(ws["dh16"], w[f"ad_dT{i}"], M, ws["dx32"], epilogue=ops.EPI_STORE_F32)
        ops.outer_reduce(narrow=ws["dh32"], wide=ws["G1"][i], mean=ad["mean"], rstd=ad["rstd"], ln_gamma=g, ln_beta=b, scratch=sc,
                         out=gv[p + ".down_adapter_proj.weight"], M=M, C=C, L=A, transposed=0, accumulate=0)
        ops.colsum(ws["dh32"], gv[p + ".down_adapter_proj.bias"], sc, M, A)
        # trainable LayerNorm in front of the adapter: input gradient accumulates into dG1, affine gradients
        ops.layernorm_bwd(ws["dx32"], ws["G1"][i], ad["mean"], ad["rstd"], g, M, C, dx=dG1, dres=dG1, dx16=ws["dG16"])
        ops.layernorm_bwd_affine(ws["dx32"], ws["G1"][i], ad["mean"], ad["rstd"], gv[p + ".adapter_layer_norm_before.weight"],
                                 gv[p + ".adapter_layer_norm_before.bias"], sc, M, C)

    # ---- MeLO / LoRA (melo.py:41-47): qkv = W x + s B_q A_q x (q columns) + s B_v A_v x (v columns) -------------------------
    def _lora_names(self, i):
        q = self.names.attn(i) + ".to_qkv"
        return q + ".linear_a_q.weight", q + ".linear_b_q.weight", q + ".linear_a_v.weight", q + ".linear_b_v.weight"

    def _melo_merge(self, ws, train):
        """Fold the rank-r update into the bf16 QKV operand (and its transpose) every step: the forward is then the plain GEMM."""
        w, C = self._w16, self.C
        for i in range(self.depth):
            aq, bq, av, bv = (self._d(n) for n in self._lora_names(i))
            ops.lora_merge(self._d(self.names.qkv_weight(i)), aq, bq, av, bv, ws["merge32"], C, self.r, self.lora_s)
            if self.fp32 and not isinstance(w.get(f"qkv{i}_own"), torch.Tensor):
                w[f"qkv{i}_own"] = torch.empty_like(ws["merge32"])     # the merged weight needs its own buffer per layer
            w[f"qkv{i}"] = ops.to_operand(ws["merge32"], w[f"qkv{i}_own"] if self.fp32 else w.get(f"qkv{i}"), self.adt)
            if train:
                w[f"qkv{i}_t"] = ops.transpose_operand(ws["merge32"], w.get(f"qkv{i}_t"), self.adt)

    def _melo_bwd(self, ws, gv, i, M):
        """dB = s dq^T u, dA = s (dq B)^T LN(x), u = LN(x) A^T -- all rank-r fp32 kernels over the bf16 dq / dv blocks."""
        C, r, d, sc = self.C, self.r, self._d, ws["scratch"]
        na_q, nb_q, na_v, nb_v = self._lora_names(i)
        a = self.names.attn(i)
        g1, b1, st, x = d(a + ".norm.weight"), d(a + ".norm.bias"), ws["stat"][i], ws["G"][i]
        lu = ws["lu"]
        ops.cast_bf16_f32_strided(ws["dqkv"], ws["dq32"], M, C, 3 * C, col0=0)
        ops.cast_bf16_f32_strided(ws["dqkv"], ws["dv32"], M, C, 3 * C, col0=2 * C)
        for na, nb, dblk, u, du in ((na_q, nb_q, ws["dq32"], lu["uq"], lu["duq"]), (na_v, nb_v, ws["dv32"], lu["uv"], lu["duv"])):
            ops.skinny_down(x=x, w=d(na), ln_gamma=g1, ln_beta=b1, y=u, M=M, C=C, L=r, act=0, w_layout=0, eps=1e-5)
            ops.outer_reduce(narrow=u, wide=dblk, scratch=sc, out=gv[nb], M=M, C=C, L=r, transposed=1, accumulate=0)
            ops.skinny_down(x=dblk, w=d(nb), y=du, M=M, C=C, L=r, act=0, w_layout=1)
            ops.outer_reduce(narrow=du, wide=x, mean=st[0], rstd=st[1], ln_gamma=g1, ln_beta=b1, scratch=sc, out=gv[na], M=M, C=C, L=r,
                             transposed=0, accumulate=0)
            if self.lora_s != 1:
                ops.scale_(gv[na], float(self.lora_s))
                ops.scale_(gv[nb], float(self.lora_s))

    # ---- unfrozen ViT tensors (`bitfit` / `fft`, train.py:123-137): biases and LayerNorm affines from column sums, weights from
    #      wgrad GEMMs dW = dY^T . X run as NT GEMMs over the transposed operands (contraction over the padded token count) -------
    def _bb_buffers(self, ws, B, device, wgrad):
        if "bbw" not in ws:
            C, M = self.C, B * self.T
            n = max(self.mlp, 3 * C, self.Kp)
            ws["bbw"] = dict(ones=torch.ones(n, device=device), zeros=torch.zeros(n, device=device), junk=torch.zeros(2 * n, device=device),
                             scratch=torch.zeros(64 * 2 * n, device=device), stat=[torch.zeros(M, device=device), torch.zeros(M, device=device)])
            ws["dyd"] = ops.act_zeros(M, C, torch.float32, device)           # dropout-masked copy of a layer gradient (bias / weight-gradient operand)
        if wgrad and "sav" not in ws:
            C, M, Mp = self.C, B * self.T, ops.pad_rows(B * self.T)
            z = lambda r, c: ops.act_zeros(r, c, self.adt, device)
            ws["sav"] = dict(xn1=[z(M, C) for _ in range(self.depth)], xn2=[z(M, C) for _ in range(self.depth)],
                             act=[z(M, self.mlp) for _ in range(self.depth)])
            ws["tA"] = z(max(self.mlp, 3 * C), Mp)
            ws["tB"] = z(max(self.mlp, self.Kp), Mp)
            ws["pg16"] = z(B * self.N, C)
            ws["pg32"] = ops.act_zeros(B * self.N, C, torch.float32, device)

    def _bb_wgrad(self, ws, dy_op, x_op, out, M, N, K):
        """out [N][K] (fp32) = dy_op[0:M, 0:N]^T . x_op[0:M, 0:K]; rows >= M of both operand buffers are zero by construction."""
        Mp = ops.pad_rows(M)
        tA, tB = ws["tA"].view(-1)[: ops.pad_rows(N) * Mp].view(-1, Mp), ws["tB"].view(-1)[: ops.pad_rows(K) * Mp].view(-1, Mp)
        ops.transpose_any(dy_op, tA, Mp, N)
        ops.transpose_any(x_op, tB, Mp, K)
        self._gemm(tA, tB[:K], N, out.view(N, K), epilogue=ops.EPI_STORE_F32)

    def _bb_linear_grads(self, ws, gv, bb, prefix, dy32, dy_op, x_op, M, N, K):
        """db = colsum(dy), dW = dy^T . x for one Linear of the backbone, for whichever of the two trains."""
        bw = ws["bbw"]
        if prefix + ".bias" in bb:
            src = dy32 if dy32 is not None else dy_op
            ops.colsum_any(src, gv[prefix + ".bias"], bw["ones"][:N], bw["zeros"][:N], bw["junk"][:N], bw["scratch"], M, N)
        if prefix + ".weight" in bb:
            self._bb_wgrad(ws, dy_op, x_op, gv[prefix + ".weight"], M, N, K)

    def _bb_ln_grads(self, ws, gv, bb, prefix, dy, x, mean, rstd, M):
        wn, bn = prefix + ".weight", prefix + ".bias"
        if wn in bb or bn in bb:
            C, junk = self.C, ws["bbw"]["junk"]
            ops.layernorm_bwd_affine(dy, x, mean, rstd, gv[wn] if wn in bb else junk[:C], gv[bn] if bn in bb else junk[C: 2 * C], ws["scratch"], M, C)

    def _bb_embed_grads(self, ws, gv, bb, dG0, B):
        """pos_embedding / cls_token (batch sums of the input gradient), conv_proj bias and weight (the patch rows)."""
        C, T, N, bw = self.C, self.T, self.N, ws["bbw"]
        nm = self.names
        if "pos_embedding" in bb:
            ops.rows_batch_sum(dG0, gv["pos_embedding"].view(T, C), None, B, T, 0, T, C)
        if "cls_token" in bb:
            ops.rows_batch_sum(dG0, gv["cls_token"].view(1, C), None, B, T, 0, 1, C)
        cw, cb = nm.conv() + ".weight", nm.conv() + ".bias"
        if cb in bb:
            ops.colsum_any(dG0, gv[cb], bw["ones"][:C], bw["zeros"][:C], bw["junk"][:C], bw["scratch"], B * N, C, rows_in=N, rows_out=T, row_off=1)
        if cw in bb:
            ops.rows_gather(dG0, ws["pg32"], B, T, N, C, 1)
            ops.to_operand(ws["pg32"], ws["pg16"], self.adt)
            self._bb_wgrad(ws, ws["pg16"], ws["cols"], gv[cw].view(C, self.Kp), B * N, C, self.Kp)

    # ---- EVP (evp.py): prompts from a high-pass copy of the volume + the patch embeddings, added in front of every layer --------
    def _evp_state(self, device):
        """Static per-engine buffers: the high-pass operator, zero-padded copies of the trainable prompt-generator weights and of
        their gradients (the latents have rank r = dim/32 = 6 / 24 / 32; the rank-L kernels run at the padded width Lp)."""
        st = self.__dict__.get("_evp")
        if st is not None:
            return st
        C, Lp, Kp = self.C, self.Lp, self.Kp
        mk = lambda *s_: torch.zeros(s_, device=device)
        D, H, W = (g * p_ for g, p_ in zip(self.grid, self.patch))
        hp, dm = evp_highpass_operator(D, H, W, self.freq)
        st = dict(hp=torch.from_numpy(hp).to(device), dmask=torch.from_numpy(dm).to(device),
                  Wp=mk(64, Kp), bp=mk(64), We=mk(Lp, C), be=mk(Lp), Ws=mk(C, Lp),
                  Wi=[mk(Lp, Lp) for _ in range(self.depth)], WiT=[mk(Lp, Lp) for _ in range(self.depth)], bi=[mk(Lp) for _ in range(self.depth)],
                  dWs=mk(C, Lp), dWe=mk(Lp, C), dWp=mk(Lp, Kp), dvec=mk(Lp), dWi=mk(Lp, Lp))
        self.__dict__["_evp"] = st
        return st

    def _evp_latents(self, ws, B):
        """s = proj(highpass(img)) + embedding_generator(conv(img))  (evp.py:76-84, 347-349), at the padded width."""
        st, ev, d = self._evp_state(ws["img"].device), ws["ev"], self._d
        r, Lp, C, Kp, BN = self.r, self.Lp, self.C, self.Kp, B * self.N
        pg = "prompt_generator."
        ops.pad2d(d(pg + "prompt_generator.proj.weight").reshape(r, Kp), r, Kp, st["Wp"], 64, Kp)
        ops.pad2d(d(pg + "prompt_generator.proj.bias"), 1, r, st["bp"], 1, 64)
        ops.pad2d(d(pg + "embedding_generator.weight"), r, C, st["We"], Lp, C)
        ops.pad2d(d(pg + "embedding_generator.bias"), 1, r, st["be"], 1, Lp)
        ops.pad2d(d(pg + "shared_mlp.weight"), C, r, st["Ws"], C, Lp)
        for i in range(self.depth):
            wi = d(pg + f"lightweight_mlp_{i}.0.weight")
            ops.pad2d(wi, r, r, st["Wi"][i], Lp, Lp)
            ops.pad2d(wi, r, r, st["WiT"][i], Lp, Lp, transpose=True)
            ops.pad2d(d(pg + f"lightweight_mlp_{i}.0.bias"), 1, r, st["bi"][i], 1, Lp)
        ops.skinny_down(x=ws["xc"], w=st["We"], bias=st["be"], y=ev["e"], M=BN, C=C, L=Lp, act=0, w_layout=0)
        ops.evp_highpass(ws["img"], st["hp"], st["dmask"], ws["hp"])
        ops.patchify(ws["hp"], ws["hcols"], self.patch)
        ops.gemm_nt(ws["hcols"], st["Wp"], BN, ws["hc"], epilogue=ops.EPI_STORE_F32, bias=st["bp"])      # fp32 GEMM in both precisions (1.6 GF)
        ops.add2d(ws["hc"], 64, ev["e"], Lp, ev["s"], Lp, BN, Lp)

    def _evp_add_prompt(self, ws, i, si, g, B):
        """prompt_i = shared_mlp(GELU(lightweight_mlp_i(s)))  (evp.py:86-95), added to the patch rows of the layer input."""
        st, ev, d = self._evp_state(g.device), ws["ev"], self._d
        Lp, C, BN = self.Lp, self.C, B * self.N
        ops.small_linear_fwd(ev["s"], st["Wi"][i], st["bi"][i], ev["pre"][si], BN, Lp, Lp)
        ops.gelu_fwd(ev["pre"][si], ev["u"][si])
        ops.skinny_up(lat=ev["u"][si], w=st["Ws"], bias=d("prompt_generator.shared_mlp.bias"), out=ev["tmp"], M=BN, C=C, L=Lp, w_layout=0)
        ops.rows_patch(g, ev["tmp"], None, B, self.T, self.N, C, 1, True)

    def _evp_bwd_layer(self, ws, gv, i, dG, B):
        """d prompt_i = the patch rows of the gradient of layer i's input; accumulates d shared_mlp over the layers and d s."""
        st, ev, bw = self._evp_state(dG.device), ws["ev"], ws["evb"]
        Lp, C, BN, r = self.Lp, self.C, B * self.N, self.r
        pg = "prompt_generator."
        top = i == self.depth - 1
        ops.rows_gather(dG, ev["tmp"], B, self.T, self.N, C, 1)
        ops.outer_reduce(narrow=ev["u"][i], wide=ev["tmp"], scratch=ws["scratch"], out=st["dWs"], colsum=gv[pg + "shared_mlp.bias"], M=BN, C=C, L=Lp,
                         transposed=1, accumulate=0 if top else 1)
        ops.skinny_down(x=ev["tmp"], w=st["Ws"], y=bw["du"], M=BN, C=C, L=Lp, act=0, w_layout=1)
        ops.gelu_bwd(bw["du"], ev["pre"][i], bw["dpre"])
        ops.reduce_batch([(bw["dpre"], ev["s"], st["dWi"], 0), (bw["dpre"], None, st["dvec"], 0)], ws["rscratch"])
        ops.pad2d(st["dWi"], r, r, gv[pg + f"lightweight_mlp_{i}.0.weight"], r, r, ld_src=Lp)
        ops.pad2d(st["dvec"], 1, r, gv[pg + f"lightweight_mlp_{i}.0.bias"], 1, r, ld_src=Lp)
        ops.small_linear_fwd(bw["dpre"], st["WiT"][i], None, bw["ds"] if top else bw["ds_tmp"], BN, Lp, Lp)       # d s = dpre . W_i
        if not top:
            ops.add2d(bw["ds"], Lp, bw["ds_tmp"], Lp, bw["ds"], Lp, BN, Lp)

    def _evp_bwd_finish(self, ws, gv, B):
        st, ev, bw = self._evp_state(ws["img"].device), ws["ev"], ws["evb"]
        Lp, C, BN, r, Kp = self.Lp, self.C, B * self.N, self.r, self.Kp
        pg = "prompt_generator."
        ops.pad2d(st["dWs"], C, r, gv[pg + "shared_mlp.weight"], C, r, ld_src=Lp)
        ops.outer_reduce(narrow=bw["ds"], wide=ws["xc"], scratch=ws["scratch"], out=st["dWe"], M=BN, C=C, L=Lp, transposed=0, accumulate=0)
        ops.pad2d(st["dWe"], r, C, gv[pg + "embedding_generator.weight"], r, C)
        ops.reduce_batch([(bw["ds"], None, st["dvec"], 0)], ws["rscratch"])
        ops.pad2d(st["dvec"], 1, r, gv[pg + "embedding_generator.bias"], 1, r, ld_src=Lp)
        ops.pad2d(st["dvec"], 1, r, gv[pg + "prompt_generator.proj.bias"], 1, r, ld_src=Lp)
        ops.outer_reduce(narrow=bw["ds"], wide=ws["hcols"], scratch=ws["scratch"], out=st["dWp"], M=BN, C=Kp, L=Lp, transposed=0, accumulate=0)
        ops.pad2d(st["dWp"], r, Kp, gv[pg + "prompt_generator.proj.weight"].view(r, Kp), r, Kp)

    # ---- DVPT (dvpt.py:24-63): share_MLP beside the MLP block ------------------------------------------------------------------
    def _dvpt_names(self, i):
        p = f"transformer.layers.{i}.0.prompt_proj"
        return p + ".prompt_key_proj_d", p + ".prompt_key_proj_u", p + ".prompt_gate"

    def _dvpt_fwd_latents(self, ws, i, si, g1, M, B):
        pd, pu, pg = self._dvpt_names(i)
        d, v = self._d, ws["dv"][si]
        ops.skinny_down(x=g1, w=d(pd + ".weight"), bias=d(pd + ".bias"), y=v["z"], M=M, C=self.C, L=self.Lat, act=0, w_layout=0, act_in=1)
        ops.dvpt_fwd(z=v["z"], enh=v["enh"], lse=v["lse"], B=B, T=self.T, P=self.P, L=self.Lat, C=self.C, scale=self.C ** -0.5)

    def _dvpt_fwd_up(self, ws, i, si, gout, M):
        pd, pu, pg = self._dvpt_names(i)
        d, v = self._d, ws["dv"][si]
        ops.skinny_up(lat=v["z"], lat_override=v["enh"], w=d(pu + ".weight"), bias=d(pu + ".bias"), alpha_ptr=d(pg), out=gout, M=M, C=self.C,
                      L=self.Lat, T=self.T, P=self.P, w_layout=0, accumulate=1)

    def _dvpt_bwd_latents(self, ws, gv, i, dGout, M, B):
        """dcomb = dGout . W_u;  dW_u, db_u (gate applied afterwards), dgate;  latent backward -> dz;  db_d, dW_d."""
        pd, pu, pg = self._dvpt_names(i)
        d, v, bw, C, Lt = self._d, ws["dv"][i], ws["dvb"], self.C, self.Lat
        ops.skinny_down(x=dGout, w=d(pu + ".weight"), y=bw["dcomb"], M=M, C=C, L=Lt, act=0, w_layout=1)
        ops.outer_reduce(narrow=v["z"], lat_override=v["enh"], wide=dGout, scratch=ws["scratch"], out=gv[pu + ".weight"], colsum=gv[pu + ".bias"],
                         M=M, C=C, L=Lt, T=self.T, P=self.P, transposed=1, accumulate=0)
        ops.dvpt_bwd(z=v["z"], enh=v["enh"], lse=v["lse"], dcomb=bw["dcomb"], gate=d(pg), bu=d(pu + ".bias"), colsum_dy=gv[pu + ".bias"],
                     delta=bw["delta"], dz=bw["dz"], dgate=gv[pg], B=B, T=self.T, P=self.P, L=Lt, C=C, scale=C ** -0.5)
        ops.scale_dev_(gv[pu + ".weight"], d(pg))
        ops.scale_dev_(gv[pu + ".bias"], d(pg))
        ops.reduce_batch([(bw["dz"], None, gv[pd + ".bias"], 0)], ws["rscratch"])
        ops.outer_reduce(narrow=bw["dz"], wide=ws["G1"][i], scratch=ws["scratch"], out=gv[pd + ".weight"], M=M, C=C, L=Lt, transposed=0,
                         accumulate=0, wide_act=1)

    def _dvpt_bwd_scatter(self, ws, i, dG1, M):
        pd, pu, pg = self._dvpt_names(i)
        ops.skinny_up(lat=ws["dvb"]["dz"], w=self._d(pd + ".weight"), out=dG1, out_bf16=None if self.fp32 else ws["dG16"], gg_x=ws["G1"][i],
                      M=M, C=self.C, L=self.Lat, w_layout=1, accumulate=1)
        if self.fp32:
            ops.copy_(ws["dG16"], dG1)

    # ---- SSF (ssf.py): effective parameters per step, scale / shift gradients per site -----------------------------------------
    def _ssf_sites(self):
        """(scale name, shift name, kind, target) for every ssf_ada of the model, in forward order."""
        nm = self.names
        sites = [("ssf_scale_1", "ssf_shift_1", "linear", ("conv", "conv_proj.0.weight", "conv_proj.0.bias"))]
        for i in range(self.depth):
            a, m = nm.attn(i), nm.mlp(i)
            sites += [(a + ".ssf_scale_0", a + ".ssf_shift_0", "ln", (a + ".norm.weight", a + ".norm.bias")),
                      (a + ".ssf_scale_1", a + ".ssf_shift_1", "linear", (f"qkv{i}", a + ".to_qkv.weight", a + ".to_qkv.bias")),
                      (a + ".ssf_scale_2", a + ".ssf_shift_2", "linear", (f"out{i}", a + ".to_out.0.weight", a + ".to_out.0.bias")),
                      (m + ".ssf_scale_0", m + ".ssf_shift_0", "ln", (m + ".net.0.weight", m + ".net.0.bias")),
                      (m + ".ssf_scale_1", m + ".ssf_shift_1", "linear", (f"fc1{i}", m + ".net.1.weight", m + ".net.1.bias")),
                      (m + ".ssf_scale_2", m + ".ssf_shift_2", "linear", (f"fc2{i}", m + ".net.4.weight", m + ".net.4.bias"))]
        sites.append(("transformer.ssf_scale_1", "transformer.ssf_shift_1", "ln", ("transformer.norm.weight", "transformer.norm.bias")))
        return sites

    def _ssf_fold(self, train):
        """gamma' = gamma*s, beta' = beta*s + t;  W' = s[:,None]*W (operand dtype, + transpose for the dgrad), b' = b*s + t.
        Runs inside the recorded step: the scales and shifts are what the optimiser updates."""
        w, eff, raw = self._w16, self._eff, (lambda n: self.p[n].detach())
        for sn, tn, kind, tgt in self._ssf_sites():
            s_, t_ = raw(sn), raw(tn)
            if kind == "ln":
                gname, bname = tgt
                if gname not in eff:
                    eff[gname], eff[bname] = torch.empty_like(s_), torch.empty_like(s_)
                ops.ssf_fold_vec(raw(gname), s_, None, eff[gname])
                ops.ssf_fold_vec(raw(bname), s_, t_, eff[bname])
            else:
                key, wname, bname = tgt
                W = raw(wname)
                W2 = W.reshape(W.shape[0], -1)
                if key not in w:
                    w[key] = torch.empty(W2.shape, dtype=self.adt, device=W.device)
                    eff[bname] = torch.empty_like(s_)
                need_t = train and key != "conv"
                if need_t and key + "_t" not in w:
                    w[key + "_t"] = torch.empty((W2.shape[1], W2.shape[0]), dtype=self.adt, device=W.device)
                ops.ssf_fold_weight(W2, s_, w[key], w[key + "_t"] if need_t else None)
                ops.ssf_fold_vec(self.p[bname].detach() if bname in self.p else None, s_, t_, eff[bname])

    def _ssf_linear_grad(self, ws, gv, prefix, idx, dy, y0, M, N, y1=None, **kw):
        sn, tn = f"{prefix}.ssf_scale_{idx}", f"{prefix}.ssf_shift_{idx}"
        ops.ssf_colgrad(dy, y0, self.p[sn].detach(), self.p[tn].detach(), gv[sn], gv[tn], ws["ssf_scratch"], M, N, y1=y1, **kw)

    def _ssf_ln_grad(self, ws, gv, prefix, ln, dy, x, mean, rstd, M):
        C, tmp = self.C, ws["ssf_tmp"]
        ops.layernorm_bwd_affine(dy, x, mean, rstd, tmp[:C], tmp[C:], ws["scratch"], M, C)
        ops.ssf_ln_grad(tmp[:C], tmp[C:], self.p[prefix + ln + ".weight"].detach(), self.p[prefix + ln + ".bias"].detach(),
                        gv[prefix + ".ssf_scale_0"], gv[prefix + ".ssf_shift_0"])

    def _offset_of(self, name) -> int:
        return (self._flat_grad["views"][name].data_ptr() - self._flat_grad["buf"].data_ptr()) // 4

    def _mwsa_bwd(self, ws, sv, gv, i, dLout, dLin, B, have_dctx=False, defer_final=False):
        """MWSA backward of layer i on the local stream.  have_dctx: the layer-boundary kernel already produced dctx (_mwsa_boundary);
        defer_final: the last step (dL_in) is left to the next-lower layer's boundary kernel (_mwsa_chain_bwd).  The `_on(...)` guards are
        the timing ablations of DESIGN.md section 7b.3."""
        if not _on("noside"):
            return
        pre = f"transformer.local_attns.{i // self.share}"
        d, C, Lt = self._d, self.C, self.Lat
        BN = B * self.N
        m, bw, sc = ws["mw"][i], ws["bw"], ws["scratch_l"]
        lin = ws["Lc"][i]
        acc = self._acc(i)
        pd, seed_p, seed_a, sp = sv["proj_drop"], 2 * i + 1, 2 * i, ws["seed"]
        if _on("loc_noupdown") and not have_dctx:
            ops.skinny_down(x=dLout, w=d(pre + ".proj_up.weight"), y=bw["dctx"], M=BN, C=C, L=Lt, act=0, w_layout=1, drop_p=pd, seed=seed_p,
                            seed_ptr=sp)
        if _on("loc_noouter"):
            ops.outer_reduce(narrow=m["ctx"], wide=dLout, scratch=sc, out=gv[pre + ".proj_up.weight"], colsum=gv[pre + ".proj_up.bias"],
                             M=BN, C=C, L=Lt, transposed=1, accumulate=acc, drop_p=pd, seed=seed_p, seed_ptr=sp)
        if _on("nowin"):
            ops.window_attn_bwd(qkv=m["qkv"], ctx=m["ctx"], lse=m["lse"], dctx=bw["dctx"], delta=bw["wdelta"], dqkv=bw["dqkv"], B=B,
                                D=self.grid[0], H=self.grid[1], W=self.grid[2], kd=self.win[0], kh=self.win[1], kw=self.win[2], L=Lt,
                                scale=C ** -0.5, drop_p=sv["attn_drop"], seed=seed_a, seed_ptr=sp)
        if _on("loc_nosmall"):
            ops.skinny_down(x=bw["dqkv"], w=d(pre + ".qkv.weight"), y=bw["dlat"], M=BN, C=3 * Lt, L=Lt, act=0, w_layout=1)
        wd = d(pre + ".proj_down.weight")
        g_, b_ = d(pre + ".norm.weight"), d(pre + ".norm.bias")
        if _on("loc_noouter"):
            # Q[l][c] = sum_m dlat[m][l] xhat[m][c], S[l] = sum_m dlat[m][l]  ->  dWd, dbd, dgamma, dbeta in one tiny kernel
            ops.outer_reduce(narrow=bw["dlat"], wide=lin, mean=m["mean"], rstd=m["rstd"], scratch=sc, out=bw["Q"], M=BN, C=C, L=Lt,
                             transposed=0, accumulate=0)
        if _on("loc_nosmall"):
            # qkv weight gradient (dqkv^T . lat) and S[l] = sum_m dlat[m][l] in one two-stage reduction
            ops.reduce_batch([(bw["dqkv"], m["lat"], gv[pre + ".qkv.weight"], acc), (bw["dlat"], None, bw["S"], 0)], ws["rscratch_l"])
            ops.ln_lowrank_affine(bw["Q"], bw["S"], wd, g_, b_, gv[pre + ".proj_down.weight"], gv[pre + ".norm.weight"],
                                  gv[pre + ".norm.bias"], gv[pre + ".proj_down.bias"], Lt, C, accumulate=bool(acc))
        if _on("loc_noupdown") and not defer_final:
            # dL_in = dL_out + LN'(dlat . Wd): the rank-L product never touches HBM
            ops.skinny_up(lat=bw["dlat"], w=wd, res=dLout, out=dLin, ln_x=lin, ln_mean=m["mean"], ln_rstd=m["rstd"], ln_gamma=g_, M=BN, C=C,
                          L=Lt, w_layout=1)
