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

from .engine_common import (SEED_EMB, SEED_LAYER, SEED_PROMPT, GRAPH_WARMUP, Names, PLAN_TIMING, SIDE_STREAM_PRIORITY, STEP_MODE, USE_GRAPHS, _ABLATE, _EPI_NAMES, _FIX_IN_LN, _LOC_SHIFT, _MODE, _SIDE_STREAMS, _on, evp_highpass_operator)  # noqa: F401
from .engine_gaviko import GavikoPaths
from .engine_peft import PeftPaths

# bench.py instrumentation: when a dict, every GEMM launch is bracketed by HIP events recorded on the launch stream
# bench.py instrumentation: when set to a dict, plans recorded from then on bracket every GEMM launch (and the patch-embed
# stage) with timestamped plan events, so the kernels are timed inside the real three-stream schedule of a replayed step.
GEMM_MARKS = None
# the bf16 attention backward as ONE pass (csrc/attention_bwd.hip::attn_bwd_fused_kernel, round 5): built, bit-compatible, and 5 % slower
# than the two passes at T = 1033 -- the two-pass kernels stay the path; GAVIKO_HIP_ATTN_BWD=fused (measurement build) is the A/B
_ATTN_FUSED = L.diag_env("GAVIKO_HIP_ATTN_BWD", "2pass") == "fused"
# the K loops of the strided-panel GEMMs (dead-row pruning) cut into pieces over idle CUs (gvk_gemm_desc.splitk_ws): +0.4 % on the step, but the
# sum of partial sums is no longer the bit pattern the full-size GEMM produces -- off, so that pruning stays bit-identical (measurement build: =1)
_PANEL_SPLITK = L.diag_env("GAVIKO_HIP_PANEL_SPLITK", "0") == "1"


# classes whose backbone tensors can train (engine_peft.py `_bb_*`): the plain ViT (`fft` / `bitfit`); every class that freezes by default, with freeze_vit=False
_BB_KINDS = ("vit", "adaptformer", "gaviko", "dvpt", "evp", "vpt", "ssf")


class Engine(GavikoPaths, PeftPaths):
    def __init__(self, kind: str, cfg: dict, params: Dict[str, torch.nn.Parameter], depth, heads, dim, mlp_dim):
        self.kind, self.cfg, self.p = kind, cfg, params
        self.depth, self.heads, self.C, self.mlp = depth, heads, dim, mlp_dim
        self.lora_layers = frozenset(cfg.get("lora_layer") or range(depth)) if kind == "melo" else frozenset()
        self.names = Names(kind, self.lora_layers if kind == "melo" else None)
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
                           and L.diag_env("GAVIKO_HIP_FUSE_PROJ", "1") != "0")
        # backward only: dcomb = dG . W_up inside the LayerNorm-1 backward on the main stream ("main"), or by the GPA stream's own projection
        # kernel in front of its backward core ("gpa": one more 12.7 MB read there, a plain LayerNorm backward here)
        self._proj_bwd_main = self._fuse_proj and L.diag_env("GAVIKO_HIP_PROJ_BWD", "main") != "gpa"
        # Rows nobody consumes are not computed (round 5).  With a frozen backbone the loss reads the LAST layer's output only at the rows
        # the head pools (prompts + CLS, gaviko.py:316), so that layer's MLP -- a row-wise function -- runs on those rows in the forward, and
        # its backward (fc2 / fc1 dgrad, LayerNorm 2) on the same rows: every other row of the incoming gradient is an exact zero.  At the
        # other end only the P prompt rows of the FIRST layer's input carry a trainable tensor (patch embedding, cls token and position
        # embedding are frozen: gaviko.py:540-548), so its qkv dgrad and LayerNorm-1 backward run on those rows.  Logits and every gradient
        # are bit-identical to the full computation (tests/test_model_gpu.py::test_pruned_rows_are_dead); GAVIKO_HIP_PRUNE=0 computes
        # everything (the golden taps of the last layer's far rows need that).
        self.prune_dead_rows = kind == "gaviko" and not self.fp32 and os.environ.get("GAVIKO_HIP_PRUNE", "1") != "0"
        # frozen backbone: the fc1 / qkv dgrad GEMMs hand the LayerNorm backward its input gradient in bf16 (half the bytes on both sides; the other
        # operands of that chain -- dpre, dqkv, the dgrad operands -- are bf16 already)
        self._dy16 = kind == "gaviko" and not self.fp32 and L.diag_env("GAVIKO_HIP_DY16", "1") != "0"
        # ... and fc1's forward epilogue leaves GELU'(pre) instead of pre (it has the exponential in hand and VALU to spare behind its two output
        # streams); the fc2 dgrad epilogue then multiplies instead of evaluating the derivative (isolated: 33.7 -> 30.3 us with a free derivative)
        self._gelu_grad = kind == "gaviko" and not self.fp32 and L.diag_env("GAVIKO_HIP_GELU_GRAD", "1") != "0"
        self._fuse_local = kind == "gaviko" and ops.side_tile_supported(self.Lat, dim)
        self._fuse_bnd = self._fuse_local and L.diag_env("GAVIKO_HIP_FUSE_BOUNDARY", "1") != "0"
        self._fuse_next = self._fuse_local and L.diag_env("GAVIKO_HIP_FUSE_NEXT", "1") != "0"
        self._mwsa_pending = None
        # every parameter gradient of a side-path module of one layer in ONE launch per stream (csrc/paramgrad.hip; round 4)
        self._pgrad = (kind == "gaviko" and ops.param_grads_supported(cfg.get("prompt_latent_dim", 20), dim)
                       and L.diag_env("GAVIKO_HIP_PGRAD", "1") != "0")
        # GPA up-projection as K-concatenation of the MLP's second Linear: 64 spare K columns carry the rank-L product in split-bf16 form,
        # A' = [act | lat_hi | lat_lo | lat_hi | 1 | 1], W' = [W_fc2 | Wup_hi | Wup_hi | Wup_lo | b_hi | b_lo] (fp32-grade: the dropped
        # lo.lo term is 2^-16 relative), so x + ff(x) + proj_up(.) (gaviko.py:187 after vision_transformer.py:34) is ONE GEMM and the
        # main stream loses a full read-modify-write pass over the token stream per layer
        self._fuse_up = (kind == "gaviko" and not self.fp32 and 3 * self.Lat + 2 <= 64 and L.diag_env("GAVIKO_HIP_FUSE_UP", "1") != "0")
        self.ldx = self.mlp + 64 if self._fuse_up else self.mlp      # row stride of the MLP hidden buffers
        # First LayerNorm of layers 1.. folded into their qkv projection (vision_transformer.py:49,61-62): the fc2 GEMM of the layer below
        # leaves the bf16 copy of its output rows and per-row (sum, sum of squares) partials, the prompt-fix kernel turns them into mean /
        # rstd, and the qkv GEMM runs on the RAW rows against gamma o W with  rstd * (acc - mean * c1) + c2  in its epilogue -- one launch
        # (and its dependency gap) less per layer on the main stream.  128-column tiles only (C % 128 == 0: ViT-B / ViT-L).
        self._fold_ln1 = (self._fuse_up and self._fuse_proj and dim % 128 == 0 and L.diag_env("GAVIKO_HIP_FOLD_LN1", "1") != "0")
        self._fold: Dict[str, torch.Tensor] = {}
        self._fold_version = None
        self._fold_names = frozenset(self._fold_deps()) if self._fold_ln1 else frozenset()
        self._fold_on = False
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
        self._flat_names = None
        self.bucket_layers = 4              # layout of the flat gradient buffer: completion groups of this many layers (set_bucket_layers)
        self._saved = None
        self._pre_is_grad = False

    def _gemm(self, a, w, M, out0, alg_k=None, **kw):
        """One NT GEMM launch.  alg_k: the ALGORITHMIC contraction length when the operand carries padding columns (fc2 with the
        K-concatenated GPA up-projection: 3072 + 20 of 3136 columns are products the reference computes, the split-bf16 copies and the
        zero padding are not) -- only the bench instrumentation reads it, for flop_per_launch."""
        if GEMM_MARKS is None or not self._recording:
            return ops.gemm_nt(a, w, M, out0, **kw)
        N, K = w.shape[0], int(kw.get("K") or w.shape[1])
        ka = int(alg_k) if alg_k is not None else K
        key = f"gemm_nt_{'f32' if self.fp32 else 'bf16'}[{_EPI_NAMES[kw['epilogue']]}] M={M} N={N} K={K}"
        mrows = M
        if kw.get("m_panels"):                               # strided row panels: 64-row tiles at the first rows of every sample
            mrows = 64 * kw["m_panels"]
            key += f" panels={kw['m_panels']}x64"
        cur = torch.cuda.current_stream()
        e0 = self._ev_record(cur)
        ops.gemm_nt(a, w, M, out0, **kw)
        e1 = self._ev_record(cur)
        self._gemm_marks.append((key, 2.0 * mrows * N * ka, [mrows, N, K], None, e0, e1))

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

    def _fold_deps(self) -> List[str]:
        """Every tensor the folded qkv operands of layers 1.. are built from: the projection weight AND the LayerNorm affine."""
        n = []
        for i in range(1, self.depth):
            a = self.names.attn(i)
            n += [self.names.qkv_weight(i), a + ".norm.weight", a + ".norm.bias"]
        return n

    def _ensure_fold(self) -> bool:
        """Folded operands (gamma o W in bf16, c1 = its row sums, c2 = beta . W^T) of every layer > 0, rebuilt IN PLACE (recorded plans keep
        their pointers) whenever one of their sources changed -- whichever tensors train.  Called by every forward that may take the
        folded path (the ones that keep no GEMM inputs: a frozen backbone, or any eval / no-grad forward), so the path never meets a
        missing or stale operand (a Gaviko(freeze_vit=False) eval forward at ViT-B used to raise KeyError 'w1' here)."""
        if not self._fold_ln1:
            return False
        names = self._fold_deps()
        version = tuple(self.p[n]._version for n in names) + tuple(self.p[n].data_ptr() for n in names)
        if version == self._fold_version and self._fold:
            return True
        for i in range(1, self.depth):
            a = self.names.attn(i)
            Wq, g, b = self._d(self.names.qkv_weight(i)), self._d(a + ".norm.weight"), self._d(a + ".norm.bias")
            w16 = ops.to_operand((Wq * g[None, :]).contiguous(), self._fold.get(f"w{i}"), self.adt)
            self._fold[f"w{i}"] = w16
            for key, val in ((f"c1_{i}", w16.float().sum(1)),                # row sums of the bf16 operand the MFMAs actually multiply
                             (f"c2_{i}", (Wq * b[None, :]).sum(1))):          # beta . W^T
                if key in self._fold:
                    self._fold[key].copy_(val)
                else:
                    self._fold[key] = val.contiguous()
        self._fold_version = version
        return True

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
        if names is None or any(n in self._fold_names for n in names):
            self._fold_version = None

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
        if self._fold_ln1:
            ws["xg16"] = z(M, C, bf16)                                          # bf16 copy of the global stream entering a folded layer
            ws["spart"] = torch.zeros((C // 64) * M * 2, device=device)         # per-row (sum, sum of squares) over 64-column groups
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
            if self._dy16:
                ws["dx16b"] = z(M, C, bf16)
            ws["dctx"] = z(M, C, bf16)
            ws["dqkv"] = z(M, 3 * C, bf16)
            ws["delta"] = torch.zeros((B, self.heads, T), device=device)
            if not self.fp32 and _ATTN_FUSED:                                     # one-pass attention backward: progress words + running dQ sums
                ws["attn_ws"] = ops.attention_bwd_workspace(B, T, self.heads, device)
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
                if self._pgrad:                                                          # one scratch + ticket block per stream (gvk_param_grads)
                    nct = (C + 63) // 64
                    ws["pscratch"] = mk(ops.param_grads_scratch_elems(Lt, [nct, nct], [ng, Lt * Lt, Lt, Lt * Lt, Lt, Lt]))
                    ws["pscratch_l"] = mk(ops.param_grads_scratch_elems(Lt, [nct, nct, (3 * Lt + 63) // 64], []))   # third job: the qkv weight, 3 Lat columns (two tiles from Lat = 24)
                    ws["ptick"] = torch.zeros(ops.PGRAD_TICKETS, dtype=torch.int32, device=device)
                    ws["ptick_l"] = torch.zeros(ops.PGRAD_TICKETS, dtype=torch.int32, device=device)
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
            prio = int(L.diag_env(f"GAVIKO_HIP_{name.upper()}_PRIORITY", SIDE_STREAM_PRIORITY))     # per-stream A/B switch (diag)
            st = self._streams[name] = _SIDE_STREAMS[key] = torch.cuda.Stream(priority=prio)
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
        """Run `fn` eagerly the first GRAPH_WARMUP times, then record it into a launch plan (csrc/runtime.hip) and replay that.
        Everything `fn` launches reads/writes workspace buffers only, so a replay is exactly one more step."""
        k = (tag,) + key + (torch.cuda.current_stream().cuda_stream,)
        g = self._graphs.get(k)
        self._last_run = ("eager", None)
        if g is not None:
            L.check(L.load().gvk_plan_replay(g), "gvk_plan_replay")
            self._last_run = ("replayed", g)
            return
        n = self._calls.get(k, 0)
        self._calls[k] = n + 1
        if not USE_GRAPHS or n < GRAPH_WARMUP:
            fn()
            return
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
        if (sv["bdrop"] > 0 or sv["edrop"] > 0 or sv["pdrop"] > 0) and self.kind not in ("vit", "melo", "vpt", "adaptformer", "gaviko", "dvpt", "evp", "ssf"):
            raise L.GavikoHipError(f"backbone dropout > 0 in training mode: unknown kind {self.kind!r}")
        self.refresh_weights(need_dgrad=train)
        ws = self.workspace(B, img.device, train)
        if img.data_ptr() != ws["img"].data_ptr():           # a caller that fills input_buffer() itself skips the copy-in launch
            ws["img"].copy_(img.detach())                   # static input buffer (the only per-step host-visible copy-in)
        # unfrozen backbone tensors (`fft` / `bitfit`, train.py:123-137): which ones train, and whether GEMM inputs must be kept
        # (`fft` / `bitfit` of the plain ViT; AdaptFormer(freeze_vit=False): adapters keep their own kernels, everything else is backbone)
        # Gaviko(freeze_vit=False): prompts / MWSA / GPA keep their own kernels, everything else is backbone)
        bb = frozenset()
        if train and self.kind in _BB_KINDS:
            bb = frozenset(n for n in self.trainable_names() if not n.startswith(self.names.head()) and not self._own_grad_kernels(n))
        sv["bb"] = bb
        sv["wgrad"] = any(self.p[n].dim() >= 2 and n.endswith("weight") for n in bb)
        if bb:
            self._bb_buffers(ws, B, img.device, sv["wgrad"])
        key = (B, train, sv["attn_drop"], sv["proj_drop"], len(bb), sv["wgrad"], sv["bdrop"], sv["edrop"], sv["pdrop"])
        self._keep_inputs = bool(sv["wgrad"])
        sv["pre_is_grad"] = self._pre_is_grad = bool(train and self._gelu_grad and not self._keep_inputs and sv["bdrop"] <= 0 and not bb)
        self._fold_on = (not self._keep_inputs) and self._ensure_fold()      # ONE place decides: operands are current whenever the fold is taken
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
        self._mark("f:begin")
        ops.seed_advance(ws["seed"], 7919)                  # device-side dropout epoch (replay safe)
        if self.kind == "ssf":
            self._ssf_fold(train)
        # ---- embedding: patch GEMM (+bias +pos, scattered to rows row_off..) and the broadcast rows
        marking = GEMM_MARKS is not None and self._recording  # bench.py: time the whole patch-embed stage (im2col + GEMM + scatter)
        cur = torch.cuda.current_stream()
        pe0 = self._ev_record(cur) if marking else None
        pos = d(nm.root + "pos_embedding")[0]
        G0 = ws["G"][0]
        # im2col -> bf16, then the MFMA GEMM scatters the token rows (a fused gather-GEMM was built in round 2 and measured slower: DESIGN.md 7b.5)
        ops.patchify(ws["img"], ws["cols"], self.patch)
        if self.kind == "evp":
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
            if self.kind == "gaviko":                         # ... and, with its own draw, over the local tokens (gaviko.py:544,548)
                ops.dropout_rows(ws["Lc"][0], sv["edrop"], SEED_EMB + 1, ws["seed"], out32=ws["Lc"][0], M=B * N, N=C)
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
            # The trainable half of W'_fc2 (gvk_pack_split_bf16) is written on the GPA stream, layer i+1's in the idle time behind layer i's GPA:
            # all twelve at the head of the step made layer 0's GPA -- and with it the main stream's prompt fix -- wait for them.
            def side_weights(i):
                if fuse_up and "noside" not in _ABLATE and i + 1 < self.depth:
                    with torch.cuda.stream(gpa):
                        pre, _ = self._gpa_names(i + 1)
                        ops.pack_split_bf16(d(pre + ".proj_up.weight"), self._w16[f"fc2{i + 1}"], self.mlp, C, b=d(pre + ".proj_up.bias"), weight_side=True)
            side_weights(-1)                                         # (layer 0's operand)
        pending_fix = None
        folded_in = False                                            # this layer's first LayerNorm rides its qkv GEMM (self._fold_ln1)
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
            self._attn_block_fwd(ws, i, si, ws["G"][gi], ws["G1"][si], Mi, sv["bdrop"], fix=pending_fix if gaviko and _on("noside") else None,
                                 folded=folded_in)
            pending_fix = None
            self._mark(f"f{i}:attn")
            fused = gaviko and self._fuse_proj
            if gaviko and not fused:
                self._wait("gpa", None)                              # G1 ready
                self._wait("gpa", "loc")                             # L' ready
                with torch.cuda.stream(gpa):
                    self._gpa_fwd_latents(ws, i, si, ws["G1"][si], ws["Lc"][go], M, B, True)
                side_weights(i)
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
                side_weights(i)
            up_in_fc2 = gaviko and fused and fuse_up
            fold_next = bool(up_in_fc2 and self._fold_on and i + 1 < self.depth and _on("noside") and not _FIX_IN_LN)
            # fc2 carries proj_up of the PLAIN latents of every row (ready right behind the LayerNorm); the GPA has the two GEMMs' time
            # to finish, and only the P prompt rows it replaces are fixed up afterwards
            # the last layer's output is read at the pooled rows only (frozen backbone, no dropout behind fc2): its MLP runs on those rows
            last_rows = (self._panels(B, self._pool_rows()[0] + self._pool_rows()[1])
                         if (gaviko and i + 1 == self.depth and up_in_fc2 and not self._keep_inputs and sv["bdrop"] <= 0) else None)
            self._mlp_block_fwd(ws, i, si, ws["G1"][si], gout, Mi, train, sv["bdrop"],
                                up_in_fc2=up_in_fc2, stats_out=fold_next, panels=last_rows)
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
                if fold_next:
                    self._wait(None, "gpa")                          # enh ready
                    sn = ws["stat"][si + 1 if train else 0]
                    ops.prompt_up_fix_stats(pending_fix["enh"], pending_fix["lat"], pending_fix["wup"], ws["G"][go], ws["xg16"], ws["spart"],
                                            sn[0], sn[1], B, self.T, self.P, C, self.Lat, pivot=ws["stat"][si][2])
                    pending_fix = None
                elif i + 1 == self.depth or not _FIX_IN_LN:
                    self._wait(None, "gpa")                          # enh ready
                    if _on("noside"):
                        ops.prompt_up_fix(pending_fix["enh"], pending_fix["lat"], pending_fix["wup"], ws["G"][go], B, self.T, self.P, C, self.Lat)
                    pending_fix = None
                folded_in = fold_next
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
        self._mark("f:head")

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

    def _attn_block_fwd(self, ws, i, si, gin, g1, M, pdrop=0.0, fix=None, folded=False):
        nm, w, d, C = self.names, self._w16, self._d, self.C
        a = nm.attn(i)
        st = ws["stat"][si]
        if folded:
            # LayerNorm folded into the projection: A = the raw rows (bf16 copy left by the fc2 GEMM below + the prompt fix), W = gamma o W,
            # mean / rstd of the rows already in st[0], st[1] (gvk_prompt_up_fix_stats)
            fo = self._fold
            self._gemm(ws["xg16"], fo[f"w{i}"], M, ws["qkv"][si], epilogue=ops.EPI_STORE_BF16, bias=fo[f"c2_{i}"], ln_mean=st[0], ln_rstd=st[1],
                       ln_c1=fo[f"c1_{i}"], scale_cols=self.heads * 64, col_scale=self.q_scale)
            ops.attention_fwd(ws["qkv"][si], ws["ctx"][si], ws["lse"][si], ws["B"], self.Ts[i], self.heads, 64 ** -0.5,
                              drop_p=pdrop, seed=SEED_LAYER + 8 * i, seed_ptr=ws["seed"], q_prescaled=True)
            self._gemm(ws["ctx"][si], w[f"out{i}"], M, g1, epilogue=ops.EPI_BIAS_RES_F32, bias=d(a + ".to_out.0.bias"), res=gin,
                       drop_p=pdrop, seed=SEED_LAYER + 8 * i + 1, seed_ptr=ws["seed"])
            return
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

    def _panels(self, B, rows):
        """gemm_nt kwargs that restrict a launch to the first `rows` rows of every sample (64-row tiles at stride T), or {}."""
        if not self.prune_dead_rows or rows > 64 or self.T < 64:
            return {}
        kw = dict(m_panels=B, m_stride=self.T)
        if _PANEL_SPLITK:
            ws = self._ws
            if "gemm_ws" not in ws:                          # 256 ticket words + up to 256 partial tiles of 32 KiB
                ws["gemm_ws"] = torch.zeros((1024 + 256 * 32768) // 4, dtype=torch.int32, device=ws["logits"].device)
            kw["splitk_ws"] = ws["gemm_ws"]
        return kw

    def _mlp_block_fwd(self, ws, i, si, g1, gout, M, train, pdrop=0.0, up_in_fc2=False, stats_out=False, panels=None):
        nm, w, d, C = self.names, self._w16, self._d, self.C
        m = nm.mlp(i)
        pk = panels or {}
        if self._keep_inputs:
            ops.copy_(ws["sav"]["xn2"][si], ws["xn"])
        gg = bool(train and self._pre_is_grad)               # decided once per step in forward(); the backward reads ws["pre"] accordingly
        self._gemm(ws["xn"], w[f"fc1{i}"], M, ws["pre"][si] if train else None, epilogue=ops.EPI_BIAS_GELU_BF16, out1=ws["act"],
                    bias=d(m + ".net.1.bias"),           # inference keeps no pre-activation (out0 = NULL)
                    ldo=self.ldx, drop_p=pdrop, seed=SEED_LAYER + 8 * i + 2, seed_ptr=ws["seed"], aux_is_grad=int(gg), **pk)
        if self._keep_inputs:
            ops.copy_(ws["sav"]["act"][si], ws["act"])
        so = (dict(epilogue=ops.EPI_BIAS_RES_F32_BF16, out1=ws["xg16"], stat_part=ws["spart"], stat_pivot=ws["stat"][si][2]) if stats_out     # pivot = mean of the residual row (LN2)
              else dict(epilogue=ops.EPI_BIAS_RES_F32))
        self._gemm(ws["act"], w[f"fc2{i}"], M, gout, bias=d(m + ".net.4.bias"), res=g1,
                   K=self.ldx if up_in_fc2 else self.mlp,      # the GPA latents ride this GEMM as 64 extra K columns (self._fuse_up)
                   alg_k=self.mlp + self.Lat if up_in_fc2 else None,
                   drop_p=pdrop, seed=SEED_LAYER + 8 * i + 3, seed_ptr=ws["seed"], **so, **pk)

    # ------------------------------------------------------------------ backward
    def trainable_names(self) -> List[str]:
        return [k for k, p in self.p.items() if p.requires_grad]

    def flat_names(self) -> List[str]:
        """Trainable tensors in the order of the flat gradient buffer: grouped by the bucket of the backward sweep that completes them
        (distributed.flat_order), so that what the data-parallel reducer sends together is one contiguous slice."""
        from .distributed import flat_order
        key = (tuple(self.trainable_names()), self.bucket_layers)
        if self._flat_names is None or self._flat_names[0] != key:
            self._flat_names = (key, flat_order(key[0], self.cfg.get("share_factor", 1) if self.kind == "gaviko" else 1, self.bucket_layers))
        return self._flat_names[1]

    def set_bucket_layers(self, k: int) -> None:
        """Layers per gradient bucket of the flat layout (model.make_reducer passes its own).  Changing it moves gradient views: recorded
        plans hold their addresses, so they are dropped."""
        k = int(k)
        if k != self.bucket_layers:
            self.bucket_layers = k
            self._flat_grad = None
            self._graphs.clear()
            self._calls.clear()

    def set_prune(self, on: bool) -> None:
        """Dead-row pruning on / off (see prune_dead_rows in __init__).  Recorded plans hold the launch lists, so they are dropped."""
        on = bool(on) and self.kind == "gaviko" and not self.fp32
        if on != self.prune_dead_rows:
            self.prune_dead_rows = on
            self._graphs.clear()
            self._calls.clear()

    def _grad_views(self, device) -> Dict[str, torch.Tensor]:
        names = self.flat_names()
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
        if reducer is not None and getattr(reducer, "mode", "segments") == "events":
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
        nw, nb = nm.root + "transformer.norm.weight", nm.root + "transformer.norm.bias"
        if backbone_bwd and (nw in bb or nb in bb):
            g, bw, Tl = self._final_stream(ws, True), ws["bbw"], self.Ts[-1]
            ops.layernorm_fwd(g, d(nw), d(nb), B * Tl, C, y16=ws["xn"], mean=bw["stat"][0], rstd=bw["stat"][1])
            ops.ssf_head_grad(g, bw["stat"][0], bw["stat"][1], d(nm.head() + ".weight"), ws["dlogits"], bw["ones"][:C], bw["zeros"][:C],
                              gv[nw] if nw in bb else bw["junk"][:C], gv[nb] if nb in bb else bw["junk"][C: 2 * C], B, Tl, C, self.K, r0, R)
        if self.kind == "ssf" and backbone_bwd:
            # final norm + ssf (ssf.py:138): statistics of the final stream, then the pooled rows' scale / shift gradients
            g = self._final_stream(ws, True)
            st = ws["ssf_stat"]
            ops.layernorm_fwd(g, self.p["transformer.norm.weight"].detach(), self.p["transformer.norm.bias"].detach(), B * T, C, y16=ws["xn"],
                              mean=st[0], rstd=st[1])
            ops.ssf_head_grad(g, st[0], st[1], d(nm.head() + ".weight"), ws["dlogits"], self.p["transformer.norm.weight"].detach(),
                              self.p["transformer.norm.bias"].detach(), gv["transformer.ssf_scale_1"], gv["transformer.ssf_shift_1"],
                              B, T, C, self.K, r0, R)
            self._ssf_unfold(gv, bb, self._ssf_sites()[-1:])                           # the final norm's affine
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
            self._mark("b:begin")
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
                    self._gpa_bwd_core(ws, sv, gv, i, dGout, M, B, par, project=not (self._proj_bwd_main and i < self.depth - 1))
                    dz_ready = self._ev_record(gpa)
                    self._gpa_bwd_params(ws, sv, gv, i, dGout, M, B, par)
                    self._bucket_mark("gpa", i)                              # prompt_projs.{i // share} gradients final once the lowest layer using it is done
            # main stream, MLP block: dG1 = dGout + LN'(fc1^T(GELU'(pre) * fc2^T(dGout)))
            self._mark(f"b{i}:start")
            bb = sv.get("bb") or ()
            pd_ = sv.get("bdrop", 0.0)
            dy_ff = dGout
            if pd_ > 0:                                                      # gradient of dropout(fc2(.)): the forward's mask on dGout
                dy_ff = self._masked_grad(ws, dGout, pd_, SEED_LAYER + 8 * i + 3, bool(bb) or self.kind == "ssf", M)
            if bb:                                                           # fc2: db = colsum(dGout), dW = dGout^T . act
                self._bb_linear_grads(ws, gv, bb, m + ".net.4", dy_ff, ws["dG16"], ws["sav"]["act"][i] if sv["wgrad"] else None, M, C, self.mlp,
                                      ldx=self.ldx if self.ldx != self.mlp else None)
            dvpt = self.kind == "dvpt"
            if dvpt:
                self._dvpt_bwd_latents(ws, gv, i, dGout, M, B)
            ssf = self.kind == "ssf"
            if ssf:                                                          # fc2 + ssf_2: dy = dGout, y = G[i+1] - G1[i]
                # (behind a live dropout the stored difference is kept / (1 - p): masked gradient, y_mul = 1 - p)
                self._ssf_linear_grad(ws, gv, m, 2, dy_ff, ws["G"][i + 1], M, C, y1=ws["G1"][i], y_mul=1.0 - pd_)
            # top layer, frozen backbone: the incoming gradient is an exact zero outside the rows the head pools -- the MLP's backward (row-wise)
            # runs on those rows; the rest of dG1 is zeroed in one memset instead of being computed as LN'(0) + 0
            top = (self._panels(B, self._pool_rows()[0] + self._pool_rows()[1])
                   if (gaviko and first and i == self.depth - 1 and not bb and pd_ <= 0 and not sv["wgrad"]) else {})
            self._gemm(ws["dG16"], w[f"fc2{i}_t"], M, ws["dpre"], epilogue=ops.EPI_GELU_BWD_BF16, aux=ws["pre"][i], ldaux=self.ldx,
                       drop_p=pd_, seed=SEED_LAYER + 8 * i + 2, seed_ptr=ws["seed"], aux_is_grad=int(sv.get("pre_is_grad", False)), **top)
            if ssf:                                                          # fc1 + ssf_1: dy = d(pre-activation), y = saved pre-activation
                self._ssf_linear_grad(ws, gv, m, 1, ws["dpre"], ws["pre"][i], M, self.mlp)
            if bb:                                                           # fc1: db = colsum(dpre), dW = dpre^T . LN2(G1)
                self._bb_linear_grads(ws, gv, bb, m + ".net.1", ws["dpre"], ws["dpre"], ws["sav"]["xn2"][i] if sv["wgrad"] else None, M, self.mlp, C)
            self._mark(f"b{i}:fc2d") if False else None
            fuse_scatter = (gaviko and self._fuse_local and not self.fp32 and "noside" not in _ABLATE
                            and L.diag_env("GAVIKO_HIP_FUSE_SCATTER", "0") == "1")     # measured: 651 vs 676 volumes/s -- off
            dy16 = bool(self._dy16 and not bb and not sv["wgrad"] and not fuse_scatter)
            if dy16:
                self._gemm(ws["dpre"], w[f"fc1{i}_t"], M, ws["dx16b"], epilogue=ops.EPI_STORE_BF16, **top)
            else:
                self._gemm(ws["dpre"], w[f"fc1{i}_t"], M, ws["dx32"], epilogue=ops.EPI_STORE_F32, **top)
            self._mark(f"b{i}:fc1d")
            if bb:
                self._bb_ln_grads(ws, gv, bb, m + ".net.0", ws["dx32"], ws["G1"][i], st[2], st[3], M)
            if ssf:                                                          # LN2 + ssf_0
                self._ssf_ln_grad(ws, gv, m, ".net.0", ws["dx32"], ws["G1"][i], st[2], st[3], M)
            adapter = self.kind == "adaptformer"
            if fuse_scatter:
                # dG1 = dGout + LN'(dx32) + dzx . W_d (+ the bf16 operand of the out-proj dgrad) in ONE pass: the GPA core of this layer
                # (started at the top of the layer on its own stream) has long finished when the two MLP dgrad GEMMs are through
                self._ev_wait(torch.cuda.current_stream(), dz_ready)
                gpre, _ = self._gpa_names(i)
                ops.layernorm_bwd_up(ws["dx32"], ws["G1"][i], st[2], st[3], d(m + ".net.0.weight"), M, C, dx=dGin, dres=dGout, dx16=ws["dG16"],
                                     lat=ws["bw"]["dzx"], w=d(gpre + ".proj_down.0.weight"), L_=self.Lat, w_layout=1)
            elif top:
                ops.memset_zero(dGin)
                if dy16:
                    ops.layernorm_bwd_dy16(ws["dx16b"], ws["G1"][i], st[2], st[3], d(m + ".net.0.weight"), M, C, dx=dGin, dres=dGout,
                                           rows=(B, self._pool_rows()[0] + self._pool_rows()[1], T))
                else:
                    ops.layernorm_bwd_rows(ws["dx32"], ws["G1"][i], st[2], st[3], d(m + ".net.0.weight"), B, self._pool_rows()[0] + self._pool_rows()[1],
                                           T, C, dx=dGin, dres=dGout)
            elif dy16:
                ops.layernorm_bwd_dy16(ws["dx16b"], ws["G1"][i], st[2], st[3], d(m + ".net.0.weight"), M, C, dx=dGin, dres=dGout)
            else:
                ops.layernorm_bwd(ws["dx32"], ws["G1"][i], st[2], st[3], d(m + ".net.0.weight"), M, C, dx=dGin, dres=dGout,
                                  dx16=None if (gaviko or adapter or dvpt) else ws["dG16"])
            if dvpt:
                self._dvpt_bwd_scatter(ws, i, dGin, M)                       # dG1 += (dz . Wd) * QuickGELU'(G1)  (+ operand copy)
            if adapter:
                self._adapter_bwd(ws, gv, i, dGout, dGin, M, refresh_operand=pd_ > 0)   # adds LN_a'(...) into dG1 and refreshes dG16
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
            dy_at = dGin
            if pd_ > 0:                                                      # gradient of dropout(to_out(.))
                dy_at = self._masked_grad(ws, dGin, pd_, SEED_LAYER + 8 * i + 1, bool(bb) or ssf, M)
            if ssf:                                                          # to_out + ssf_2: dy = dG1, y = G1[i] - G[i]
                self._ssf_linear_grad(ws, gv, a, 2, dy_at, ws["G1"][i], M, C, y1=ws["G"][i], y_mul=1.0 - pd_)
            if bb:                                                           # to_out: db = colsum(dG1), dW = dG1^T . ctx
                self._bb_linear_grads(ws, gv, bb, a + ".to_out.0", dy_at, ws["dG16"], ws["ctx"][i], M, C, C)
            self._gemm(ws["dG16"], w[f"out{i}_t"], M, ws["dctx"], epilogue=ops.EPI_STORE_BF16)
            self._mark(f"b{i}:outd")
            # (bottom layer of a frozen backbone: only dq / dk / dv of the prompt rows are read -- by the row-panel qkv dgrad below)
            rows0 = self.P if (gaviko and last and i == 0 and not bb and pd_ <= 0 and not sv["wgrad"] and self._panels(B, self.P)) else None
            ops.attention_bwd(ws["qkv"][i], ws["ctx"][i], ws["dctx"], ws["lse"][i], ws["delta"], ws["dqkv"], B, T, self.heads, 64 ** -0.5,
                              drop_p=pd_, seed=SEED_LAYER + 8 * i, seed_ptr=ws["seed"], q_prescaled=True,
                              ws=ws.get("attn_ws") if (_ATTN_FUSED and rows0 is None) else None, need_rows=rows0)
            self._mark(f"b{i}:attnb")
            if gaviko and shift:
                self._mwsa_chain_bwd(ws, sv, gv, i, par, B, loc, self._ev_record(torch.cuda.current_stream()))
            if self.kind == "melo" and i in self.lora_layers:
                self._melo_bwd(ws, gv, i, M)
            if ssf:                                                          # to_qkv + ssf_1: dy = dqkv, y = saved qkv
                uq = {} if self.fp32 else dict(y0_cols=self.heads * 64, y0_mul=1.0 / self.q_scale)     # the saved q block is pre-scaled
                self._ssf_linear_grad(ws, gv, a, 1, ws["dqkv"], ws["qkv"][i], M, 3 * C, **uq)
            if bb:                                                           # to_qkv (bias-free): dW = dqkv^T . LN1(G)
                self._bb_linear_grads(ws, gv, bb, a + ".to_qkv", None, ws["dqkv"], ws["sav"]["xn1"][i] if sv["wgrad"] else None, M, 3 * C, C)
            # bottom layer, frozen backbone: of this layer's INPUT gradient only the P prompt rows of every sample are read (prompt_embeddings and
            # their position embedding; patch embedding, cls token and pos_embedding carry none) -- qkv dgrad and LayerNorm 1 on those rows
            bot = (self._panels(B, self.P) if (gaviko and last and i == 0 and not bb and pd_ <= 0 and not sv["wgrad"] and self.P > 0) else {})
            if dy16:
                self._gemm(ws["dqkv"], w[f"qkv{i}_t"], M, ws["dx16b"], epilogue=ops.EPI_STORE_BF16, **bot)
            else:
                self._gemm(ws["dqkv"], w[f"qkv{i}_t"], M, ws["dx32"], epilogue=ops.EPI_STORE_F32, **bot)
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
                if dy16:
                    g1 = d(a + ".norm.weight")
                    if self._proj_bwd_main and i > 0:
                        pre_lo, _ = self._gpa_names(i - 1)
                        ops.layernorm_bwd_dy16(ws["dx16b"], ws["G"][i], st[0], st[1], g1, M, C, dx=dGnext, dres=dGin, dx16=ws["dG16"],
                                               proj=dict(w=d(pre_lo + ".proj_up.weight"), y=ws["bw"]["dcomb"], w_layout=1, L_=self.Lat))
                    elif bot:
                        ops.layernorm_bwd_dy16(ws["dx16b"], ws["G"][i], st[0], st[1], g1, M, C, dx=dGnext, dres=dGin, rows=(B, self.P, T))
                    else:
                        ops.layernorm_bwd_dy16(ws["dx16b"], ws["G"][i], st[0], st[1], g1, M, C, dx=dGnext, dres=dGin, dx16=ws["dG16"])
                elif self._proj_bwd_main and i > 0:
                    pre_lo, _ = self._gpa_names(i - 1)
                    ops.layernorm_bwd_proj(ws["dx32"], ws["G"][i], st[0], st[1], d(a + ".norm.weight"), M, C, dx=dGnext, dres=dGin,
                                           dx16=ws["dG16"], w=d(pre_lo + ".proj_up.weight"), y=ws["bw"]["dcomb"], w_layout=1, L_=self.Lat)
                elif bot:
                    ops.layernorm_bwd_rows(ws["dx32"], ws["G"][i], st[0], st[1], d(a + ".norm.weight"), B, self.P, T, C, dx=dGnext, dres=dGin)
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
            if ssf and bb:
                self._ssf_unfold(gv, bb, self._ssf_sites()[1 + 6 * i: 7 + 6 * i])     # this layer's six sites, before its bucket is final
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
                # frozen embedding: the local stream's INPUT gradient (layer 0's deferred last step) has no reader -- conv / pos_embedding carry none
                self._mwsa_flush(ws, B, loc, dead=self.prune_dead_rows and lo == 0 and not sv.get("bb"))
            self._wait(None, "gpa")
            self._wait(None, "loc")
        if last and sv.get("bb"):
            if sv.get("edrop", 0.0) > 0:                                     # through emb_dropout
                ops.dropout_rows(dGout, sv["edrop"], SEED_EMB, ws["seed"], out32=dGout, M=B * self.T, N=C)
            dlocal = ws["dL"][(self.depth - 1 - lo + 1) & 1] if gaviko else None      # what _mwsa_final of the lowest layer wrote
            if gaviko and sv.get("edrop", 0.0) > 0:
                ops.dropout_rows(dlocal, sv["edrop"], SEED_EMB + 1, ws["seed"], out32=dlocal, M=B * self.N, N=C)
            if self.kind == "evp":
                # the raw conv output also feeds embedding_generator (evp.py:347-348): d xc += d s . W_e  (d s is complete once layer 0 is through)
                dlocal = ws["ev"]["tmp"]
                ops.skinny_up(lat=ws["evb"]["ds"], w=self._evp_state(dGout.device)["We"], out=dlocal, M=B * self.N, C=C, L=self.Lp, w_layout=1,
                              accumulate=0)
            self._bb_embed_grads(ws, gv, sv["bb"], dGout, B, dlocal=dlocal, dlocal_to_pos=gaviko)
        if last and self.kind == "evp":
            self._evp_bwd_finish(ws, gv, B)
        if last and self.kind == "ssf":
            # patch embedding + ssf (ssf.py:229-232): dy = the patch rows of the input gradient, y = G[0] patch rows - pos[1:]
            pos = self.p["pos_embedding"].detach()[0]
            ops.ssf_colgrad(dGout, ws["G"][0], self.p["ssf_scale_1"].detach(), self.p["ssf_shift_1"].detach(), gv["ssf_scale_1"], gv["ssf_shift_1"],
                            ws["ssf_scratch"], B * self.N, C, pos=pos[1:], rows_in=self.N, rows_out=T, row_off=1,
                            y_mul=1.0 - sv.get("edrop", 0.0))              # (dGout is already masked by emb_dropout when that is live)
            self._ssf_unfold(gv, sv.get("bb") or (), self._ssf_sites()[:1])            # the patch embedding's conv tensors
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
        self._mark("b:tail")                                                 # side streams joined, embedding-side gradients issued

    def _grad_supported(self, name: str) -> bool:
        # head: always; backbone tensors: the classes of _BB_KINDS (plain ViT `linear` / `bitfit` / `fft`, AdaptFormer and Gaviko with
        # freeze_vit=False); everything else: the method's own tensors
        return name.startswith(self.names.head()) or self.kind in _BB_KINDS or self._own_grad_kernels(name)

    def _own_grad_kernels(self, name: str) -> bool:
        """The method's OWN trainable tensors (prompts, adapters, LoRA factors, ...): gradients from the method-specific kernels."""
        if self.kind == "vit":
            return False
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
