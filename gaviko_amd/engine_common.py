"""Constants, switches and small helpers shared by the launch-plan engine (engine.py) and its method-specific halves
(engine_gaviko.py: the MWSA / GPA side paths; engine_peft.py: AdaptFormer, LoRA, unfrozen-backbone gradients, EVP, DVPT, SSF)."""
from __future__ import annotations

import os

from . import lib as L

# How a step is issued after the GRAPH_WARMUP eager passes (env GAVIKO_HIP_GRAPHS):
#   "plan"  (default, also "1") -- the library's launch plan: recorded once, replayed from one C loop on the real streams
#   "0" / "eager"               -- every launch from Python
# (a captured hipGraph per pass was the round-1 form; its executor serialises three forked branches on this runtime,
#  tools/probe/probe_streams.hip: 7.6 ms vs 3.9 ms eager -- removed in round 3)
_MODE = os.environ.get("GAVIKO_HIP_GRAPHS", "plan")
STEP_MODE = {"1": "plan", "0": "eager"}.get(_MODE, _MODE)
if STEP_MODE not in ("plan", "eager"):
    raise L.GavikoHipError(f"GAVIKO_HIP_GRAPHS={_MODE!r}: expected 'plan' (default) or 'eager'")
USE_GRAPHS = STEP_MODE != "eager"
GRAPH_WARMUP = 2
PLAN_TIMING = os.environ.get("GAVIKO_HIP_PLAN_TIMING") is not None
# Timing ablations (tools/ablate_streams.py) -- the RESULTS ARE WRONG with either switch; they only answer "where does the step go":
#   nowait: the main stream skips its waits on the side chains;  noside: the MWSA / GPA chains are not launched at all.
# (measurement build only: L.diag_env reads the environment when GAVIKO_HIP_DIAG=1 selects libgaviko_hip_diag.so, else returns the default --
#  in the product every switch below is a constant and no `_on(...)` branch can be taken the wrong way)
_ABLATE = set(filter(None, (L.diag_env("GAVIKO_HIP_ABLATE", "") or "").split(",")))


def _on(tag: str) -> bool:
    """False when the timing ablation `tag` is switched on (GAVIKO_HIP_ABLATE, diagnostics only: bench.py refuses it without --allow-ablate)."""
    return tag not in _ABLATE
# Site seeds of the backbone's own nn.Dropout modules (added to the device epoch word): embedding, VPT prompts of layer i, and per layer
# {+0 attention probabilities, +1 to_out, +2 after GELU, +3 after fc2}.  The MWSA sites use 2*i and 2*i + 1.
SEED_EMB, SEED_PROMPT, SEED_LAYER = 900, 950, 1000

# Dispatch priority of the side-chain streams (negative = higher).  Measured: -1 drops the step rate from 448 to 274 volumes/s
# (priority queues serialise against the captured graph's main queue on this runtime), so the default stays 0.
SIDE_STREAM_PRIORITY = int(L.diag_env("GAVIKO_HIP_SIDE_PRIORITY", "0"))
# GPA prompt fix inside the next layer's first LayerNorm (gvk_layernorm_fwd_fix) instead of its own 128-row launch: measured 709-711 vs
# 719-721 volumes/s -- the 128 prompt rows' waves become the tail of a 4132-row kernel; opt-in
_FIX_IN_LN = L.diag_env("GAVIKO_HIP_FIX_IN_LN", "0") == "1"
_SIDE_STREAMS = {}                       # (device index, kind) -> the process-wide side stream of that kind
# MWSA backward chain held behind the layer's attention backward: measured 669 vs 688 volumes/s -- the chain then slows the dgrad GEMMs
# of the next layer by as much as it slowed the attention kernels before (start->fc1d 82 -> 98 us); opt-in only
_LOC_SHIFT = L.diag_env("GAVIKO_HIP_LOC_SHIFT", "0") == "1"
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

    def __init__(self, kind: str, lora_layers=None):
        self.kind = kind
        self.lora_layers = None if lora_layers is None else frozenset(lora_layers)     # MeLO: layers whose to_qkv is wrapped (melo.py:53-68)
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
        wrapped = self.kind == "melo" and (self.lora_layers is None or i in self.lora_layers)
        return self.attn(i) + (".to_qkv.qkv.weight" if wrapped else ".to_qkv.weight")

    def head(self):
        return "mlp_head.head" if self.kind == "gaviko" else f"{self.root}mlp_head"
