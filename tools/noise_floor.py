#!/usr/bin/env python3
"""CPU (any machine): the error the ORACLE ITSELF shows when the operands the HIP path hands to bf16 MFMAs are rounded to bf16
(oracle.vit_ref.BF16_OPERANDS: inputs + weights of the four backbone Linears, q / k / v, attention probabilities, patch columns; fp32
accumulation, fp32 residual stream / LayerNorm / softmax -- as in the kernels).  fp32 oracle vs that mode = the noise floor a bf16
tolerance has to absorb; the straight-through rounding leaves the backward's own operand rounding out, so the gradient figures are
LOWER bounds of the floor.

    python tools/noise_floor.py [case ...] > profiles/r02_bf16_noise_floor.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import vit_ref  # noqa: E402
from gaviko_amd.utils import synth  # noqa: E402

BASE = dict(image_size=160, image_patch_size=16, frames=120, frame_patch_size=12, num_classes=5, channels=1, pool="cls", dim_head=64)
GAVIKO = dict(num_prompts=32, prompt_latent_dim=20, local_dim=20, local_k=(6, 6, 6), DHW=(10, 10, 10), attn_drop=0.0, proj_drop=0.0, freeze_vit=True,
              share_factor=1)
VPT = dict(num_prompts=8, prompt_dim=64, prompt_dropout=0.0, freeze_vit=True, deep_prompt=True)
CASES = {**{f"cfg3_deep_vpt_b16_shard{s}": ("deep_vpt", "vit-b16", 4, VPT, 4 * s) for s in range(8)}, "cfg2_gaviko_b16_b4": ("gaviko", "vit-b16", 4, GAVIKO, 0),
         "cfg4_adaptformer_b16_b8": ("adaptformer", "vit-b16", 8, dict(freeze_vit=True), 0), "cfg4_melo_b16_b8": ("melo", "vit-b16", 8, dict(r=4, alpha=4), 0),
         "gaviko_t16_b2": ("gaviko", "vit-t16", 2, GAVIKO, 0), "melo_t16_b2": ("melo", "vit-t16", 2, dict(r=4, alpha=4), 0)}


def run(method, backbone, B, extra, first, bf16):
    cfg = dict(BASE, backbone=backbone, method=method, **extra)
    sd = {k: torch.from_numpy(v).requires_grad_(oracle.trainable(method, k)) for k, v in synth.fill_state_dict(oracle.SHAPES[method](cfg)).items()}
    x, y = torch.from_numpy(synth.volumes(first, B)), torch.from_numpy(synth.labels(first, B))
    vit_ref.BF16_OPERANDS = bf16
    try:
        logits = oracle.FORWARD[method](sd, x, cfg, None)
        torch.nn.functional.cross_entropy(logits, y).backward()
    finally:
        vit_ref.BF16_OPERANDS = False
    return logits.detach().numpy(), {k: v.grad.numpy() for k, v in sd.items() if v.grad is not None}


def main():
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
    names = sys.argv[1:] or list(CASES)
    out = {"note": __doc__.split("\n\n")[0], "cases": {}}
    for n in names:
        method, backbone, B, extra, first = CASES[n]
        l0, g0 = run(method, backbone, B, extra, first, False)
        l1, g1 = run(method, backbone, B, extra, first, True)
        d = np.abs(l1 - l0).max()
        gn = sorted(((abs(np.linalg.norm(g1[k]) - np.linalg.norm(g0[k])) / max(np.linalg.norm(g0[k]), 1e-12), k) for k in g0), reverse=True)
        ge = sorted(((np.abs(g1[k] - g0[k]).max() / max(np.abs(g0[k]).max(), 1e-12), k) for k in g0 if g0[k].size <= 200000), reverse=True)
        e = np.array([v for v, _ in gn])
        out["cases"][n] = {"logits_max_abs": float(d), "logits_rel": float(d / np.abs(l0).max()), "logits_ref_max": float(np.abs(l0).max()),
                           "argmax_equal": bool((l0.argmax(-1) == l1.argmax(-1)).all()),
                           "gradnorm_rel": {"median": float(np.median(e)), "p90": float(np.percentile(e, 90)), "max": float(e.max()), "worst": gn[0][1]},
                           "grad_elementwise_rel_worst": [{"tensor": k, "rel": float(v)} for v, k in ge[:3]]}
        print(n, json.dumps(out["cases"][n]), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
