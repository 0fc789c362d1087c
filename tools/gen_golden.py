#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (imported from /root/reference/src, CPU fp32).

Only runs in the build container (the reference never travels).  For each case it
  1. instantiates the reference class (timm stubbed, pretrained-weight fetch bypassed -- SURVEY 8(c)),
  2. overwrites every parameter with the formula recipe of gaviko_amd.utils.synth (so fixtures hold no weights),
  3. runs forward + CrossEntropy backward on synthetic volumes, hooks the per-layer module outputs,
  4. stores logits / losses / grad norms / selected full grads / strided activation samples,
  5. cross-checks the oracle restatement tensor-for-tensor and records the max deviation.

Usage:  python tools/gen_golden.py [case ...]      (no args = all cases)
"""
from __future__ import annotations

import os
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("GAVIKO_REFERENCE", "/root/reference/src")

import torch  # noqa: E402

from gaviko_amd.utils import synth  # noqa: E402

torch.set_num_threads(8)

BASE = dict(image_size=160, image_patch_size=16, frames=120, frame_patch_size=12, num_classes=5, channels=1,
            pool="cls", dim_head=64, dropout=0.0, emb_dropout=0.0)
GAVIKO = dict(num_prompts=32, prompt_latent_dim=20, local_dim=20, local_k=(6, 6, 6), DHW=(10, 10, 10),
              attn_drop=0.0, proj_drop=0.0, freeze_vit=True, share_factor=1, fp16=False)

CASES = {
    # name: (method, backbone, batch, extra cfg)
    "cfg1_linear_t16_b1": ("linear", "vit-t16", 1, {}),
    "gaviko_t16_b2": ("gaviko", "vit-t16", 2, dict(GAVIKO)),
    "gaviko_t16_b2_k366_p8": ("gaviko", "vit-t16", 2, dict(GAVIKO, local_k=(3, 6, 6), num_prompts=8)),
    "gaviko_t16_b1_share2": ("gaviko", "vit-t16", 1, dict(GAVIKO, share_factor=2)),
    "gaviko_t16_b2_unfrozen": ("gaviko", "vit-t16", 2, dict(GAVIKO, freeze_vit=False)),       # gaviko.py:428-434 skipped: the whole backbone trains too
    "gaviko_t16_b2_lat16": ("gaviko", "vit-t16", 2, dict(GAVIKO, prompt_latent_dim=16, local_dim=16)),   # a latent width the L = 20 tile kernels do not cover
    "cfg2_gaviko_b16_b4": ("gaviko", "vit-b16", 4, dict(GAVIKO)),
    "deep_vpt_t16_b2": ("deep_vpt", "vit-t16", 2, dict(num_prompts=8, prompt_dim=64, prompt_dropout=0.0, freeze_vit=True, deep_prompt=True)),
    "deep_vpt_t16_b2_unfrozen": ("deep_vpt", "vit-t16", 2, dict(num_prompts=8, prompt_dim=64, prompt_dropout=0.0, freeze_vit=False, deep_prompt=True)),
    "shallow_vpt_t16_b2_unfrozen": ("shallow_vpt", "vit-t16", 2, dict(num_prompts=8, prompt_dim=64, prompt_dropout=0.0, freeze_vit=False, deep_prompt=False)),
    "shallow_vpt_t16_b2": ("shallow_vpt", "vit-t16", 2, dict(num_prompts=8, prompt_dim=64, prompt_dropout=0.0, freeze_vit=True, deep_prompt=False)),
    "cfg3_deep_vpt_b16_8x4": ("deep_vpt", "vit-b16", 4, dict(num_prompts=8, prompt_dim=64, prompt_dropout=0.0, freeze_vit=True, deep_prompt=True, shards=8)),
    "adaptformer_t16_b2": ("adaptformer", "vit-t16", 2, dict(freeze_vit=True)),
    "adaptformer_t16_b2_unfrozen": ("adaptformer", "vit-t16", 2, dict(freeze_vit=False)),       # every tensor trains (adaptformer.py:163: the freeze loop is skipped)
    "melo_t16_b2": ("melo", "vit-t16", 2, dict(r=4, alpha=4)),
    "melo_t16_b2_layers": ("melo", "vit-t16", 2, dict(r=4, alpha=8, lora_layer=[0, 5, 11])),      # LoRA on a subset of layers, integer scale alpha // r = 2
    "cfg4_adaptformer_b16_b8": ("adaptformer", "vit-b16", 8, dict(freeze_vit=True)),
    "cfg4_melo_b16_b8": ("melo", "vit-b16", 8, dict(r=4, alpha=4)),
    "cfg5_gaviko_l16_b2": ("gaviko", "vit-l16", 2, dict(GAVIKO)),
    "ssf_t16_b2": ("ssf", "vit-t16", 2, dict(freeze_vit=True)),
    "ssf_t16_b2_unfrozen": ("ssf", "vit-t16", 2, dict(freeze_vit=False)),        # ssf.py:192 skipped: the backbone trains beside the scales / shifts
    "ssf_b16_b4": ("ssf", "vit-b16", 4, dict(freeze_vit=True)),
    "dvpt_t16_b2": ("dvpt", "vit-t16", 2, dict(num_prompts=50, freeze_vit=True)),
    "dvpt_t16_b2_unfrozen": ("dvpt", "vit-t16", 2, dict(num_prompts=8, freeze_vit=False)),      # dvpt.py:156 skipped: the backbone trains too
    "dvpt_t16_b2_mean_p8": ("dvpt", "vit-t16", 2, dict(num_prompts=8, freeze_vit=True, pool="mean")),
    "dvpt_b16_b4": ("dvpt", "vit-b16", 4, dict(num_prompts=50, freeze_vit=True)),
    "bitfit_t16_b2": ("bitfit", "vit-t16", 2, {}),
    "fft_t16_b2": ("fft", "vit-t16", 2, {}),
    "fft_b16_b2": ("fft", "vit-b16", 2, {}),
    "evp_t16_b2": ("evp", "vit-t16", 2, dict(freeze_vit=True)),
    "evp_t16_b2_unfrozen": ("evp", "vit-t16", 2, dict(freeze_vit=False)),        # evp.py:322 skipped: the backbone trains too
    "evp_b16_b2": ("evp", "vit-b16", 2, dict(freeze_vit=True)),
}


def import_reference():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("timm", types.ModuleType("timm"))
    sys.path.insert(0, REF)
    import utils.load_pretrained as lp
    lp_orig = lp.load_pretrain
    lp.load_pretrain = lambda *a, **k: {}
    import model.vision_transformer as vt
    import model.gaviko as gv
    import model.vpt as vpt
    import model.adaptformer as af
    import model.melo as melo
    import model.ssf as ssf
    import model.dvpt as dvpt
    import model.evp as evp
    import losses.focal_loss as fl
    for m in (vt, gv, af, ssf, dvpt, evp):
        m.load_pretrain = lp.load_pretrain
    return dict(vt=vt, gv=gv, vpt=vpt, af=af, melo=melo, ssf=ssf, dvpt=dvpt, evp=evp, fl=fl, lp=lp, lp_orig=lp_orig)


def build_reference(mods, method, cfg):
    """The factory of train.py:111-153, restated."""
    if method == "gaviko":
        return mods["gv"].Gaviko(**cfg)
    if method in ("linear", "fft", "bitfit"):
        m = mods["vt"].VisionTransformer(**cfg)
        for k, p in m.named_parameters():
            if method == "linear":
                p.requires_grad = "head" in k
            elif method == "bitfit":
                p.requires_grad = ("bias" in k) or ("head" in k)
        return m
    if method in ("deep_vpt", "shallow_vpt"):
        return mods["vpt"].PromptedVisionTransformer(**cfg)
    if method == "adaptformer":
        return mods["af"].AdaptFormer(**cfg)
    if method == "melo":
        return mods["melo"].MeLO(vit=mods["vt"].VisionTransformer(**cfg), **cfg)
    if method == "ssf":
        return mods["ssf"].ScalingShiftingFeatures(**cfg)
    if method == "dvpt":
        return mods["dvpt"].DynamicVisualPromptTuning(**cfg)
    if method == "evp":
        return mods["evp"].ExplicitVisualPrompting(**cfg)
    raise ValueError(method)


def sample_rows(T):
    return sorted(set(r for r in (0, 1, 7, 8, 9, 31, 32, 33, 34, 66, 500, T - 1) if r < T))


def sample_cols(C):
    return list(range(0, C, max(1, C // 32)))[:32]


def tap(t):
    """[B,T,C] -> strided sample, float32 numpy."""
    t = t.detach()
    return t[:, sample_rows(t.shape[1])][:, :, sample_cols(t.shape[2])].float().numpy().copy()


def attach_hooks(model, method, taps):
    """Record residual-stream values around each block from module inputs/outputs."""
    hs = []
    if method == "gaviko":
        tr = model.transformer
        state = {}
        for i in range(tr.depth):
            def h_attn(mod, inp, out, i=i):
                taps[f"layer{i}.post_attn"] = tap(out + inp[0])

            def h_mlp(mod, inp, out, i=i):
                taps[f"layer{i}.post_mlp"] = tap(out + inp[0] + state["prompt"])

            hs.append(tr.attns[i].register_forward_hook(h_attn))
            hs.append(tr.mlps[i].register_forward_hook(h_mlp))
        # shared modules are called several times: keep a running layer counter
        state["li"] = 0

        def h_local_any(mod, inp, out):
            taps[f"layer{state['li']}.local"] = tap(out + inp[0])

        def h_gpa_any(mod, inp, out):
            taps[f"layer{state['li']}.gpa"] = tap(out)
            state["prompt"] = out.detach()
            state["li"] += 1

        for m in tr.local_attns:
            hs.append(m.register_forward_hook(h_local_any))
        for m in tr.prompt_projs:
            hs.append(m.register_forward_hook(h_gpa_any))
        hs.append(tr.register_forward_hook(lambda mod, inp, out: taps.__setitem__("final_norm", tap(out))))
    else:
        vit = model
        if method in ("deep_vpt", "shallow_vpt"):
            vit = model.vision_transformer
        if method == "melo":
            vit = model.lora_vit
        for i, layer in enumerate(vit.transformer.layers):
            ff = layer[-1].mlp if method == "dvpt" else layer[-1]

            def h_ff(mod, inp, out, i=i):
                taps[f"layer{i}.ff_out"] = tap(out)

            hs.append(ff.register_forward_hook(h_ff))
        hs.append(vit.transformer.norm.register_forward_hook(lambda mod, inp, out: taps.__setitem__("final_norm", tap(out))))
    return hs


FULL_GRAD_PATTERNS = (
    "mlp_head", "ssf_s", "cls_token", "transformer.norm.", "conv_proj.0.bias", "layers.0.0.norm", "layers.0.0.to_out.0.bias", "layers.0.0.prompt_proj", "prompt_generator.shared_mlp", "prompt_generator.embedding_generator",
    "prompt_generator.lightweight_mlp_0.", "prompt_generator.prompt_generator.proj.bias", "prompt_embeddings", "prompt_positional_embedding", "prompt_proj.",
    "prompt_projs.0.", "local_attns.0.", "layers.0.1.", "layers.0.0.to_qkv.linear_",
)


def want_full_grad(name, depth):
    last = f".{depth - 1}."
    if any(p in name for p in FULL_GRAD_PATTERNS):
        return True
    return (f"prompt_projs{last}" in name or f"local_attns{last}" in name
            or f"layers{last}1." in name and "adapter" in name or f"layers{last}0.to_qkv.linear_" in name)


def run_case(mods, name, outdir):
    import oracle

    method, backbone, B, extra = CASES[name]
    extra = dict(extra)
    shards = extra.pop("shards", 1)
    cfg = dict(BASE, backbone=backbone, method=method, **extra)
    t0 = time.time()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)  # vpt.py:54-55 appends to ./deep_prompt.txt
        try:
            model = build_reference(mods, method, cfg)
        finally:
            os.chdir(cwd)
    sd = model.state_dict()
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in sd.items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model.train()   # train() overrides keep the frozen backbone in eval (gaviko.py:513-528)
    depth = oracle.mapping_vit(backbone)[0]
    trainable = [k for k, p in model.named_parameters() if p.requires_grad]

    out = {"meta/method": method, "meta/backbone": backbone, "meta/batch": B, "meta/shards": shards,
           "meta/cfg": repr({k: v for k, v in cfg.items()}), "meta/trainable": np.array(trainable)}
    focal = mods["fl"].FocalLoss(gamma=1.2)
    grads_acc = None
    logits_all = []
    for s in range(shards):
        x = torch.from_numpy(synth.volumes(s * B, B))
        y = torch.from_numpy(synth.labels(s * B, B))
        taps = {}
        hooks = attach_hooks(model, method, taps) if s == 0 else []
        model.zero_grad(set_to_none=True)
        logits = model(x)
        for h in hooks:
            h.remove()
        loss = torch.nn.functional.cross_entropy(logits, y)
        loss.backward()
        logits_all.append(logits.detach().numpy().copy())
        g = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.requires_grad}
        grads_acc = g if grads_acc is None else {k: grads_acc[k] + g[k] for k in g}
        if s == 0:
            out["loss_ce"] = np.float32(loss.item())
            with torch.no_grad():
                out["loss_focal"] = np.float32(focal(logits.detach(), y).item())
            for k, v in taps.items():
                out["tap/" + k] = v
            # ---- oracle cross-check on the same weights/inputs (shard 0) ----
            osd = {k: v.detach().clone().requires_grad_(oracle.trainable(method, k, cfg)) for k, v in model.state_dict().items()}
            otaps = {}
            ologits = oracle.FORWARD[method](osd, x, cfg, otaps)
            oloss = torch.nn.functional.cross_entropy(ologits, y)
            oloss.backward()
            dev = {"logits": (ologits - logits).abs().max().item()}
            otrain = sorted(k for k, v in osd.items() if v.requires_grad and ".global_query." not in k and ".local_query." not in k)
            assert otrain == sorted(trainable), (set(otrain) ^ set(trainable))
            gdev = 0.0
            for k in trainable:
                og = osd[k].grad
                # the alias keys share storage in the reference; the oracle reads only the canonical key
                gdev = max(gdev, ((og - g[k]).abs().max() / (g[k].abs().max() + 1e-12)).item())
            dev["grad_rel"] = gdev
            for k in taps:
                if k in otaps:
                    dev["tap/" + k] = float(np.abs(tap(otaps[k]) - taps[k]).max())
            out["meta/oracle_dev"] = repr(dev)
            print(f"  oracle vs reference: logits {dev['logits']:.3e}  grad_rel {gdev:.3e}  "
                  f"taps max {max([v for k, v in dev.items() if k.startswith('tap/')] or [0]):.3e}")
            assert dev["logits"] < 2e-5 and gdev < 2e-4, dev
    grads = {k: v / shards for k, v in grads_acc.items()}
    out["logits"] = np.concatenate(logits_all, 0)
    out["argmax"] = out["logits"].argmax(-1).astype(np.int64)
    for k, gk in grads.items():
        out["gradnorm/" + k] = np.float32(gk.norm().item())
        if want_full_grad(k, depth) and gk.numel() <= 70000:
            out["grad/" + k] = gk.numpy().copy()
    path = os.path.join(outdir, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: logits[0]={out['logits'][0]}, argmax={out['argmax'].tolist()}, "
          f"{len(trainable)} trainable tensors, {os.path.getsize(path) / 1024:.0f} KiB, {time.time() - t0:.1f}s")


def focal_case(mods, outdir):
    """FocalLoss fwd + dL/dlogits on logits spanning <0, (0,1), >1."""
    fl = mods["fl"].FocalLoss(gamma=1.2)
    lg = torch.from_numpy(synth.symmetric("focal.logits", (16, 5), 2.0)).requires_grad_(True)
    y = torch.from_numpy(synth.labels(0, 16))
    loss = fl(lg, y)
    loss.backward()
    np.savez_compressed(os.path.join(outdir, "focal_loss.npz"), logits=lg.detach().numpy(), target=y.numpy(),
                        loss=np.float32(loss.item()), grad=lg.grad.numpy())
    print("focal_loss:", loss.item())
    # class weights + ignored rows + both reductions (focal_loss.py:54-58, 60-68, 103-118)
    w = torch.tensor([0.5, 1.0, 2.0, 1.5, 0.7])
    y2 = y.clone()
    y2[3] = -100
    y2[11] = -100
    out = {"logits": lg.detach().numpy(), "target": y2.numpy(), "weights": w.numpy()}
    for red in ("mean", "sum"):
        lg2 = lg.detach().clone().requires_grad_(True)
        loss = mods["fl"].FocalLoss(gamma=1.2, weights=w, reduction=red)(lg2, y2)
        loss.backward()
        out["loss_" + red], out["grad_" + red] = np.float32(loss.item()), lg2.grad.numpy()
        print("focal_loss weighted/ignored", red, loss.item())
    # reduction='none' (focal_loss.py:40,117-118): the per-sample vector and the gradient of its plain sum weighted by `up`
    up = torch.from_numpy(synth.symmetric("focal.up", (16,), 0.5)) + 1.0
    lg3 = lg.detach().clone().requires_grad_(True)
    vec = mods["fl"].FocalLoss(gamma=1.2, weights=w, reduction="none")(lg3, y2)
    (vec * up).sum().backward()
    out["up"], out["loss_none"], out["grad_none"] = up.numpy(), vec.detach().numpy(), lg3.grad.numpy()
    np.savez_compressed(os.path.join(outdir, "focal_loss_weighted.npz"), **out)


def fake_timm_state_dict(C=16, depth=2, seed=0):
    """A tiny state dict with the key set and tensor ranks of a timm in21k ViT (vision_transformer.py of timm 0.x: cls_token, pos_embed,
    patch_embed.proj, blocks.i.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}, norm, pre_logits.fc, head)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    sd = {"cls_token": r(1, 1, C), "pos_embed": r(1, 197, C), "patch_embed.proj.weight": r(C, 3, 16, 16), "patch_embed.proj.bias": r(C)}
    for i in range(depth):
        b = f"blocks.{i}."
        sd.update({b + "norm1.weight": r(C), b + "norm1.bias": r(C), b + "attn.qkv.weight": r(3 * C, C), b + "attn.qkv.bias": r(3 * C),
                   b + "attn.proj.weight": r(C, C), b + "attn.proj.bias": r(C), b + "norm2.weight": r(C), b + "norm2.bias": r(C),
                   b + "mlp.fc1.weight": r(4 * C, C), b + "mlp.fc1.bias": r(4 * C), b + "mlp.fc2.weight": r(C, 4 * C), b + "mlp.fc2.bias": r(C)})
    sd.update({"norm.weight": r(C), "norm.bias": r(C), "pre_logits.fc.weight": r(C, C), "pre_logits.fc.bias": r(C),
               "head.weight": r(7, C), "head.bias": r(7)})
    return sd


def pretrain_case(mods, outdir):
    """The reference's load_pretrain (load_pretrained.py:8-99) run on a synthetic timm-shaped state dict: timm.create_model is
    pointed at an object that returns it (there is no network); everything after line 24 is the reference's own code."""
    import tempfile
    lp = mods["lp"]
    for tag, num_patches, depth_dim in (("", 1000, 12), ("_n216_d4", 216, 4)):
        sd = fake_timm_state_dict()

        class Fake:
            def state_dict(self):
                return sd
        lp.timm.create_model = lambda name, pretrained=True: Fake()
        with tempfile.TemporaryDirectory() as tmp:
            out = mods["lp_orig"]("vit-t16", num_patches, depth_dim, tmp)
            saved = sorted(os.listdir(tmp))
        fx = {"in/" + k: v.numpy() for k, v in sd.items()}
        fx.update({"out/" + k: v.numpy() for k, v in out.items()})
        fx["meta/num_patches"], fx["meta/depth_dim"], fx["meta/saved_as"] = np.int64(num_patches), np.int64(depth_dim), np.array(saved)
        path = os.path.join(outdir, f"pretrain_convert{tag}.npz")
        np.savez_compressed(path, **fx)
        print("pretrain_convert" + tag, len(out), "keys,", saved, f"{os.path.getsize(path) / 1024:.0f} KiB")


def mask_case(mods, outdir):
    """MWSA window masks of the reference for both shipped local_k (gaviko.py:212-227), as bit-packed allow maps."""
    for lk in ((6, 6, 6), (3, 6, 6), (3, 3, 3)):
        m = mods["gv"].LocalSelfAttention(32, lk, (10, 10, 10)).mask[0]
        allow = (m == 0).numpy()
        np.savez_compressed(os.path.join(outdir, f"mwsa_mask_{lk[0]}{lk[1]}{lk[2]}.npz"), allow=np.packbits(allow, axis=1),
                            count=allow.sum(1).astype(np.int32))
        print("mask", lk, allow.sum(1).min(), allow.sum(1).max(), allow.sum(1).mean())


def main():
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    mods = import_reference()
    names = sys.argv[1:] or (["focal", "mask", "pretrain"] + list(CASES))
    for n in names:
        if n == "focal":
            focal_case(mods, outdir)
        elif n == "mask":
            mask_case(mods, outdir)
        elif n == "pretrain":
            pretrain_case(mods, outdir)
        else:
            run_case(mods, n, outdir)


if __name__ == "__main__":
    main()
