"""Backbone nn.Dropout live in training (the shipped configs set dropout = emb_dropout = prompt_dropout = 0.1, and the classes without a
train() override keep them live): the HIP path with p = 0.1 against the ORACLE run with exactly the masks the kernels drew
(tests/dropmask.py rebuilds them from the device seed word)."""
import numpy as np
import pytest
import torch

import dropmask
import oracle

pytestmark = pytest.mark.gpu

BASE = dict(image_size=160, image_patch_size=16, frames=120, frame_patch_size=12, num_classes=5, channels=1, pool="cls", dim_head=64)
P = 0.1


def build(method, extra, dev):
    from gaviko_amd.registry import build_model
    from gaviko_amd.utils import synth
    cfg = dict(BASE, backbone="vit-t16", method=method, **extra)
    m = build_model(cfg)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    m.to(dev)
    m.train()
    return m, cfg


def masks_for(eng, word, B, backbone_p, emb_p, prompt_p):
    from gaviko_amd import engine as E
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    masks = {}
    C, H, mlp = eng.C, eng.heads, eng.mlp
    if emb_p > 0:
        masks[("emb", 0)] = t(dropmask.rows_mask(E.SEED_EMB + word, B * eng.T, C, emb_p)).view(B, eng.T, C)
        if eng.kind == "vpt":                       # drawn over the [cls | prompt slots | patches] rows of the buffer; the reference drops
            m_ = masks[("emb", 0)]                  # [cls | patches] before the prompts go in (vpt.py:171): the slots' draws are unused
            masks[("emb", 0)] = torch.cat((m_[:, :1], m_[:, 1 + eng.P:]), dim=1)
        if eng.kind == "gaviko":                    # self.dropout is applied a second time, to the local tokens (gaviko.py:548)
            masks[("emb_local", 0)] = t(dropmask.rows_mask(E.SEED_EMB + 1 + word, B * eng.N, C, emb_p)).view(B, eng.N, C)
    for i in range(eng.depth):
        T = eng.Ts[i]
        if backbone_p > 0:
            s = E.SEED_LAYER + 8 * i + word
            masks[("attn", i)] = t(dropmask.attn_mask(s, B, H, T, backbone_p))
            masks[("proj", i)] = t(dropmask.rows_mask(s + 1, B * T, C, backbone_p)).view(B, T, C)
            masks[("act", i)] = t(dropmask.rows_mask(s + 2, B * T, mlp, backbone_p)).view(B, T, mlp)
            masks[("ff", i)] = t(dropmask.rows_mask(s + 3, B * T, C, backbone_p)).view(B, T, C)
        if prompt_p > 0:
            masks[("prompt", i)] = t(dropmask.rows_mask(E.SEED_PROMPT + i + word, B * eng.P, C, prompt_p)).view(B, eng.P, C)
    return masks


CASES = [("fft", dict(dropout=P, emb_dropout=P), (P, P, 0.0)),
         ("bitfit", dict(dropout=P, emb_dropout=P), (P, P, 0.0)),
         ("linear", dict(dropout=P, emb_dropout=P), (P, P, 0.0)),
         ("melo", dict(dropout=P, emb_dropout=P, r=4, alpha=4), (P, P, 0.0)),
         ("adaptformer", dict(dropout=P, emb_dropout=P, freeze_vit=False), (P, P, 0.0)),     # unfrozen: the backbone's dropouts follow .training (adaptformer.py:175-191)
         ("deep_vpt", dict(dropout=P, emb_dropout=P, num_prompts=8, prompt_dim=64, prompt_dropout=P, freeze_vit=False, deep_prompt=True), (P, P, P)),
         ("shallow_vpt", dict(dropout=P, emb_dropout=P, num_prompts=8, prompt_dim=64, prompt_dropout=P, freeze_vit=False, deep_prompt=False), (P, P, P)),
         ("dvpt", dict(dropout=P, emb_dropout=P, freeze_vit=False, num_prompts=8), (P, P, 0.0)),
         ("evp", dict(dropout=P, emb_dropout=P, freeze_vit=False), (P, P, 0.0)),
         ("ssf", dict(dropout=P, emb_dropout=P, freeze_vit=False), (P, P, 0.0)),
         # Gaviko(freeze_vit=False) with the shipped dropout = emb_dropout = 0.1 (gaviko.py:513-528); the MWSA dropouts have their own test below
         ("gaviko", dict(dropout=P, emb_dropout=P, freeze_vit=False, num_prompts=8, prompt_latent_dim=20, local_dim=20, local_k=(3, 6, 6),
                         DHW=(10, 10, 10), attn_drop=0.0, proj_drop=0.0, share_factor=1), (P, P, 0.0)),
         ("deep_vpt", dict(dropout=P, emb_dropout=P, num_prompts=8, prompt_dim=64, prompt_dropout=P, freeze_vit=True, deep_prompt=True), (0.0, 0.0, P)),
         ("shallow_vpt", dict(dropout=P, emb_dropout=P, num_prompts=8, prompt_dim=64, prompt_dropout=P, freeze_vit=True, deep_prompt=False), (0.0, 0.0, P))]


@pytest.mark.parametrize("method,extra,live", CASES)
def test_training_step_with_live_dropout_matches_oracle_with_the_same_masks(dev, method, extra, live):
    from gaviko_amd.utils import synth
    B = 2
    m, cfg = build(method, extra, dev)
    x = torch.from_numpy(synth.volumes(0, B))
    y = torch.from_numpy(synth.labels(0, B))
    logits = m(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    eng = m._engine()
    word = int(eng._ws["seed"].item())
    # ---- the oracle on the CPU with the masks of THIS step
    masks = masks_for(eng, word, B, *live)
    for k, v in masks.items():
        keep = (v > 0).float().mean().item()
        assert abs(keep - (1 - P)) < 0.02, (k, keep)
    osd = {k: v.detach().cpu().clone().requires_grad_(oracle.trainable(method, k, cfg)) for k, v in m.state_dict().items()}
    ologits = oracle.FORWARD[method](osd, x, dict(cfg, _masks=masks), None)
    oloss = torch.nn.functional.cross_entropy(ologits, y)
    oloss.backward()
    lg = logits.detach().cpu()
    scale = ologits.abs().max().item()
    assert (lg - ologits.detach()).abs().max().item() < 1.5e-2 * scale, (lg, ologits)
    # argmax must agree wherever the oracle's own decision is not a tie at the stated tolerance (a draw of the dropout masks can leave
    # two classes closer than bf16 rounding; the mask-free parity tests in test_model_gpu.py keep the unconditional check)
    top2 = ologits.detach().topk(2, dim=-1).values
    decided = (top2[:, 0] - top2[:, 1]) > 1.5e-2 * scale
    assert (lg.argmax(-1) == ologits.argmax(-1))[decided].all()
    # and the masks matter: without them the oracle lands somewhere else
    plain = oracle.FORWARD[method]({k: v.detach() for k, v in osd.items()}, x, cfg, None)
    # (prompt dropout touches 8 of ~1000 tokens: within bf16 noise on the logits -- there the prompt gradients below carry the proof:
    #  a dropped element contributes exactly nothing to them)
    if "vpt" not in method:
        assert (plain - ologits.detach()).abs().max().item() > 5 * (lg - ologits.detach()).abs().max().item()
    errs, cosines = [], []
    for k, p in m.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, k
        want = osd[k].grad
        if want is None or ".global_query." in k:
            continue
        wn = want.norm().item()
        errs.append((abs(p.grad.norm().item() - wn) / max(wn, 1e-12), k))
        if wn > 1e-6 and want.numel() <= 200000:
            got = p.grad.cpu()
            if method == "adaptformer":
                # AdaptFormer's ReLU mask comes from bf16 pre-activations: a unit within rounding of zero flips, and with it one whole row of
                # its adapter's dW (4.7e-1 of the largest element on layers.3.1.down_adapter_proj.weight; the reference's own arithmetic
                # with bf16 operands moves single elements by 3.3e-1, tests/golden/bf16_noise_floor.json).  Direction and norm are the
                # robust statements: cosine to the oracle's gradient, the norm bounds below, and the fp32 run of this case (next test),
                # which is exact to 5e-4 on every element.
                cos = torch.nn.functional.cosine_similarity(got.flatten().double(), want.flatten().double(), dim=0).item()
                cosines.append((cos, k))
            else:
                e = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-12)
                # GAViKO's gl_balancer scalars (norms 1e-4, 100x below every other tensor) are cancellation-dominated: 1.4e-1 here, 1.1e-1 in
                # the reference's own bf16-operand arithmetic (tests/golden/bf16_noise_floor.json); exact on the fp32 path (next test)
                assert e < (0.2 if "gl_balancer" in k else 6e-2), (k, e)
    if cosines:
        print("lowest cosines:", sorted(cosines)[:5])
        assert min(c for c, _ in cosines) > 0.97, sorted(cosines)[:5]
    e = np.array([v[0] for v in errs])
    assert len(e) > 0 and np.median(e) < 1.5e-2 and e.max() < 0.15, sorted(errs, reverse=True)[:5]


def test_dropout_draws_fresh_masks_every_step_and_eval_is_deterministic(dev):
    from gaviko_amd.utils import synth
    m, cfg = build("fft", dict(dropout=P, emb_dropout=P), dev)
    x = torch.from_numpy(synth.volumes(0, 2)).to(dev)
    a = m(x).detach().clone()
    b = m(x).detach().clone()
    assert not torch.equal(a, b)                                         # the device epoch word moved between the two (replayed) forwards
    m.eval()
    with torch.no_grad():
        c, d = m(x).clone(), m(x).clone()
    assert torch.equal(c, d)


@pytest.mark.parametrize("method,extra,live", [c for c in CASES if c[0] in ("fft", "melo", "deep_vpt", "adaptformer", "gaviko", "dvpt", "evp", "ssf")])
def test_fp32_path_with_live_dropout_is_exact_against_the_oracle(dev, method, extra, live):
    """The same masks through exact fp32 arithmetic: logits to 2e-5, every gradient to 2e-4 -- the mask logic itself, free of bf16 noise."""
    from gaviko_amd.utils import synth
    B = 2
    m, cfg = build(method, dict(extra, precision="fp32"), dev)
    x = torch.from_numpy(synth.volumes(0, B))
    y = torch.from_numpy(synth.labels(0, B))
    logits = m(x.to(dev))
    torch.nn.functional.cross_entropy(logits, y.to(dev)).backward()
    torch.cuda.synchronize()
    eng = m._engine()
    masks = masks_for(eng, int(eng._ws["seed"].item()), B, *live)
    ocfg = {k: v for k, v in cfg.items() if k != "precision"}
    osd = {k: v.detach().cpu().clone().requires_grad_(oracle.trainable(method, k, ocfg)) for k, v in m.state_dict().items()}
    ologits = oracle.FORWARD[method](osd, x, dict(ocfg, _masks=masks), None)
    torch.nn.functional.cross_entropy(ologits, y).backward()
    assert (logits.detach().cpu() - ologits.detach()).abs().max().item() < 2e-5 * max(1.0, ologits.abs().max().item())
    worst, who = 0.0, None
    for k, p in m.named_parameters():
        if p.requires_grad and osd[k].grad is not None and ".global_query." not in k:
            want = osd[k].grad
            e = (p.grad.cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-12)
            if e > worst:
                worst, who = e, k
    assert worst < 5e-4, (worst, who)


@pytest.mark.parametrize("backbone,B,P_,lk,share", [("vit-t16", 2, 8, (3, 6, 6), 1), ("vit-t16", 2, 8, (3, 6, 6), 2),
                                                     ("vit-b16", 4, 32, (6, 6, 6), 1)])           # the last one IS BASELINE cfg2 / bench.py
def test_gaviko_step_with_live_mwsa_dropouts_matches_oracle_with_the_same_masks(dev, backbone, B, P_, lk, share):
    """GAViKO trains with LocalSelfAttention's attn_drop / proj_drop live (gaviko.py:513-528 keeps only the backbone in eval; the shipped config
    has 0.2 / 0.2 and bench.py times exactly that).  The masks are regenerated on the host from the device seed word (window attention: layer
    seed 2i, projection: 2i + 1) and the ORACLE is run with them: forward and every gradient -- including the chained MWSA kernels that carry
    a dropout mask across a layer boundary -- against the reference arithmetic."""
    from gaviko_amd.registry import build_model
    from gaviko_amd.utils import synth
    p = 0.2
    cfg = dict(BASE, backbone=backbone, method="gaviko", num_prompts=P_, prompt_latent_dim=20, local_dim=20, local_k=lk, DHW=(10, 10, 10),
               attn_drop=p, proj_drop=p, freeze_vit=True, share_factor=share, fp16=False)
    cfg["dropout"] = cfg["emb_dropout"] = 0.0
    m = build_model(cfg)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    m.to(dev).train()
    N = 1000
    x = torch.from_numpy(synth.volumes(0, B))
    y = torch.from_numpy(synth.labels(0, B))
    for _ in range(4):                                                   # eager, eager, recorded, replayed
        for q in m.parameters():
            q.grad = None
        logits = m(x.to(dev))
        torch.nn.functional.cross_entropy(logits, y.to(dev)).backward()
        torch.cuda.synchronize()
    eng = m._engine()
    assert eng._last_run[0] == "replayed"
    word = int(eng._ws["seed"].item())
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    masks = {}
    for i in range(eng.depth):
        masks[("mwsa_attn", i)] = t(dropmask.window_attn_mask(2 * i + word, B, N, p))
        masks[("mwsa_proj", i)] = t(dropmask.rows_mask(2 * i + 1 + word, B * N, eng.C, p)).view(B, N, eng.C)
    keep = masks[("mwsa_proj", 3)].gt(0).float().mean().item()
    assert abs(keep - (1 - p)) < 0.01
    osd = {k: v.detach().cpu().clone().requires_grad_(oracle.trainable("gaviko", k)) for k, v in m.state_dict().items()}
    ologits = oracle.FORWARD["gaviko"](osd, x, dict(cfg, _masks=masks), None)
    torch.nn.functional.cross_entropy(ologits, y).backward()
    lg = logits.detach().cpu()
    scale = ologits.abs().max().item()
    assert (lg - ologits.detach()).abs().max().item() < 1.5e-2 * scale, (lg, ologits)
    plain = oracle.FORWARD["gaviko"]({k: v.detach() for k, v in osd.items()}, x, cfg, None)
    assert (plain - ologits.detach()).abs().max().item() > 3 * (lg - ologits.detach()).abs().max().item()      # the masks matter
    errs = []
    for k, q in m.named_parameters():
        if not q.requires_grad or osd[k].grad is None:
            continue
        want = osd[k].grad
        wn = want.norm().item()
        errs.append((abs(q.grad.norm().item() - wn) / max(wn, 1e-12), k))
    e = np.array([v[0] for v in errs])
    assert len(e) > 20 and np.median(e) < 1.5e-2 and np.percentile(e, 90) < 5e-2 and e.max() < 0.15, sorted(errs, reverse=True)[:5]
