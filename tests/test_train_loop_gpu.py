"""End to end on the GPU: the reference's training loop (train.py:80-504) rebuilt from this package's pieces (examples/train_synthetic.py)
-- data, device transforms, model, fused loss, fused optimizer, metrics, trainable-only checkpoint -- runs, learns and round-trips."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


@pytest.mark.parametrize("method", ["gaviko", "deep_vpt", "bitfit"])
def test_training_loop_end_to_end(dev, tmp_path, method):
    import train_synthetic as ts
    from gaviko_amd.registry import build_model
    res = ts.run(method=method, backbone="vit-t16", epochs=4, samples=8, out=str(tmp_path), batch_size=4, lr=2e-3, log=lambda *a: None)
    h = res["history"]
    assert all(torch.isfinite(torch.tensor(e["train_loss"])) for e in h)
    if method == "bitfit":
        # 76 k bias parameters under live dropout + random augmentation: the epoch MEANS move by +-1 % with the draws, so the loop run only
        # has to stay finite here; that bitfit LEARNS is asserted deterministically in test_bitfit_learns_on_a_fixed_batch below
        pass
    else:
        assert min(e["train_loss"] for e in h[1:]) < h[0]["train_loss"], h   # it learns something on 8 volumes
    assert os.path.exists(res["checkpoint"]) and os.path.exists(res["results_csv"])
    # the trainable-only checkpoint restores the trained model's predictions in a fresh instance (eval.py:87-92)
    ck = torch.load(res["checkpoint"], map_location="cpu")
    model = res["model"]
    fresh = build_model(res["config"]["model"]).to(dev)
    frozen = {k: v for k, v in model.state_dict().items() if k not in ck}
    fresh.load_state_dict({**frozen, **ck}, strict=True)
    model.eval(); fresh.eval()
    xb, _ = next(iter(res["val_loader"]))
    xb = res["pre"].val_transforms(xb.to(dev))
    with torch.no_grad():
        a, b = model(xb), fresh(xb)
    # the checkpoint was written at the best epoch, which may precede the last one: compare only when it IS the last epoch
    if h[-1]["val_acc"] > max(e["val_acc"] for e in h[:-1]):
        assert torch.equal(a, b)
    assert torch.isfinite(b).all()


def test_bitfit_learns_on_a_fixed_batch(dev):
    """`--method bitfit` (train.py:131-137: biases + head train) with every dropout off and one fixed batch: the step is deterministic, so
    "the loss falls and the bias tensors move" is a hard assertion -- a no-op optimizer or zeroed bias gradients fail it."""
    from gaviko_amd.optim import FusedAdamOneCycle
    from gaviko_amd.registry import build_model
    from gaviko_amd.utils import synth
    model = build_model(_tiny_cfg("bitfit"))
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model.to(dev).train()
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    assert names and all(("bias" in k) or ("head" in k) for k in names)
    before = {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}
    x = torch.from_numpy(synth.volumes(0, 4)).to(dev)
    y = torch.from_numpy(synth.labels(0, 4)).to(dev)
    steps = 12
    opt = FusedAdamOneCycle(model, lr=2e-3, eps=1e-8, max_norm=1.0, max_lr=2e-3, total_steps=steps, pct_start=0.3, div_factor=10, final_div_factor=1000)
    hist = []
    for _ in range(steps):
        loss = torch.nn.functional.cross_entropy(model(x), y)
        loss.backward()
        opt.step()
        opt.zero_grad()
        hist.append(loss.item())
    assert all(torch.isfinite(torch.tensor(hist))), hist
    assert hist[-1] < hist[0] - 1e-3 and min(hist[1:]) < hist[0], hist
    after = dict(model.named_parameters())
    moved = [k for k in names if (after[k].detach() - before[k]).abs().max().item() > 0]
    backbone_bias = [k for k in moved if "head" not in k]
    assert len(moved) == len(names) and backbone_bias, (len(moved), len(names))


def _tiny_cfg(method, **kw):
    cfg = dict(image_size=160, image_patch_size=16, frames=120, frame_patch_size=12, num_classes=5, channels=1, pool="cls", dim_head=64,
               dropout=0.0, emb_dropout=0.0, backbone="vit-t16", method=method, fp16=False)
    cfg.update(kw)
    return cfg


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_fft_fused_adam_keeps_weight_shadows_fresh(dev, precision):
    """`--method fft` + FusedAdamOneCycle: the optimizer writes the backbone weights through raw pointers (no torch version bump), so the
    engine's bf16 operand / transposed dgrad shadows must be invalidated by the step.  After a few steps the model's logits must equal
    those of a FRESH model (fresh engine, fresh shadows) loaded with the trained state_dict."""
    from gaviko_amd.optim import FusedAdamOneCycle
    from gaviko_amd.registry import build_model
    from gaviko_amd.utils import synth
    cfg = _tiny_cfg("fft", precision=precision)
    model = build_model(cfg)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model.to(dev).train()
    x = torch.from_numpy(synth.volumes(0, 2)).to(dev)
    y = torch.from_numpy(synth.labels(0, 2)).to(dev)
    opt = FusedAdamOneCycle(model, lr=1e-3, max_norm=1.0)
    first = None
    for step in range(4):
        logits = model(x)
        if first is None:
            first = logits.detach().clone()
        torch.nn.functional.cross_entropy(logits, y).backward()
        opt.step()
        opt.zero_grad()
    model.eval()
    with torch.no_grad():
        got = model(x).clone()
    assert (got - first).abs().max().item() > 1e-4, "four Adam steps at lr 1e-3 must move the logits"
    fresh = build_model(cfg)
    fresh.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    fresh.to(dev).eval()
    with torch.no_grad():
        want = fresh(x)
    assert torch.equal(got, want), f"stale operand shadows: max diff {(got - want).abs().max().item():.3e}"


def test_second_forward_before_backward_raises(dev):
    """The engine keeps ONE forward's saved activations: backward through an older forward must fail loudly, not use the newer state."""
    from gaviko_amd import lib
    from gaviko_amd.registry import build_model
    from gaviko_amd.utils import synth
    model = build_model(_tiny_cfg("linear"))
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model.to(dev).train()
    x = torch.from_numpy(synth.volumes(0, 2)).to(dev)
    a = model(x[:1])
    b = model(x[1:])
    with pytest.raises(lib.GavikoHipError, match="saved activations"):
        (a.sum() + b.sum()).backward()
    model(x[:1]).sum().backward()                            # a single forward / backward pair still works afterwards


def test_fft_stepped_by_torch_adam_and_clip_grad_norm(dev):
    """What the unchanged train.py does (train.py:185-189,315-318): torch.optim.Adam over the trainable tensors + clip_grad_norm_ on an
    UNFROZEN model.  torch's in-place updates bump the parameters' version counters, which is what refreshes the engine's bf16 operand
    shadows: after a few steps the logits must equal those of a FRESH model loaded with the trained state_dict, and the clipped norm
    torch reports must be the norm of the engine's flat gradient buffer."""
    from gaviko_amd.registry import build_model
    from gaviko_amd.utils import synth
    cfg = _tiny_cfg("fft")
    model = build_model(cfg)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model.to(dev).train()
    x = torch.from_numpy(synth.volumes(0, 2)).to(dev)
    y = torch.from_numpy(synth.labels(0, 2)).to(dev)
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-3, eps=1e-8)
    first = None
    for step in range(3):
        opt.zero_grad()
        logits = model(x)
        if first is None:
            first = logits.detach().clone()
        torch.nn.functional.cross_entropy(logits, y).backward()
        flat_norm = model._engine().flat_grad.norm().item()
        total = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0).item()
        assert abs(total - flat_norm) < 1e-4 * max(1.0, flat_norm)
        opt.step()
    model.eval()
    with torch.no_grad():
        got = model(x).clone()
    assert (got - first).abs().max().item() > 1e-4
    fresh = build_model(cfg)
    fresh.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    fresh.to(dev).eval()
    with torch.no_grad():
        want = fresh(x)
    assert torch.equal(got, want), f"stale operand shadows after torch.optim.Adam: max diff {(got - want).abs().max().item():.3e}"


def test_autograd_grad_on_a_hot_path_model_fails_loudly(dev):
    """The autograd node returns no input gradients (the engine writes .grad itself, one anchor parameter ties the node into the graph):
    torch.autograd.grad must raise instead of handing back None for every tensor."""
    from gaviko_amd import lib
    from gaviko_amd.registry import build_model
    from gaviko_amd.utils import synth
    model = build_model(_tiny_cfg("linear"))
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model.to(dev).train()
    x = torch.from_numpy(synth.volumes(0, 1)).to(dev)
    y = torch.from_numpy(synth.labels(0, 1)).to(dev)
    params = [p for p in model.parameters() if p.requires_grad]
    with pytest.raises(lib.GavikoHipError, match="loss.backward"):
        torch.autograd.grad(torch.nn.functional.cross_entropy(model(x), y), params, allow_unused=True)
    with pytest.raises(RuntimeError):            # a tensor that is not the node's anchor: torch itself reports it as unused (GavikoHipError is a RuntimeError too)
        torch.autograd.grad(torch.nn.functional.cross_entropy(model(x), y), params[-1:])
    torch.nn.functional.cross_entropy(model(x), y).backward()          # the supported call still works afterwards
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)


def test_zero_grad_fast_path_equals_the_walk(dev):
    """model.zero_grad(set_to_none=False) after the first backward is one memset of the flat buffer and lets the next backward skip its
    per-parameter bookkeeping: gradients must be bit-identical to the set_to_none=True route, and accumulation (no zero_grad between two
    backwards) must still add."""
    from gaviko_amd.registry import build_model
    from gaviko_amd.utils import synth
    cfg = _tiny_cfg("gaviko", num_prompts=8, prompt_latent_dim=20, local_dim=20, local_k=(3, 6, 6), DHW=(10, 10, 10), attn_drop=0.0, proj_drop=0.0,
                    freeze_vit=True, share_factor=1)
    model = build_model(cfg)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model.to(dev)
    model.train()
    x = torch.from_numpy(synth.volumes(0, 2)).to(dev)
    y = torch.from_numpy(synth.labels(0, 2)).to(dev)
    step = lambda: torch.nn.functional.cross_entropy(model(x), y).backward()
    model.zero_grad(set_to_none=True); step()
    ref = model._engine().flat_grad.clone()
    for _ in range(4):                                                  # past the warm-up: replayed plans
        model.zero_grad(set_to_none=False)
        assert model.__dict__["_grads_zeroed"] and float(model._engine().flat_grad.abs().max()) == 0.0
        step()
        assert torch.equal(model._engine().flat_grad, ref)
    step()                                                              # no zero_grad: torch semantics = accumulate
    assert torch.allclose(model._engine().flat_grad, 2 * ref, rtol=1e-6, atol=1e-12)
    model.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in model.parameters())


def test_zero_grad_fast_path_is_rechecked_at_backward(dev):
    """the fast path trusted a flag set at zero_grad time.  A torch optimizer's zero_grad() (set_to_none=True: it never calls
    model.zero_grad) or a re-allocated flat buffer (make_reducer / set_bucket_layers) between model.zero_grad(set_to_none=False) and
    backward() left every .grad None / pointing at the old buffer while the engine wrote elsewhere -- optimizer.step() then skipped or used
    stale zeros without a word.  The promise is now tied to the buffer object and re-checked against the .grad views in backward."""
    from gaviko_amd.registry import build_model
    from gaviko_amd.utils import synth
    cfg = _tiny_cfg("gaviko", num_prompts=8, prompt_latent_dim=20, local_dim=20, local_k=(3, 6, 6), DHW=(10, 10, 10), attn_drop=0.0, proj_drop=0.0,
                    freeze_vit=True, share_factor=1)
    model = build_model(cfg)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model.to(dev)
    model.train()
    x = torch.from_numpy(synth.volumes(0, 2)).to(dev)
    y = torch.from_numpy(synth.labels(0, 2)).to(dev)
    step = lambda: torch.nn.functional.cross_entropy(model(x), y).backward()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.0)
    model.zero_grad(set_to_none=True); step()
    named = dict(model.named_parameters())
    ref = {n: named[n].grad.clone() for n in model._engine().trainable_names()}
    # (1) torch optimizer's zero_grad between the arming and the backward
    model.zero_grad(set_to_none=False)
    assert model.__dict__["_grads_zeroed"] is not None
    opt.zero_grad()                                                     # set_to_none=True: every .grad is None now
    assert all(p.grad is None for p in params)
    step()
    assert all(named[n].grad is not None and torch.equal(named[n].grad, ref[n]) for n in ref)
    # (2) the flat buffer re-allocated in between (another bucket layout)
    model.zero_grad(set_to_none=False)
    old_buf = model._engine().flat_grad
    model._engine().set_bucket_layers(1)
    step()
    eng = model._engine()
    assert eng.flat_grad is not old_buf
    for n in ref:
        assert named[n].grad.data_ptr() == eng._flat_grad["views"][n].data_ptr(), n     # .grad follows the NEW buffer
        assert torch.allclose(named[n].grad, ref[n], rtol=1e-6, atol=1e-12), n
    # (3) and the fast path itself still runs when nothing happened in between
    model.zero_grad(set_to_none=False); step()
    assert all(torch.allclose(named[n].grad, ref[n], rtol=1e-6, atol=1e-12) for n in ref)
