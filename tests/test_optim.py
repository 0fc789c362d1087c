"""Fused optimisation step against the reference's torch calls (oracle/optim_ref.py)."""
import numpy as np
import pytest
import torch


SCHED = dict(max_lr=3e-4, total_steps=40, pct_start=0.3, div_factor=10, final_div_factor=1000)


@pytest.mark.parametrize("three_phase,anneal", [(False, "cos"), (True, "cos"), (False, "linear")])
def test_onecycle_mirror_matches_torch_scheduler(three_phase, anneal):
    """CPU: the host-side schedule reproduces OneCycleLR's lr and beta1 cycling at every step."""
    from gaviko_amd.optim import OneCycle
    from oracle import optim_ref
    p = [torch.nn.Parameter(torch.zeros(3))]
    opt, sch = optim_ref.make(p, lr=1e-3, eps=1e-8, anneal_strategy=anneal, three_phase=three_phase, **SCHED)
    mine = OneCycle(anneal_strategy=anneal, three_phase=three_phase, **SCHED)
    for t in range(SCHED["total_steps"]):
        lr, b1 = mine.at(t)
        assert abs(lr - opt.param_groups[0]["lr"]) <= 1e-12 + 1e-9 * lr
        assert abs(b1 - opt.param_groups[0]["betas"][0]) <= 1e-12
        p[0].grad = torch.ones(3)
        opt.step()
        if t + 1 < SCHED["total_steps"]:
            sch.step()


@pytest.mark.gpu
def test_fused_clip_adam_onecycle_matches_reference_step(dev):
    """GPU: 12 steps of gvk_sumsq + gvk_adam_step on 7 ragged tensors against clip_grad_norm_ + Adam + OneCycleLR on the CPU.
    Gradients alternate between large (clipped) and tiny (unclipped) so both branches run."""
    from gaviko_amd.optim import FusedAdamOneCycle
    from oracle import optim_ref
    g = torch.Generator().manual_seed(7)
    shapes = [(5,), (20, 768), (768, 20), (1,), (1025,), (3, 341), (64, 20)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    mine = [p.detach().clone().to(dev) for p in ref]
    total = sum(p.numel() for p in ref)
    flat = torch.zeros(total, device=dev)
    opt, sch = optim_ref.make(ref, lr=1e-3, eps=1e-8, **SCHED)
    fused = FusedAdamOneCycle(mine, lr=1e-3, eps=1e-8, max_norm=1.0, flat_grad=flat, **SCHED)
    for t in range(12):
        scale = 1.0 if t % 2 == 0 else 1e-4
        grads = [torch.randn(s, generator=g) * scale for s in shapes]
        for p, gr in zip(ref, grads):
            p.grad = gr.clone()
        flat.copy_(torch.cat([gr.reshape(-1) for gr in grads]).to(dev))
        n_ref = optim_ref.step(ref, opt, sch, 1.0)
        fused.step()
        torch.cuda.synchronize()
        assert abs(fused.grad_norm().item() - n_ref.item()) <= 1e-5 * n_ref.item()
        for p, q in zip(ref, mine):
            assert torch.allclose(q.cpu(), p.detach(), atol=1e-7, rtol=2e-6), t
        off = 0                                              # the stored gradients are the clipped ones, as torch leaves them
        for p in ref:
            assert torch.allclose(flat[off: off + p.numel()].cpu(), p.grad.reshape(-1), atol=1e-9, rtol=1e-5)
            off += p.numel()


@pytest.mark.gpu
def test_fused_optimizer_drives_a_model(dev):
    """GPU: the optimizer binds to a gaviko_amd model (flat gradient buffer + trainable tensors) and the loss goes down."""
    import test_model_gpu as tm
    from gaviko_amd.optim import FusedAdamOneCycle
    from gaviko_amd.utils import synth
    m, cfg = tm.build("gaviko", "vit-t16", dict(tm.GAVIKO), dev)
    opt = FusedAdamOneCycle(m, lr=1e-3, eps=1e-8, **SCHED)
    x = torch.from_numpy(synth.volumes(0, 2)).to(dev)
    y = torch.from_numpy(synth.labels(0, 2)).to(dev)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(x), y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses
    frozen = [p for n, p in m.named_parameters() if not p.requires_grad]
    assert all(p.grad is None for p in frozen)


@pytest.mark.gpu
def test_adam_moments_follow_names_when_the_flat_layout_moves(dev):
    """exp_avg / exp_avg_sq are positional in the flat gradient layout, and that layout is regrouped by make_reducer(mode=...)
    / set_bucket_layers; a same-size buffer used to be kept as is -- Adam state silently permuted against the parameters.  Step, change
    the layout, step again: every parameter must equal torch.optim.Adam's fed with the same gradients; the state dict carries the names
    and a state saved under one layout loads under another."""
    import test_model_gpu as tm
    from gaviko_amd.optim import FusedAdamOneCycle
    from gaviko_amd.utils import synth
    m, cfg = tm.build("gaviko", "vit-t16", dict(tm.GAVIKO), dev)
    named = dict(m.named_parameters())
    eng = m._engine()
    tr = eng.trainable_names()
    ref_p = {n: torch.nn.Parameter(named[n].detach().clone()) for n in tr}
    ref = torch.optim.Adam(list(ref_p.values()), lr=2e-3, eps=1e-8)
    opt = FusedAdamOneCycle(m, lr=2e-3, eps=1e-8, max_norm=None)
    x = torch.from_numpy(synth.volumes(0, 2)).to(dev)
    y = torch.from_numpy(synth.labels(0, 2)).to(dev)

    def one():
        opt.zero_grad()
        torch.nn.functional.cross_entropy(m(x), y).backward()
        for n in tr:
            ref_p[n].grad = named[n].grad.detach().clone()
        ref.step()
        opt.step()
        torch.cuda.synchronize()

    def check(tag):
        for n in tr:
            assert torch.allclose(named[n].detach(), ref_p[n].detach(), rtol=2e-5, atol=2e-7), (tag, n)

    one(); one()
    check("before")
    order0 = list(eng.flat_names())
    sd = opt.state_dict()
    assert sd["names"] == order0 and sum(sd["numels"]) == sd["exp_avg"].numel()
    eng.set_bucket_layers(1)                                            # what make_reducer(layers_per_bucket=1) does to the layout
    assert list(eng.flat_names()) != order0
    one(); one()
    check("after the layout moved")
    # a state saved under the first layout, loaded into a fresh optimizer that binds under the second
    opt2 = FusedAdamOneCycle(m, lr=2e-3, eps=1e-8, max_norm=None)
    opt2.load_state_dict(sd)
    opt2._bind()
    off_new = {}
    o = 0
    for n in eng.flat_names():
        off_new[n] = o
        o += named[n].numel()
    o = 0
    for n, k in zip(sd["names"], sd["numels"]):
        assert torch.equal(opt2.m[off_new[n]: off_new[n] + k], sd["exp_avg"][o: o + k]), n
        o += k
    # a state that covers other tensors than the model trains is an error, not a silent reuse
    bad = dict(sd, names=["nope"] + sd["names"][1:])
    opt3 = FusedAdamOneCycle(m, lr=2e-3, eps=1e-8, max_norm=None)
    opt3.load_state_dict(bad)
    with pytest.raises(Exception, match="covers other tensors"):
        opt3._bind()
