"""Fused loss seeds (csrc/loss.hip via gaviko_amd.losses) against the reference's FocalLoss fixtures and the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    from gaviko_amd import lib
    lib.require_device()
    return torch.device("cuda:0")


def test_focal_loss_matches_reference_fixture(dev):
    from gaviko_amd.losses import FocalLoss
    g = np.load(os.path.join(GOLD, "focal_loss.npz"))
    lg = torch.from_numpy(g["logits"]).to(dev).requires_grad_(True)
    loss = FocalLoss(gamma=1.2)(lg, torch.from_numpy(g["target"]).to(dev))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-6
    assert np.abs(lg.grad.cpu().numpy() - g["grad"]).max() < 2e-6
    outside = (g["logits"] <= 0) | (g["logits"] >= 1)          # the clamp of focal_loss.py:86 kills these gradients exactly
    assert (lg.grad.cpu().numpy()[outside] == 0).all()


@pytest.mark.parametrize("red", ["mean", "sum"])
def test_focal_loss_weights_ignore_index_reductions(dev, red):
    from gaviko_amd.losses import FocalLoss
    g = np.load(os.path.join(GOLD, "focal_loss_weighted.npz"))
    lg = torch.from_numpy(g["logits"]).to(dev).requires_grad_(True)
    crit = FocalLoss(gamma=1.2, weights=torch.from_numpy(g["weights"]), reduction=red)
    loss = crit(lg, torch.from_numpy(g["target"]).to(dev))
    (2.5 * loss).backward()                                      # a non-unit upstream gradient
    want = float(g["loss_" + red])
    assert abs(loss.item() - want) < 2e-6 * max(1.0, want)
    assert np.abs(lg.grad.cpu().numpy() - 2.5 * g["grad_" + red]).max() < 5e-6
    assert (lg.grad.cpu().numpy()[g["target"] == -100] == 0).all()


@pytest.mark.parametrize("B,K", [(4, 5), (32, 5), (300, 7), (2, 2)])
def test_cross_entropy_and_meter_vs_oracle(dev, B, K):
    from gaviko_amd.losses import CrossEntropyLoss, StepMeter
    gen = torch.Generator().manual_seed(B * 31 + K)
    x = torch.randn(B, K, generator=gen) * 3
    y = torch.randint(0, K, (B,), generator=gen)
    if B > 4:
        y[1] = -100
    w = torch.rand(K, generator=gen) + 0.5
    for weight in (None, w):
        xo = x.clone().requires_grad_(True)
        lo = oracle.cross_entropy(xo, y, weight=weight)
        lo.backward()
        meter = StepMeter(dev)
        crit = CrossEntropyLoss(weight=weight).attach_meter(meter)
        xg = x.to(dev).requires_grad_(True)
        for _ in range(2):                                      # two steps accumulate in the meter
            xg.grad = None
            lg = crit(xg, y.to(dev))
            lg.backward()
        assert abs(lg.item() - lo.item()) < 1e-5 * max(1.0, abs(lo.item()))
        assert (xg.grad.cpu() - xo.grad).abs().max().item() < 1e-6
        mean_loss, acc, n = meter.read()
        ok = y != -100
        correct = ((x.argmax(-1) == y) & ok).sum().item()
        assert n == 2 * B and abs(mean_loss - lo.item()) < 1e-5 * max(1.0, abs(lo.item())) and abs(acc - correct / B) < 1e-6


def test_focal_loss_vs_oracle_random(dev):
    from gaviko_amd.losses import FocalLoss
    gen = torch.Generator().manual_seed(7)
    x = torch.rand(64, 5, generator=gen) * 1.6 - 0.3             # most logits inside (0,1), some outside on either side
    y = torch.randint(0, 5, (64,), generator=gen)
    for gamma in (0.5, 1.2, 2.0):
        xo = x.clone().requires_grad_(True)
        lo = oracle.focal_loss(xo, y, gamma=gamma)
        lo.backward()
        xg = x.to(dev).requires_grad_(True)
        lg = FocalLoss(gamma=gamma)(xg, y.to(dev))
        lg.backward()
        assert abs(lg.item() - lo.item()) < 2e-6
        assert (xg.grad.cpu() - xo.grad).abs().max().item() < 2e-6


def test_loss_rejects_what_is_not_built(dev):
    from gaviko_amd.losses import FocalLoss
    with pytest.raises(NotImplementedError):
        FocalLoss(gamma=1.2, reduction="median")
    with pytest.raises(NotImplementedError):
        FocalLoss(gamma=1.2)(torch.zeros(4, device=dev), torch.zeros(4, dtype=torch.int64, device=dev))


@pytest.mark.parametrize("kind", ["focal", "ce"])
def test_loss_reduction_none_vs_oracle(dev, kind):
    """reduction='none' (focal_loss.py:40,117-118 accepts it): the per-sample vector, ignored rows zero, and the backward of an arbitrary
    per-sample upstream gradient -- against the oracle (itself checked against the reference class in tests/test_oracle_vs_golden.py)."""
    from gaviko_amd.losses import CrossEntropyLoss, FocalLoss
    gen = torch.Generator().manual_seed(11)
    x = torch.rand(40, 5, generator=gen) * 1.6 - 0.3
    y = torch.randint(0, 5, (40,), generator=gen)
    y[3] = y[17] = -100
    w = torch.tensor([0.5, 1.0, 2.0, 1.5, 0.25])
    up = torch.rand(40, generator=gen) + 0.5
    xo = x.clone().requires_grad_(True)
    lo = (oracle.focal_loss(xo, y, gamma=1.2, weights=w, reduction="none") if kind == "focal"
          else oracle.cross_entropy(xo, y, weight=w, reduction="none"))
    (lo * up).sum().backward()
    xg = x.to(dev).requires_grad_(True)
    crit = FocalLoss(gamma=1.2, weights=w, reduction="none") if kind == "focal" else CrossEntropyLoss(weight=w, reduction="none")
    lg = crit(xg, y.to(dev))
    assert lg.shape == (40,) and float(lg[3]) == 0.0 and float(lg[17]) == 0.0
    (lg * up.to(dev)).sum().backward()
    assert (lg.detach().cpu() - lo.detach()).abs().max().item() < 5e-6
    assert (xg.grad.cpu() - xo.grad).abs().max().item() < 5e-6


def test_focal_loss_reduction_none_matches_reference_fixture(dev):
    from gaviko_amd.losses import FocalLoss
    g = np.load(os.path.join(GOLD, "focal_loss_weighted.npz"))
    lg = torch.from_numpy(g["logits"]).to(dev).requires_grad_(True)
    vec = FocalLoss(gamma=1.2, weights=torch.from_numpy(g["weights"]), reduction="none")(lg, torch.from_numpy(g["target"]).to(dev))
    (vec * torch.from_numpy(g["up"]).to(dev)).sum().backward()
    assert np.abs(vec.detach().cpu().numpy() - g["loss_none"]).max() < 2e-6
    assert np.abs(lg.grad.cpu().numpy() - g["grad_none"]).max() < 5e-6
