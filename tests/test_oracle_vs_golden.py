"""CPU: the oracle (oracle/*.py) reproduces every golden fixture generated from the imported reference
(tools/gen_golden.py) -- this is what pins the oracle on machines where /root/reference does not exist."""
import ast

import numpy as np
import pytest
import torch

import oracle
from conftest import golden
from gaviko_amd.utils import synth

FAST = ["cfg1_linear_t16_b1", "gaviko_t16_b2", "gaviko_t16_b2_unfrozen", "gaviko_t16_b2_k366_p8", "gaviko_t16_b1_share2", "gaviko_t16_b2_lat16", "deep_vpt_t16_b2", "shallow_vpt_t16_b2", "deep_vpt_t16_b2_unfrozen", "shallow_vpt_t16_b2_unfrozen",
        "adaptformer_t16_b2", "adaptformer_t16_b2_unfrozen", "melo_t16_b2", "melo_t16_b2_layers", "ssf_t16_b2", "ssf_t16_b2_unfrozen", "dvpt_t16_b2", "dvpt_t16_b2_unfrozen", "dvpt_t16_b2_mean_p8", "evp_t16_b2", "evp_t16_b2_unfrozen", "bitfit_t16_b2", "fft_t16_b2"]


def _cfg(g):
    cfg = ast.literal_eval(str(g["meta/cfg"]))
    return cfg, str(g["meta/method"]), int(g["meta/batch"])


def _sample_rows(T):
    return sorted(set(r for r in (0, 1, 7, 8, 9, 31, 32, 33, 34, 66, 500, T - 1) if r < T))


def _tap(t):
    C = t.shape[2]
    cols = list(range(0, C, max(1, C // 32)))[:32]
    return t.detach()[:, _sample_rows(t.shape[1])][:, :, cols].numpy()


@pytest.mark.parametrize("name", FAST)
def test_oracle_forward_backward_matches_golden(name):
    g = golden(name)
    cfg, method, B = _cfg(g)
    shapes = oracle.SHAPES[method](cfg)
    sd = {k: torch.from_numpy(v).requires_grad_(oracle.trainable(method, k, cfg)) for k, v in synth.fill_state_dict(shapes).items()}
    x = torch.from_numpy(synth.volumes(0, B))
    y = torch.from_numpy(synth.labels(0, B))
    taps = {}
    logits = oracle.FORWARD[method](sd, x, cfg, taps)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    assert np.abs(logits.detach().numpy() - g["logits"]).max() < 2e-5
    assert (logits.argmax(-1).numpy() == g["argmax"]).all()
    assert abs(loss.item() - float(g["loss_ce"])) < 1e-5
    assert abs(oracle.focal_loss(logits.detach(), y).item() - float(g["loss_focal"])) < 1e-5
    trainable = sorted(k for k, v in sd.items() if v.requires_grad)
    assert trainable == sorted(str(s) for s in g["meta/trainable"])
    for k in g.files:
        if k.startswith("gradnorm/"):
            want = float(g[k])
            assert abs(sd[k[9:]].grad.norm().item() - want) <= 1e-4 * max(want, 1e-6) + 1e-9, k
        elif k.startswith("grad/"):
            want = g[k]
            assert np.abs(sd[k[5:]].grad.numpy() - want).max() <= 2e-5 * max(1e-6, np.abs(want).max()) + 1e-9, k
        elif k.startswith("tap/") and k[4:] in taps:
            want = g[k]
            assert np.abs(_tap(taps[k[4:]]) - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), k


def test_oracle_cfg2_forward_matches_golden():
    g = golden("cfg2_gaviko_b16_b4")
    cfg, method, B = _cfg(g)
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(oracle.SHAPES[method](cfg)).items()}
    with torch.no_grad():
        logits = oracle.gaviko_forward(sd, torch.from_numpy(synth.volumes(0, B)), cfg)
    assert np.abs(logits.numpy() - g["logits"]).max() < 3e-5
    assert (logits.argmax(-1).numpy() == g["argmax"]).all()


def test_focal_loss_quirk_matches_reference_fixture():
    g = golden("focal_loss")
    lg = torch.from_numpy(g["logits"]).requires_grad_(True)
    loss = oracle.focal_loss(lg, torch.from_numpy(g["target"]))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    assert np.abs(lg.grad.numpy() - g["grad"]).max() < 1e-6
    # gradient is exactly zero for logits outside (0,1): the clamp of focal_loss.py:86 kills it
    outside = (g["logits"] <= 0) | (g["logits"] >= 1)
    assert (g["grad"][outside] == 0).all()


@pytest.mark.parametrize("red", ["mean", "sum"])
def test_focal_loss_weights_and_ignored_rows_match_reference_fixture(red):
    g = golden("focal_loss_weighted")
    lg = torch.from_numpy(g["logits"]).requires_grad_(True)
    loss = oracle.focal_loss(lg, torch.from_numpy(g["target"]), weights=torch.from_numpy(g["weights"]), reduction=red)
    loss.backward()
    assert abs(loss.item() - float(g["loss_" + red])) < 1e-5 * max(1.0, float(g["loss_" + red]))
    assert np.abs(lg.grad.numpy() - g["grad_" + red]).max() < 1e-6
    assert (lg.grad.numpy()[g["target"] == -100] == 0).all()


def test_focal_loss_reduction_none_fixture():
    """reduction='none' of the reference class (focal_loss.py:40,117-118): per-sample vector (ignored rows 0) and the gradient of its
    `up`-weighted sum."""
    g = golden("focal_loss_weighted")
    lg = torch.from_numpy(g["logits"]).requires_grad_(True)
    vec = oracle.focal_loss(lg, torch.from_numpy(g["target"]), weights=torch.from_numpy(g["weights"]), reduction="none")
    (vec * torch.from_numpy(g["up"])).sum().backward()
    assert vec.shape == (16,) and np.abs(vec.detach().numpy() - g["loss_none"]).max() < 1e-6
    assert (vec.detach().numpy()[g["target"] == -100] == 0).all()
    assert np.abs(lg.grad.numpy() - g["grad_none"]).max() < 1e-6


@pytest.mark.parametrize("lk", [(6, 6, 6), (3, 6, 6), (3, 3, 3)])
def test_window_mask_matches_reference_mask(lk):
    g = golden(f"mwsa_mask_{lk[0]}{lk[1]}{lk[2]}")
    allow = np.unpackbits(g["allow"], axis=1)[:, :1000].astype(bool)
    m = oracle.window_mask((10, 10, 10), lk)
    assert ((m == 0).numpy() == allow).all()
    assert (allow.sum(1) == g["count"]).all()


def test_cfg3_data_parallel_fixture_is_mean_of_shards():
    """The cfg3 fixture holds 32 per-sample logits and the 8x4 mean-reduced gradients (pins DDP equivalence)."""
    g = golden("cfg3_deep_vpt_b16_8x4")
    assert g["logits"].shape == (32, 5) and int(g["meta/shards"]) == 8
    assert sum(1 for k in g.files if k.startswith("grad/")) == 5
