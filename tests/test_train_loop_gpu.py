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
    assert min(e["train_loss"] for e in h[1:]) < h[0]["train_loss"], h       # it learns something on 8 volumes
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
