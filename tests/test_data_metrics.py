"""Data side + evaluation metrics.  CPU part: host logic, the numpy oracle against scipy / sklearn (the reference's own
metric dependency, eval.py:120-122).  GPU part: csrc/augment.hip and csrc/metrics.hip through the C-ABI against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import data_ref

RNG = np.random.default_rng(0)


def _vol(shape=(24, 32, 40), seed=0):
    r = np.random.default_rng(seed)
    return (r.standard_normal(shape) * 300 + 1000).astype(np.float32)


# ------------------------------------------------------------------------------------------------ CPU: oracle and host logic
def test_oracle_rescale_and_flip_properties():
    v = _vol()
    y = data_ref.rescale_intensity(v)
    assert y.dtype == np.float32 and y.min() == 0.0 and y.max() == 1.0
    assert np.array_equal(data_ref.rescale_intensity(np.full((4, 4, 4), 7, np.float32)), np.full((4, 4, 4), 7, np.float32))   # constant volume: unchanged
    assert np.array_equal(data_ref.flip(data_ref.flip(v, 5), 5), v)
    assert np.array_equal(data_ref.flip(v, 1), v[::-1])


def test_oracle_affine_matches_scipy_grid_constant():
    from scipy import ndimage
    from gaviko_amd.data import affine_matrix
    v = _vol((20, 24, 28), 3)
    m = affine_matrix((1.05, 0.93, 1.08), (11.0, -7.0, 14.0), (0.0, 0.0, 0.0), v.shape)
    ours = data_ref.affine_resample(v, m, float(v.min()))
    ref = ndimage.affine_transform(v.astype(np.float64), m[:, :3], offset=m[:, 3], order=1, mode="grid-constant", cval=float(v.min()))
    assert np.abs(ours - ref).max() < 2e-3 * np.abs(v).max() * 1e-1          # float32 coordinates vs float64
    ident = affine_matrix((1, 1, 1), (0, 0, 0), (0, 0, 0), v.shape)
    assert np.allclose(ident, np.eye(3, 4)) and np.array_equal(data_ref.affine_resample(v, ident, 0.0), v)
    # a rotation by 90 degrees about axis 0 of a cube maps the grid onto itself: exact permutation of the voxels
    c = _vol((16, 16, 16), 5)
    r90 = data_ref.affine_resample(c, affine_matrix((1, 1, 1), (90, 0, 0), (0, 0, 0), c.shape), 0.0)
    assert np.abs(r90 - np.rot90(c, k=-1, axes=(1, 2))).max() < 1e-2 or np.abs(r90 - np.rot90(c, k=1, axes=(1, 2))).max() < 1e-2


def test_transform_sampling_is_torchio_parameterisation():
    from gaviko_amd import data
    tf = data.train_transforms(seed=123)
    assert [type(t).__name__ for t in tf.affine + tf.flips + tf.rescale] == ["RandomAffine", "RandomFlip", "RescaleIntensity"]
    n_aff = n_flip = 0
    for _ in range(400):
        mats, flags = tf.sample(1, (120, 160, 160))
        bits, aff = tf.last_params[0]
        n_flip += bits & 1
        assert bits & 6 == 0                                              # only axis 0 is ever flipped (train.py:41)
        if aff is not None:
            n_aff += 1
            s, d, t = aff
            assert ((0.9 <= s) & (s <= 1.1)).all() and (np.abs(d) <= 15).all() and (t == 0).all() and flags[0] & 8
        else:
            assert np.array_equal(mats[0], np.eye(3, 4, dtype=np.float32))
    assert 150 < n_aff < 250 and 150 < n_flip < 250                       # p = 0.5 each
    with pytest.raises(NotImplementedError):
        data.RandomAffine(default_pad_value="mean")
    with pytest.raises(RuntimeError, match="GPU"):
        tf(torch.zeros(1, 1, 4, 4, 4))


def test_dataset_reads_npz_like_the_reference(tmp_path):
    import pandas as pd
    from gaviko_amd import data
    rows = []
    for i, sub in enumerate(["train", "train", "val", "test"]):
        v = _vol((6, 8, 8), i)
        np.savez(tmp_path / f"s{i}.npz", data=v)
        rows.append(dict(mri_path=f"s{i}.npz", kl_grade=i % 5, subset=sub))
    df = pd.DataFrame(rows)
    ds = data.CustomDataset(df, image_folder=str(tmp_path))
    x, y = ds[1]
    assert x.shape == (1, 6, 8, 8) and x.dtype == torch.float32 and y == 1 and np.array_equal(x[0].numpy(), _vol((6, 8, 8), 1)) and len(ds) == 4
    host = data.CustomDataset(df, transforms=lambda a: data_ref.rescale_intensity(a), image_folder=str(tmp_path))   # host-side callable, as in the reference
    assert float(host[0][0].max()) == 1.0
    df2 = df.assign(mri_path=[str(tmp_path / p) for p in df["mri_path"]])
    assert data.CustomDatasetPrediction(df2)[2].shape == (1, 6, 8, 8)
    csv = tmp_path / "all.csv"
    df.to_csv(csv, index=False)
    pre = data.DataPreprocessor({"data": dict(data_path=str(csv), image_folder=str(tmp_path), batch_size=2, num_workers=0)}, seed=0)
    tl, vl, sl, tds, vds, sds = pre.preprocess(None)
    assert (len(tds), len(vds), len(sds)) == (2, 1, 1)
    xb, yb = next(iter(vl))
    assert xb.shape == (1, 1, 6, 8, 8) and yb.tolist() == [2]


def test_kappa_and_auc_host_math_match_sklearn():
    from sklearn.metrics import cohen_kappa_score, confusion_matrix, roc_auc_score
    from gaviko_amd import metrics
    r = np.random.default_rng(4)
    for K, N, absent in ((5, 300, None), (5, 120, 2), (3, 50, None)):
        y = r.integers(0, K, N)
        p = r.integers(0, K, N)
        if absent is not None:                                            # a class that never occurs: sklearn squeezes it out of the weights
            y[y == absent] = 0
            p[p == absent] = 0
        conf = confusion_matrix(y, p, labels=list(range(K)))
        assert abs(metrics.kappa_quadratic(conf) - cohen_kappa_score(y, p, weights="quadratic")) < 1e-12
    y = r.integers(0, 5, 400)
    s = r.random((400, 5))
    s[::7] = s[3]                                                         # ties
    s = s / s.sum(1, keepdims=True)
    counts = []
    for c in range(5):
        pos, neg = s[y == c, c], s[y != c, c]
        counts.append([2 * (pos[:, None] > neg[None, :]).sum() + (pos[:, None] == neg[None, :]).sum(), len(pos), len(neg)])
    want = roc_auc_score(y, s, multi_class="ovr", average="macro")
    assert abs(metrics.macro_ovr_auc(np.array(counts)) - want) < 1e-12
    with pytest.raises(ValueError):
        metrics.macro_ovr_auc(np.array([[0, 0, 10], [5, 5, 5]]))


def test_eval_outputs_files(tmp_path):
    from gaviko_amd import metrics
    a = metrics.write_eval_outputs(str(tmp_path), "gaviko", "vit-b16", ["/x/a.npz", "/y/b.npz"], [3, 0], 0.5, 0.25, 0.75)
    b = metrics.write_eval_outputs(str(tmp_path), "gaviko", "vit-b16", ["/x/a.npz"], [1], 1.0, 1.0, None)
    assert os.path.basename(a) == "gaviko_vit_b16_eval_results_v1.csv" and os.path.basename(b) == "gaviko_vit_b16_eval_results_v2.csv"   # eval.py:137-144
    assert open(a).read() == "mri_path,outputs\na.npz,3\nb.npz,0\n"
    assert open(str(tmp_path / "gaviko_vit_b16_eval_results_v1_metrics.txt")).read() == "Test Accuracy: 0.5\nTest Quadratic Kappa: 0.25\nTest AUC: 0.75\n"
    c = metrics.write_inference_outputs(str(tmp_path), "gaviko", "vit-b16", ["/x/a.npz", "b.npz"], [4, 2])          # inference.py:125-139
    assert os.path.basename(c) == "gaviko_vit_b16_inference_results_v1.csv" and open(c).read() == "mri_path,outputs\na.npz,4\nb.npz,2\n"


# ------------------------------------------------------------------------------------------------ GPU: kernels vs the oracle
@pytest.fixture(scope="module")
def dev():
    from gaviko_amd import lib
    lib.require_device()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(24, 32, 40), (120, 160, 160)])
def test_rescale_intensity_bit_exact(dev, shape):
    from gaviko_amd import data
    vols = np.stack([_vol(shape, s) for s in range(3)] + [np.full(shape, 5, np.float32)])[:, None]
    vols[1] *= -1
    y = data.eval_transforms()(torch.from_numpy(vols).to(dev)).cpu().numpy()
    for b in range(4):
        assert np.array_equal(y[b, 0], data_ref.rescale_intensity(vols[b, 0])), b
    assert y[0].min() == 0.0 and y[0].max() == 1.0 and (y[3] == 5).all()


@pytest.mark.gpu
def test_spatial_transform_vs_oracle(dev):
    from gaviko_amd import data, ops
    shape = (24, 32, 40)
    B = 6
    vols = np.stack([_vol(shape, 10 + s) for s in range(B)])
    mats = np.tile(np.eye(3, 4, dtype=np.float32), (B, 1, 1))
    flags = np.array([0, 1, 5, 8, 9, 8 | 6], dtype=np.int32)
    par = [((1.05, 0.93, 1.08), (11.0, -7.0, 14.0)), ((0.9, 1.1, 1.0), (-15.0, 15.0, 3.0)), ((1.0, 1.0, 1.0), (0.0, 0.0, 12.0))]
    for b, (s, d) in zip((3, 4, 5), par):
        mats[b] = data.affine_matrix(s, d, (0, 0, 0), shape).astype(np.float32)
    x = torch.from_numpy(vols).to(dev)
    part = ops.minmax_partials(B, dev)
    ops.volume_minmax(x, part)
    out = torch.empty_like(x)
    ops.spatial_transform(x, out, torch.from_numpy(mats).to(dev), torch.from_numpy(flags).to(dev), part)
    out = out.cpu().numpy()
    for b in range(B):
        want = data_ref.spatial(vols[b], mats[b] if flags[b] & 8 else None, int(flags[b] & 7))
        if flags[b] & 8:
            assert np.abs(out[b] - want).max() < 1e-3 * np.abs(vols[b]).max(), b      # float32 interpolation, possibly different summation order
        else:
            assert np.array_equal(out[b], want), b                                    # copies and flips are exact gathers


@pytest.mark.gpu
def test_train_transforms_pipeline(dev):
    from gaviko_amd import data
    tf = data.train_transforms(seed=5)
    vols = np.stack([_vol((24, 32, 40), 20 + s) for s in range(8)])[:, None]
    y = tf(torch.from_numpy(vols).to(dev)).cpu().numpy()
    for b, (bits, aff) in enumerate(tf.last_params):
        mat = None if aff is None else data.affine_matrix(*aff, vols.shape[-3:]).astype(np.float32)
        want = data_ref.rescale_intensity(data_ref.spatial(vols[b, 0], mat, bits))
        assert np.abs(y[b, 0] - want).max() < (2e-3 if aff is not None else 0.0) + 1e-12, (b, bits)
        assert y[b].min() == 0.0 and y[b].max() == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("N,K", [(257, 5), (1500, 5), (40, 3)])
def test_eval_metrics_match_sklearn(dev, N, K):
    from sklearn.metrics import accuracy_score, cohen_kappa_score, roc_auc_score
    from gaviko_amd.metrics import Evaluator
    g = torch.Generator().manual_seed(N)
    y = torch.randint(0, K, (N,), generator=g)
    logits = torch.randn(N, K, generator=g) + 2.0 * torch.nn.functional.one_hot(y, K) * (torch.rand(N, 1, generator=g) > 0.4)
    logits[::9] = logits[4]                                               # ties in the scores
    ev = Evaluator(K, dev)
    for a in range(0, N, 64):                                             # eval.py:106-116 feeds batches
        ev.update(logits[a:a + 64].to(dev), y[a:a + 64].to(dev))
    r = ev.compute()
    pred = logits.argmax(1).numpy()
    assert np.array_equal(r["y_pred"], pred) and np.array_equal(r["y_test"], y.numpy())
    assert abs(r["accuracy"] - accuracy_score(y.numpy(), pred)) < 1e-12
    assert abs(r["quadratic_kappa"] - cohen_kappa_score(y.numpy(), pred, weights="quadratic")) < 1e-12
    proba = r["y_pred_proba"].astype(np.float64)
    assert np.abs(proba - logits.softmax(1).numpy()).max() < 1e-6
    want = roc_auc_score(y.numpy(), proba, multi_class="ovr", average="macro")      # rows sum to 1 within float32 rounding: sklearn accepts
    assert abs(r["auc"] - want) < 1e-9
