"""Evaluation side of eval.py:97-153 with the per-sample work on the device.

`Evaluator` accumulates logits/labels of the validation loop without host reads; `compute()` runs `gvk_eval_rows` (softmax,
argmax, K x K confusion counts) and `gvk_ovr_auc_counts` (exact one-vs-rest pair counts) and finishes the three numbers the
reference logs -- accuracy, quadratic-weighted Cohen kappa, macro one-vs-rest ROC AUC (`sklearn.metrics` calls of eval.py:120-122)
-- in float64 on the host from K*K + 3K integers.  `write_eval_outputs` reproduces the files of eval.py:127-153.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops


def kappa_quadratic(confusion: np.ndarray) -> float:
    """sklearn.metrics.cohen_kappa_score(weights='quadratic') from a confusion matrix.  sklearn builds its label set from the labels
    that OCCUR (in y_true or y_pred), so absent classes are squeezed out before the (i - j)^2 weights are laid down."""
    c = np.asarray(confusion, dtype=np.float64)
    present = (c.sum(0) + c.sum(1)) > 0
    c = c[present][:, present]
    n = c.shape[0]
    if n < 2:
        return float("nan")                              # sklearn: 0/0 with a RuntimeWarning
    expected = np.outer(c.sum(1), c.sum(0)) / c.sum()
    idx = np.arange(n)
    w = (idx[:, None] - idx[None, :]).astype(np.float64) ** 2
    return float(1.0 - (w * c).sum() / (w * expected).sum())


def macro_ovr_auc(counts: np.ndarray) -> float:
    """roc_auc_score(multi_class='ovr', average='macro') from per-class {2*greater + ties, n_pos, n_neg}; sklearn raises when a class
    has no positive (or no negative) sample, and so does this."""
    counts = np.asarray(counts, dtype=np.float64).reshape(-1, 3)
    if (counts[:, 1] == 0).any() or (counts[:, 2] == 0).any():
        raise ValueError("Only one class present in y_true for at least one one-vs-rest problem. ROC AUC score is not defined in that case.")
    return float((counts[:, 0] / (2.0 * counts[:, 1] * counts[:, 2])).mean())


class Evaluator:
    def __init__(self, num_classes: int, device):
        self.K, self.device = num_classes, device
        self._logits: List[torch.Tensor] = []
        self._labels: List[torch.Tensor] = []

    def update(self, outputs: torch.Tensor, labels: torch.Tensor) -> None:      # eval.py:112-116 without the three .cpu() copies
        self._logits.append(outputs.detach().float())
        self._labels.append(labels.detach().to(torch.int64))

    def compute(self) -> Dict[str, object]:
        logits = torch.cat(self._logits).contiguous()
        labels = torch.cat(self._labels).contiguous().to(self.device)
        N, K = logits.shape
        proba = torch.empty_like(logits)
        pred = torch.empty(N, dtype=torch.int32, device=self.device)
        confusion = torch.zeros(K * K, dtype=torch.int64, device=self.device)
        counts = torch.zeros(3 * K, dtype=torch.int64, device=self.device)
        ops.eval_rows(logits, labels, proba, pred, confusion)
        ops.ovr_auc_counts(proba, labels, counts)
        conf = confusion.cpu().numpy().reshape(K, K)
        try:
            auc: Optional[float] = macro_ovr_auc(counts.cpu().numpy())
        except ValueError:
            auc = None
        return {"accuracy": float(np.trace(conf)) / float(N), "quadratic_kappa": kappa_quadratic(conf), "auc": auc, "confusion": conf,
                "y_pred": pred.cpu().numpy(), "y_pred_proba": proba.cpu().numpy(), "y_test": labels.cpu().numpy()}


def _versioned_csv(results_dir: str, method: str, backbone: str, kind: str, mri_paths, y_pred):
    os.makedirs(results_dir, exist_ok=True)
    version = 1
    bb = backbone.replace("-", "_")
    while True:
        name = f"{method}_{bb}_{kind}_results_v{version}.csv"
        path = os.path.join(results_dir, name)
        if not os.path.exists(path):
            break
        version += 1
    with open(path, "w") as f:
        f.write("mri_path,outputs\n")
        for p, y in zip(mri_paths, y_pred):
            f.write(f"{os.path.basename(str(p))},{int(y)}\n")
    return path, name


def write_inference_outputs(results_dir: str, method: str, backbone: str, mri_paths, y_pred) -> str:
    """inference.py:117-139: '<method>_<backbone>_inference_results_v<n>.csv' (first free n), columns mri_path (basename), outputs."""
    return _versioned_csv(results_dir, method, backbone, "inference", mri_paths, y_pred)[0]


@torch.no_grad()
def predict(model, loader, transforms=None, device=None) -> np.ndarray:
    """The loop of inference.py:100-113: argmax class of every volume a CustomDatasetPrediction loader yields (one host copy at the end)."""
    device = device or next(model.parameters()).device
    model.eval()
    preds = []
    for inputs in loader:
        x = inputs.to(device)
        preds.append(torch.argmax(model(transforms(x) if transforms is not None else x), dim=1))
    return torch.cat(preds).cpu().numpy()


def write_eval_outputs(results_dir: str, method: str, backbone: str, mri_paths, y_pred, acc, qwk, auc) -> str:
    """eval.py:127-153: '<method>_<backbone>_eval_results_v<n>.csv' (first free n) with columns mri_path (basename), outputs; and the
    '_metrics.txt' next to it."""
    path, name = _versioned_csv(results_dir, method, backbone, "eval", mri_paths, y_pred)
    with open(os.path.join(results_dir, name.replace(".csv", "") + "_metrics.txt"), "w") as f:
        f.write(f"Test Accuracy: {acc}\n")
        f.write(f"Test Quadratic Kappa: {qwk}\n")
        f.write(f"Test AUC: {auc}\n")
    return path
