#!/usr/bin/env python3
"""The reference's training entry point (train.py:80-504) condensed onto this package's pieces, end to end on synthetic data:

    .npz volumes + CSV  ->  DataPreprocessor (raw volumes from the workers)          gaviko_amd.data          (train.py:33-78)
    batch.to(device)    ->  RandomAffine + RandomFlip + RescaleIntensity on the GPU  gaviko_amd.data          (train.py:38-62)
    build_model(config['model'])                                                     gaviko_amd.registry      (train.py:111-153)
    FocalLoss(gamma=1.2) with device-side running loss / accuracy                    gaviko_amd.losses        (train.py:176-179,327-328)
    clip_grad_norm_(1.0) + Adam + OneCycleLR as one fused step                       gaviko_amd.optim         (train.py:185-206,315-319)
    validation: accuracy / quadratic kappa / macro OvR AUC                           gaviko_amd.metrics       (eval.py:103-122)
    best-model checkpoint with the trainable tensors only                            gaviko_amd.utils         (train.py:460-485)

usage: python examples/train_synthetic.py [--method gaviko] [--backbone vit-t16] [--epochs 3] [--samples 8] [--out /tmp/gaviko_run]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaviko_amd import data, losses, metrics  # noqa: E402
from gaviko_amd.optim import FusedAdamOneCycle  # noqa: E402
from gaviko_amd.registry import build_model  # noqa: E402
from gaviko_amd.utils import load_pretrained  # noqa: E402


def make_dataset(root, n, num_classes, seed=0):
    """n synthetic (120,160,160) volumes per subset whose class shows in a coarse intensity pattern; CSV with the reference's columns."""
    import pandas as pd
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(seed)
    rows = []
    for subset, count in (("train", n), ("val", max(num_classes, n // 2)), ("test", max(num_classes, n // 2))):
        for i in range(count):
            k = i % num_classes
            v = rng.standard_normal((120, 160, 160)).astype(np.float32) * 40 + 500
            v[k * 20:(k + 1) * 20] += 400.0                              # a bright slab whose depth position is the label
            name = f"{subset}_{i}.npz"
            np.savez(os.path.join(root, name), data=v)
            rows.append(dict(mri_path=name, kl_grade=k, subset=subset))
    csv = os.path.join(root, "data.csv")
    pd.DataFrame(rows).to_csv(csv, index=False)
    return csv


def run(method="gaviko", backbone="vit-t16", epochs=3, samples=8, out="/tmp/gaviko_run", batch_size=4, lr=1e-3, seed=0, log=print):
    dev = torch.device("cuda:0")
    K = 5
    csv = make_dataset(os.path.join(out, "data"), samples, K, seed)
    config = {"data": dict(data_path=csv, image_folder=os.path.join(out, "data"), batch_size=batch_size, num_workers=0),
              "model": dict(method=method, backbone=backbone, image_size=160, image_patch_size=16, frames=120, frame_patch_size=12, num_classes=K,
                            channels=1, pool="cls", dim_head=64, dropout=0.1, emb_dropout=0.1, freeze_vit=True, num_prompts=8, prompt_dim=64,
                            prompt_dropout=0.1, deep_prompt=method == "deep_vpt", prompt_latent_dim=20, local_dim=20, local_k=(6, 6, 6),
                            DHW=(10, 10, 10), attn_drop=0.2, proj_drop=0.2, share_factor=1, r=4, alpha=4),
              "train": dict(save_dir=out, save_threshold=0.0)}
    torch.manual_seed(seed)
    pre = data.DataPreprocessor(config, seed=seed)
    train_loader, val_loader, _, train_ds, val_ds, _ = pre.preprocess(None)
    model = build_model(config["model"]).to(dev)
    tuning_params = load_pretrained.tuning_param_names(model)            # train.py:160-169
    log(f"{method}/{backbone}: {len(tuning_params)} trainable tensors, {len(train_ds)} train / {len(val_ds)} val volumes")
    meter = losses.StepMeter(dev)
    criterion = losses.FocalLoss(gamma=1.2).attach_meter(meter)          # train.py:176-177
    total_steps = epochs * len(train_loader)
    opt = FusedAdamOneCycle(model, lr=lr, eps=1e-8, max_norm=1.0, max_lr=lr, total_steps=total_steps, pct_start=0.3, div_factor=10,
                            final_div_factor=1000)
    history, best_acc, best_path = [], -1.0, None
    for epoch in range(epochs):
        model.train()
        meter.reset()
        for inputs, labels in train_loader:
            inputs = pre.train_transforms(inputs.to(dev, non_blocking=True))
            loss = criterion(model(inputs), labels.to(dev))
            loss.backward()
            opt.step()
            opt.zero_grad()
        train_loss, train_acc, n = meter.read()                         # the epoch's only host read of the training loop
        model.eval()
        ev = metrics.Evaluator(K, dev)
        with torch.no_grad():
            for inputs, labels in val_loader:
                ev.update(model(pre.val_transforms(inputs.to(dev))), labels.to(dev))
        r = ev.compute()
        history.append(dict(epoch=epoch, train_loss=train_loss, train_acc=train_acc, val_acc=r["accuracy"], val_kappa=r["quadratic_kappa"], val_auc=r["auc"]))
        log(f"epoch {epoch}: train loss {train_loss:.4f} acc {train_acc:.3f} | val acc {r['accuracy']:.3f} kappa {r['quadratic_kappa']:.3f} auc {r['auc']}")
        if r["accuracy"] > best_acc:                                      # train.py:460-485
            best_acc = r["accuracy"]
            best_path = load_pretrained.save_trainable(model, out, method, backbone, epoch, best_acc, tuning_params)
    paths = [os.path.join(config["data"]["image_folder"], p) for p in val_ds.df["mri_path"]]
    csv_out = metrics.write_eval_outputs(os.path.join(out, "results"), method, backbone, paths, r["y_pred"], r["accuracy"], r["quadratic_kappa"], r["auc"])
    return dict(history=history, checkpoint=best_path, results_csv=csv_out, model=model, config=config, pre=pre, val_loader=val_loader)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--method", default="gaviko")
    ap.add_argument("--backbone", default="vit-t16")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--out", default="/tmp/gaviko_run")
    a = ap.parse_args()
    res = run(a.method, a.backbone, a.epochs, a.samples, a.out)
    print("checkpoint:", res["checkpoint"])
    print("results   :", res["results_csv"])
