"""Data side of train.py:33-78 / data/dataset.py with the transforms on the device.

The reference reads one `.npz['data']` (120,160,160) per sample, adds the channel axis and runs torchio transforms on the CPU in the
DataLoader workers (train: RandomAffine(degrees=15, p=.5) + RandomFlip(axes=(0), p=.5), then RescaleIntensity((0,1)); val/test:
RescaleIntensity only).  Here the datasets return the RAW volume (the npz read stays on the host, in workers, into pinned memory)
and the transforms run on the whole batch once it is in HBM, as three HBM-bound kernels (`csrc/augment.hip`):

    x = batch.to(device, non_blocking=True)          # [B, 1, D, H, W] raw
    x = pre.train_transforms(x)                      # affine + flip in one resampling pass, then min/max + rescale

torchio 0.20.16 (requirements.txt:6) and its SimpleITK backend are not installed here, so:
* RescaleIntensity and RandomFlip follow torchio's published arithmetic exactly (float32 steps in the same order; exact gathers);
* RandomAffine keeps torchio's parameterisation (per-axis scale ~ U(1-s, 1+s) with s = 0.1 by default, per-axis angle ~ U(-d, d),
  rotation about the image centre, linear interpolation, padding with the volume minimum) but its axis/sign conventions are this
  module's own (rotation R = R2.R1.R0 about the array axes), not SimpleITK's LPS ones: same augmentation distribution up to
  axis naming, not the same voxels for the same seed.  Parity vs torchio is UNPINNED; the kernels are pinned against
  `oracle/data_ref.py` (numpy) and scipy.ndimage.affine_transform.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import ops


# ---- datasets: data/dataset.py:5-49 -------------------------------------------------------------------------------------------
class CustomDataset(Dataset):
    """dataset.py:5-28.  `transforms` (optional) is a host-side callable on the (1,D,H,W) array, as in the reference; leave it None to
    get the raw float32 volume and apply `DeviceCompose` to the batch on the GPU."""

    def __init__(self, dataframe, transforms=None, image_folder=None):
        self.df = dataframe
        self.transforms = transforms
        self.image_folder = image_folder

    def _path(self, index):
        p = self.df.loc[index]["mri_path"]
        return p if not self.image_folder else os.path.join(self.image_folder, p)

    def _volume(self, index):
        mri_object = np.load(self._path(index))["data"]
        mri_object = np.expand_dims(mri_object, 0)                       # (1 x 120 x 160 x 160)
        if self.transforms is not None:
            mri_object = self.transforms(mri_object)
        return torch.as_tensor(np.ascontiguousarray(mri_object, dtype=np.float32))

    def __getitem__(self, index):
        return self._volume(index), self.df.loc[index]["kl_grade"]

    def __len__(self):
        return len(self.df)


class CustomDatasetPrediction(CustomDataset):
    """dataset.py:30-49: no image_folder, no label."""

    def __init__(self, dataframe, transforms=None):
        super().__init__(dataframe, transforms, None)

    def __getitem__(self, index):
        return self._volume(index)


# ---- device transforms ----------------------------------------------------------------------------------------------------------
def _pair(v, name):
    if isinstance(v, (int, float)):
        return (-float(v), float(v))
    v = tuple(float(t) for t in v)
    if len(v) != 2:
        raise ValueError(f"{name}: a number or a (low, high) pair")
    return v


class RandomFlip:
    """tio.RandomFlip(axes, flip_probability): every listed spatial axis is mirrored independently with that probability."""

    def __init__(self, axes=0, flip_probability: float = 0.5):
        axes = (axes,) if isinstance(axes, int) else tuple(axes)
        if any(a not in (0, 1, 2) for a in axes):
            raise ValueError("RandomFlip: axes are spatial axes 0, 1, 2")
        self.axes, self.p = axes, float(flip_probability)

    def sample(self, rng: np.random.Generator) -> int:
        bits = 0
        for a in self.axes:
            if rng.random() < self.p:
                bits |= 1 << a
        return bits


class RandomAffine:
    """tio.RandomAffine(scales=0.1, degrees=10, translation=0, isotropic=False, center='image', default_pad_value='minimum',
    image_interpolation='linear', p=1) -- the subset train.py:40 uses, torchio's parameter sampling."""

    def __init__(self, scales=0.1, degrees=10, translation=0, isotropic: bool = False, center: str = "image", default_pad_value="minimum",
                 image_interpolation: str = "linear", p: float = 1.0):
        if center != "image" or default_pad_value != "minimum" or image_interpolation != "linear":
            raise NotImplementedError("RandomAffine: only center='image', default_pad_value='minimum', image_interpolation='linear' are built")
        self.scales = (1.0 - float(scales), 1.0 + float(scales)) if isinstance(scales, (int, float)) else tuple(float(s) for s in scales)
        self.degrees, self.translation = _pair(degrees, "degrees"), _pair(translation, "translation")
        self.isotropic, self.p = bool(isotropic), float(p)

    def sample(self, rng: np.random.Generator):
        """None (not applied) or (scales[3], degrees[3], translation[3])."""
        if rng.random() >= self.p:
            return None
        s = rng.uniform(self.scales[0], self.scales[1], 3)
        if self.isotropic:
            s[:] = s[0]
        return s, rng.uniform(self.degrees[0], self.degrees[1], 3), rng.uniform(self.translation[0], self.translation[1], 3)


def affine_matrix(scales, degrees, translation, shape) -> np.ndarray:
    """3x4 map from an OUTPUT voxel q to the input position p = A q + t for the forward transform T(p) = c + R S (p - c) + tr about the
    image centre c, R = R2 R1 R0 (Rk: rotation by degrees[k] about array axis k), S = diag(scales): A = S^-1 R^T, t = c - A (c + tr)."""
    a = np.radians(np.asarray(degrees, dtype=np.float64))
    c0, s0, c1, s1, c2, s2 = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
    r0 = np.array([[1, 0, 0], [0, c0, -s0], [0, s0, c0]])
    r1 = np.array([[c1, 0, s1], [0, 1, 0], [-s1, 0, c1]])
    r2 = np.array([[c2, -s2, 0], [s2, c2, 0], [0, 0, 1]])
    A = np.diag(1.0 / np.asarray(scales, dtype=np.float64)) @ (r2 @ r1 @ r0).T
    c = (np.asarray(shape, dtype=np.float64) - 1.0) / 2.0
    t = c - A @ (c + np.asarray(translation, dtype=np.float64))
    return np.concatenate([A, t[:, None]], axis=1)


class RescaleIntensity:
    """tio.RescaleIntensity(out_min_max): per-volume min-max to [out_min, out_max] (percentiles (0,100), no mask)."""

    def __init__(self, out_min_max=(0, 1)):
        self.out_min, self.out_max = float(out_min_max[0]), float(out_min_max[1])


class DeviceCompose:
    """The transforms of train.py:38-62 on a batch [B, 1, D, H, W] (or [B, D, H, W]) already on the GPU.  Spatial transforms are merged
    into one resampling pass (affine first, flips on its output), RescaleIntensity runs last, as in the reference's Compose."""

    def __init__(self, transforms: Sequence, seed: Optional[int] = None):
        self.affine = [t for t in transforms if isinstance(t, RandomAffine)]
        self.flips = [t for t in transforms if isinstance(t, RandomFlip)]
        self.rescale = [t for t in transforms if isinstance(t, RescaleIntensity)]
        if len(self.affine) > 1 or len(self.rescale) > 1 or len(self.affine) + len(self.flips) + len(self.rescale) != len(transforms):
            raise NotImplementedError("DeviceCompose: RandomAffine (at most one), RandomFlip, RescaleIntensity (at most one)")
        self.rng = np.random.default_rng(seed)
        self.last_params = None                                           # [(flip bits, affine params or None)] of the last call: tests, logging

    def sample(self, B: int, shape):
        mats = np.tile(np.eye(3, 4, dtype=np.float32), (B, 1, 1))
        flags = np.zeros(B, dtype=np.int32)
        params = []
        for b in range(B):
            aff = self.affine[0].sample(self.rng) if self.affine else None
            bits = 0
            for f in self.flips:
                bits ^= f.sample(self.rng)
            if aff is not None:
                mats[b] = affine_matrix(*aff, shape).astype(np.float32)
                bits |= 8
            flags[b] = bits
            params.append((bits & 7, aff))
        self.last_params = params
        return mats, flags

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("DeviceCompose runs on the GPU: move the batch first (there is no CPU path)")
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        B, shape = x.shape[0], tuple(x.shape[-3:])
        part = ops.minmax_partials(B, x.device)
        if self.affine or self.flips:
            mats, flags = self.sample(B, shape)
            if flags.any():
                ops.volume_minmax(x, part)                                # pad value of the resampling = the input volume's minimum
                out = torch.empty_like(x)
                ops.spatial_transform(x, out, torch.from_numpy(mats).to(x.device), torch.from_numpy(flags).to(x.device), part)
                x = out
        if self.rescale:
            ops.volume_minmax(x, part)
            y = torch.empty_like(x)
            ops.rescale_intensity(x, part, y, self.rescale[0].out_min, self.rescale[0].out_max)
            x = y
        return x


def train_transforms(seed: Optional[int] = None) -> DeviceCompose:
    """train.py:38-52 (the intensity_augment dict of 43-48 is dead code there: its OneOf line is commented out)."""
    return DeviceCompose([RandomAffine(degrees=15, p=0.5), RandomFlip(axes=(0,), flip_probability=0.5), RescaleIntensity((0, 1))], seed)


def eval_transforms() -> DeviceCompose:
    """train.py:54-60: val and test."""
    return DeviceCompose([RescaleIntensity((0, 1))])


class DataPreprocessor:
    """train.py:33-78: same CSV columns (`subset`, `mri_path`, `kl_grade`), same loaders and return tuple; the transforms are the
    device-side `train_transforms` / `val_transforms` attributes to call on each batch after `.to(device)`."""

    def __init__(self, config, seed: Optional[int] = None):
        self.config = config
        self.train_transforms, self.val_transforms, self.test_transforms = train_transforms(seed), eval_transforms(), eval_transforms()

    def preprocess(self, df=None):
        import pandas as pd
        d = self.config["data"]
        df = pd.read_csv(d["data_path"])
        sub = {s: df[df["subset"] == s].reset_index(drop=True) for s in ("train", "val", "test")}
        ds = {s: CustomDataset(sub[s], transforms=None, image_folder=d["image_folder"]) for s in sub}
        mk = lambda s, shuffle: DataLoader(ds[s], batch_size=d["batch_size"], shuffle=shuffle, num_workers=d["num_workers"], pin_memory=True)  # noqa: E731
        return mk("train", True), mk("val", False), mk("test", False), ds["train"], ds["val"], ds["test"]
