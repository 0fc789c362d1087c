"""ORACLE -- test infrastructure only.  numpy restatement of the torchio transforms train.py:38-62 applies (torchio==0.20.16,
requirements.txt:6; NOT installed here, so this follows its published algorithm and parity with torchio itself is unpinned) and of
the resampling convention gaviko_amd/data.py documents."""
from __future__ import annotations

import numpy as np


def rescale_intensity(vol: np.ndarray, out_min: float = 0.0, out_max: float = 1.0) -> np.ndarray:
    """torchio RescaleIntensity.rescale with percentiles=(0,100), no mask: float32 array; clip to (min,max) is a no-op;
    array -= in_min; array /= in_range; array *= out_range; array += out_min; unchanged (with a warning) when in_range == 0."""
    a = np.array(vol, dtype=np.float32, copy=True)
    in_min, in_max = a.min(), a.max()
    in_range = in_max - in_min
    if in_range == 0:
        return a
    a -= in_min
    a /= in_range
    a *= np.float32(out_max - out_min)
    a += np.float32(out_min)
    return a


def flip(vol: np.ndarray, bits: int) -> np.ndarray:
    """torchio RandomFlip on the spatial axes of a (D,H,W) volume: bit k mirrors axis k."""
    for k in range(3):
        if bits & (1 << k):
            vol = np.flip(vol, axis=k)
    return np.ascontiguousarray(vol)


def affine_resample(vol: np.ndarray, mat: np.ndarray, pad: float) -> np.ndarray:
    """out[q] = trilinear sample of vol at p = A q + t (mat = [A | t], float32 arithmetic in the kernel's order); neighbours outside
    the volume contribute `pad`."""
    D, H, W = vol.shape
    m = mat.astype(np.float32)
    qz, qy, qx = np.meshgrid(np.arange(D, dtype=np.float32), np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    coords = [((m[r, 0] * qz + m[r, 1] * qy) + m[r, 2] * qx) + m[r, 3] for r in range(3)]
    f = [np.floor(c) for c in coords]
    i = [x.astype(np.int64) for x in f]
    w = [c - x for c, x in zip(coords, f)]
    out = np.zeros(vol.shape, dtype=np.float32)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                zz, yy, xx = i[0] + dz, i[1] + dy, i[2] + dx
                ok = (zz >= 0) & (zz < D) & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
                v = np.where(ok, vol[np.clip(zz, 0, D - 1), np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], np.float32(pad)).astype(np.float32)
                wt = ((w[0] if dz else 1 - w[0]) * (w[1] if dy else 1 - w[1])) * (w[2] if dx else 1 - w[2])
                out += (wt * v).astype(np.float32)
    return out


def spatial(vol: np.ndarray, mat, bits: int) -> np.ndarray:
    """affine (when mat is not None) then flips -- the order DeviceCompose uses."""
    if mat is not None:
        vol = affine_resample(vol, mat, float(vol.min()))
    return flip(vol, bits)
