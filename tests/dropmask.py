"""Host-side reproduction of the device dropout masks (gaviko_amd/csrc/dropout.hpp) -- test infrastructure: lets a test run the ORACLE
with exactly the masks the kernels drew."""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def threshold(p: float) -> int:
    if p <= 0:
        return 0
    return int(min(float(np.float32(p)) * 4294967296.0, 4294967295.0))


def inv_keep(p: float) -> np.float32:
    return np.float32(1.0) / (np.float32(1.0) - np.float32(p)) if p > 0 else np.float32(1.0)


def hash_u32(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = idx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        x ^= x >> np.uint64(32); x *= np.uint64(0xD6E8FEB86659FD93)
        x ^= x >> np.uint64(32); x *= np.uint64(0xD6E8FEB86659FD93)
        x ^= x >> np.uint64(32)
    return (x & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def rows_mask(seed: int, M: int, N: int, p: float) -> np.ndarray:
    """scale mask [M, N] of an element-indexed site (index m*N + n): 0 or 1/(1-p)."""
    idx = np.arange(M * N, dtype=np.uint64).reshape(M, N)
    return np.where(hash_u32(seed, idx) >= np.uint32(threshold(p)), inv_keep(p), np.float32(0)).astype(np.float32)


def attn_mask(seed: int, B: int, H: int, T: int, p: float) -> np.ndarray:
    """scale mask [B, H, T, T] of the attention probabilities."""
    seed &= 0xFFFFFFFFFFFFFFFF
    out = np.empty((B * H, T, T), dtype=np.float32)
    ij = (np.arange(T, dtype=np.uint32)[:, None] * np.uint32(T) + np.arange(T, dtype=np.uint32)[None, :])
    with np.errstate(over="ignore"):
        for bh in range(B * H):
            key = np.uint32(((seed ^ (seed >> 32)) + bh * 0x85EBCA77) & 0xFFFFFFFF)
            x = ij * np.uint32(0x9E3779B1) + key
            x ^= x >> np.uint32(16); x *= np.uint32(0x85EBCA6B)
            x ^= x >> np.uint32(13); x *= np.uint32(0xC2B2AE35)
            x ^= x >> np.uint32(16)
            out[bh] = np.where(x >= np.uint32(threshold(p)), inv_keep(p), np.float32(0))
    return out.reshape(B, H, T, T)


def window_attn_mask(seed: int, B: int, N: int, p: float) -> np.ndarray:
    """scale mask [B, N, N] of the MWSA attention probabilities (csrc/window_*.hip: index (b*N + i) * N + j)."""
    idx = np.arange(B * N * N, dtype=np.uint64).reshape(B, N, N)
    return np.where(hash_u32(seed, idx) >= np.uint32(threshold(p)), inv_keep(p), np.float32(0)).astype(np.float32)
