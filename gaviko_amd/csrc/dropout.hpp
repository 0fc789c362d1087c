// Counter-based dropout masks shared by every kernel that drops something.  A mask is a pure function of (seed word, element index),
// so the backward regenerates the forward's mask instead of storing it, and a test can rebuild it on the host (tests/dropmask.py).
// The effective seed is  site seed + *seed_ptr : the device word is bumped once per forward by a recorded kernel, so replayed plans
// draw fresh masks.
#pragma once
#include "common.hpp"

namespace gvk {

// element-indexed sites (activations [M][N]): keep iff hash >= p * 2^32
__device__ __forceinline__ unsigned int hash_u32(unsigned long long seed, unsigned long long idx) {
  unsigned long long x = idx * 0x9E3779B97F4A7C15ull + seed;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  return (unsigned int)x;
}
__device__ __forceinline__ float drop_scale(unsigned long long seed, unsigned long long idx, unsigned int thresh, float inv_keep) {
  return (hash_u32(seed, idx) >= thresh) ? inv_keep : 0.f;
}

// attention probabilities [B*H][T][T]: 51 M elements per layer and three kernels regenerate them, so the hash is 32-bit
// (murmur3 finaliser over  (i*T + j) * golden + key(seed, bh)): 2 multiplies instead of the 64-bit pair above.
__device__ __forceinline__ unsigned int attn_key(unsigned long long seed, int bh) {
  return (unsigned int)(seed ^ (seed >> 32)) + (unsigned int)bh * 0x85EBCA77u;
}
__device__ __forceinline__ unsigned int hash_attn(unsigned int key, unsigned int ij) {
  unsigned int x = ij * 0x9E3779B1u + key;
  x ^= x >> 16; x *= 0x85EBCA6Bu;
  x ^= x >> 13; x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float attn_drop_scale(unsigned int key, unsigned int ij, unsigned int thresh, float inv_keep) {
  return (hash_attn(key, ij) >= thresh) ? inv_keep : 0.f;
}

// p -> threshold on the 32-bit hash (host side)
static inline unsigned int drop_threshold_u32(float p) {
  if (p <= 0.f) return 0u;
  double t = (double)p * 4294967296.0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (unsigned int)t;
}

}  // namespace gvk
