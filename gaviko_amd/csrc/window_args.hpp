// Shared by the MWSA window-attention kernels (window_attn.hip: one wave per token row, any L; window_mfma.hip: 16x16 tiles on the
// fp32 matrix cores, L = 20).
#pragma once
#include "common.hpp"

namespace gvk {

__device__ __forceinline__ unsigned int hash_u32_w(unsigned long long seed, unsigned long long idx) {
  unsigned long long x = idx * 0x9E3779B97F4A7C15ull + seed;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  return (unsigned int)x;
}

struct WinArgs {
  const float* qkv;   // [B*N][3L]
  float* ctx;         // [B*N][L]
  float* lse;         // [B*N]
  const float* dctx;  // bwd
  float* delta;       // bwd scratch [B*N]
  float* dqkv;        // bwd out [B*N][3L]
  int B, D, H, W, kd, kh, kw;
  float scale;
  unsigned long long seed; const unsigned long long* seed_ptr; unsigned int drop_thresh; float inv_keep;
  int kv_side;        // window_mfma.hip backward: 0 = the query side (dq, delta), 1 = the key side (dk, dv)
};

struct Win {   // forward window of a query / reverse window of a key, per axis [lo, lo+n)
  int d0, nd, h0, nh, w0, nw;
  __device__ __forceinline__ int count() const { return nd * nh * nw; }
  __device__ __forceinline__ int index(int kk, int H, int W) const {
    const int ww = kk % nw, t = kk / nw, hh = t % nh, dd = t / nh;
    return ((d0 + dd) * H + (h0 + hh)) * W + (w0 + ww);
  }
};
__device__ __forceinline__ void axis_fwd(int q, int k, int n, int& lo, int& cnt) {
  const int a = max(0, q - k / 2), b = min(n, q - k / 2 + k);
  lo = a; cnt = b - a;
}
__device__ __forceinline__ void axis_rev(int key, int k, int n, int& lo, int& cnt) {
  // queries q with  q - k/2 <= key < q - k/2 + k   <=>   key - (k - 1 - k/2) <= q <= key + k/2
  const int a = max(0, key - (k - 1 - k / 2)), b = min(n - 1, key + k / 2);
  lo = a; cnt = b - a + 1;
}


// L = 20 on the fp32 matrix cores (window_mfma.hip); return 1 = not covered, fall back to the row-per-wave kernels
int launch_win_mfma_fwd(const WinArgs& a, int L, hipStream_t s);
int launch_win_mfma_bwd(const WinArgs& a, int L, hipStream_t s);

}  // namespace gvk
