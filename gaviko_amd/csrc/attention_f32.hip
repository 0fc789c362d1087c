// fp32 multi-head self-attention (head dim 64) for the reference's fp32 configurations (BASELINE cfg4, tolerance 1e-5):
// same interface and data layout as the bf16 kernels (qkv [B*T][3*H*64] in, ctx [B*T][H*64] out, lse saved), plain fp32
// VALU arithmetic, flash-style (no [T][T] score matrix in memory).  One thread owns one query row (forward, dQ) or one key
// row (dK, dV) with that row in registers; the other side streams through LDS in 32-row tiles read as broadcasts.  This path
// exists for parity, not for the headline metric (fp32 has no fast matrix path on CDNA4: 157 TFLOP/s peak, 1/16 of bf16).
#include "common.hpp"
#include "dropout.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

// attention-probability dropout (thresh = 0: off): same mask function as the bf16 kernels, element (b*H + h, query, key)
struct AttnDropF { unsigned long long seed; const unsigned long long* seed_ptr; unsigned int thresh; float inv_keep; };
__device__ __forceinline__ unsigned int f32_akey(const AttnDropF& dr, int bh) { return dr.thresh != 0u ? attn_key(dr.seed + *dr.seed_ptr, bh) : 0u; }
__device__ __forceinline__ float f32_mask(const AttnDropF& dr, unsigned int akey, int q, int k, int T) {
  return dr.thresh != 0u ? attn_drop_scale(akey, (unsigned int)q * (unsigned int)T + (unsigned int)k, dr.thresh, dr.inv_keep) : 1.f;
}

constexpr int kRowsPerWG = 128;   // one row per thread
constexpr int kTile = 32;         // rows of the streamed operand per LDS tile

// stage rows [r0, r0+32) x 64 floats of `src` (row stride ld) into dst[32][64]; rows >= T are zero
__device__ __forceinline__ void stage32(float* dst, const float* __restrict__ src, int r0, int T, int ld) {
  for (int i = threadIdx.x; i < kTile * 16; i += kRowsPerWG) {
    const int r = i >> 4, c = (i & 15) * 4;
    const f32x4 v = (r0 + r < T) ? *(const f32x4*)(src + (size_t)(r0 + r) * ld + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    *(f32x4*)(dst + r * 64 + c) = v;
  }
}

__device__ __forceinline__ float dot64(const float (&a)[64], const float* __restrict__ b) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int d = 0; d < 64; d += 4) {
    const f32x4 v = *(const f32x4*)(b + d);
    s0 = __builtin_fmaf(a[d], v[0], s0); s1 = __builtin_fmaf(a[d + 1], v[1], s1);
    s2 = __builtin_fmaf(a[d + 2], v[2], s2); s3 = __builtin_fmaf(a[d + 3], v[3], s3);
  }
  return (s0 + s1) + (s2 + s3);
}
__device__ __forceinline__ void axpy64(float (&y)[64], float a, const float* __restrict__ x) {
#pragma unroll
  for (int d = 0; d < 64; d += 4) {
    const f32x4 v = *(const f32x4*)(x + d);
    y[d] = __builtin_fmaf(a, v[0], y[d]); y[d + 1] = __builtin_fmaf(a, v[1], y[d + 1]);
    y[d + 2] = __builtin_fmaf(a, v[2], y[d + 2]); y[d + 3] = __builtin_fmaf(a, v[3], y[d + 3]);
  }
}
__device__ __forceinline__ void load64(float (&r)[64], const float* __restrict__ src) {
#pragma unroll
  for (int d = 0; d < 64; d += 4) {
    const f32x4 v = *(const f32x4*)(src + d);
    r[d] = v[0]; r[d + 1] = v[1]; r[d + 2] = v[2]; r[d + 3] = v[3];
  }
}
__device__ __forceinline__ void store64(float* __restrict__ dst, const float (&r)[64], float a) {
#pragma unroll
  for (int d = 0; d < 64; d += 4) *(f32x4*)(dst + d) = f32x4{r[d] * a, r[d + 1] * a, r[d + 2] * a, r[d + 3] * a};
}

// ---- forward: vision_transformer.py:63-71
__global__ __launch_bounds__(kRowsPerWG) void attn_f32_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse,
                                                                  int T, int H, int ld_qkv, int ld_out, float scale, AttnDropF dr) {
  __shared__ __attribute__((aligned(16))) float sK[kTile * 64], sV[kTile * 64];
  const int b = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * kRowsPerWG + threadIdx.x;
  const int inner = H * 64;
  const float* base = qkv + (size_t)b * T * ld_qkv + h * 64;
  const unsigned int akey = f32_akey(dr, b * H + h);
  float qr[64], o[64];
  load64(qr, base + (size_t)min(q, T - 1) * ld_qkv);
#pragma unroll
  for (int d = 0; d < 64; ++d) o[d] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < T; k0 += kTile) {
    __syncthreads();
    stage32(sK, base + inner, k0, T, ld_qkv);
    stage32(sV, base + 2 * inner, k0, T, ld_qkv);
    __syncthreads();
    const int nk = min(kTile, T - k0);
    for (int j = 0; j < nk; ++j) {
      const float s = dot64(qr, sK + j * 64) * scale;
      const float mn = fmaxf(m, s);
      const float sc = __expf(m - mn), e = __expf(s - mn);
      l = l * sc + e;
#pragma unroll
      for (int d = 0; d < 64; ++d) o[d] *= sc;
      axpy64(o, e * f32_mask(dr, akey, q, k0 + j, T), sV + j * 64);        // the softmax statistics stay those of the undropped scores
      m = mn;
    }
  }
  if (q < T) {
    store64(out + ((size_t)b * T + q) * ld_out + h * 64, o, 1.0f / l);
    if (lse != nullptr) lse[((size_t)b * H + h) * T + q] = m + __logf(l);
  }
}

// ---- backward, query side: delta_i = dO_i . O_i ; dQ_i = scale * sum_j P_ij (dO_i . V_j - delta_i) K_j
__global__ __launch_bounds__(kRowsPerWG) void attn_f32_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                                     const float* __restrict__ dout, const float* __restrict__ lse,
                                                                     float* __restrict__ delta, float* __restrict__ dqkv, int T, int H,
                                                                     int ld_qkv, int ld_out, float scale, AttnDropF dr) {
  __shared__ __attribute__((aligned(16))) float sK[kTile * 64], sV[kTile * 64];
  const int b = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * kRowsPerWG + threadIdx.x;
  const int qc = min(q, T - 1);
  const int inner = H * 64;
  const float* base = qkv + (size_t)b * T * ld_qkv + h * 64;
  float qr[64], dor[64], dq[64];
  load64(qr, base + (size_t)qc * ld_qkv);
  load64(dor, dout + ((size_t)b * T + qc) * ld_out + h * 64);
  const float dl = dot64(dor, o + ((size_t)b * T + qc) * ld_out + h * 64);
  const float ls = lse[((size_t)b * H + h) * T + qc];
  const unsigned int akey = f32_akey(dr, b * H + h);
  if (q < T) delta[((size_t)b * H + h) * T + q] = dl;
#pragma unroll
  for (int d = 0; d < 64; ++d) dq[d] = 0.f;
  for (int k0 = 0; k0 < T; k0 += kTile) {
    __syncthreads();
    stage32(sK, base + inner, k0, T, ld_qkv);
    stage32(sV, base + 2 * inner, k0, T, ld_qkv);
    __syncthreads();
    const int nk = min(kTile, T - k0);
    for (int j = 0; j < nk; ++j) {
      const float pr = __expf(dot64(qr, sK + j * 64) * scale - ls);
      const float ds = pr * (f32_mask(dr, akey, q, k0 + j, T) * dot64(dor, sV + j * 64) - dl);
      axpy64(dq, ds, sK + j * 64);
    }
  }
  if (q < T) store64(dqkv + ((size_t)b * T + q) * ld_qkv + h * 64, dq, scale);
}

// ---- backward, key side.  WHICH = 0: dV_j = sum_i P_ij dO_i ;  WHICH = 1: dK_j = scale * sum_i P_ij (dO_i . V_j - delta_i) Q_i
template <int WHICH>
__global__ __launch_bounds__(kRowsPerWG) void attn_f32_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                     const float* __restrict__ lse, const float* __restrict__ delta,
                                                                     float* __restrict__ dqkv, int T, int H, int ld_qkv, int ld_out, float scale,
                                                                     AttnDropF dr) {
  __shared__ __attribute__((aligned(16))) float sQ[kTile * 64], sD[kTile * 64];
  __shared__ float sL[kTile], sDl[kTile];
  const int b = blockIdx.z, h = blockIdx.y;
  const int k = blockIdx.x * kRowsPerWG + threadIdx.x;
  const int kc = min(k, T - 1);
  const int inner = H * 64;
  const float* base = qkv + (size_t)b * T * ld_qkv + h * 64;
  const unsigned int akey = f32_akey(dr, b * H + h);
  float kr[64], acc[64];
  float vr[WHICH == 1 ? 64 : 1];
  load64(kr, base + inner + (size_t)kc * ld_qkv);
  if constexpr (WHICH == 1) load64(vr, base + 2 * inner + (size_t)kc * ld_qkv);
#pragma unroll
  for (int d = 0; d < 64; ++d) acc[d] = 0.f;
  for (int q0 = 0; q0 < T; q0 += kTile) {
    __syncthreads();
    stage32(sQ, base, q0, T, ld_qkv);
    stage32(sD, dout + (size_t)b * T * ld_out + h * 64, q0, T, ld_out);
    if (threadIdx.x < kTile) {
      const int qi = min(q0 + (int)threadIdx.x, T - 1);
      sL[threadIdx.x] = lse[((size_t)b * H + h) * T + qi];
      sDl[threadIdx.x] = delta[((size_t)b * H + h) * T + qi];
    }
    __syncthreads();
    const int nq = min(kTile, T - q0);
    for (int i = 0; i < nq; ++i) {
      const float pr = __expf(dot64(kr, sQ + i * 64) * scale - sL[i]);
      const float mm = f32_mask(dr, akey, q0 + i, k, T);
      if constexpr (WHICH == 0) {
        axpy64(acc, pr * mm, sD + i * 64);
      } else {
        const float ds = pr * (mm * dot64(vr, sD + i * 64) - sDl[i]);
        axpy64(acc, ds, sQ + i * 64);
      }
    }
  }
  if (k < T) store64(dqkv + ((size_t)b * T + k) * ld_qkv + (WHICH == 0 ? 2 : 1) * inner + h * 64, acc, WHICH == 0 ? 1.0f : scale);
}

}  // namespace gvk

extern "C" int gvk_attention_fwd_f32_dropout(const float* qkv, float* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale,
                                             float drop_p, uint64_t seed, const void* seed_ptr, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed_ptr != nullptr), "gvk_attention_fwd_f32: drop_p in [0,1) and a seed word");
  GVK_REQUIRE(drop_p == 0.f || (int64_t)T * T < (int64_t)1 << 32, "gvk_attention_fwd_f32: the dropout mask index (query*T + key) is 32-bit");
  const AttnDropF dr{seed, (const unsigned long long*)seed_ptr, drop_threshold_u32(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f};
  GVK_REQUIRE(qkv && out, "gvk_attention_fwd_f32: null pointer");
  GVK_REQUIRE(B > 0 && T > 0 && H > 0, "gvk_attention_fwd_f32: empty shape");
  GVK_REQUIRE(ld_qkv >= 3 * H * 64 && ld_qkv % 4 == 0 && ld_out >= H * 64 && ld_out % 4 == 0,
              "gvk_attention_fwd_f32: head dim is fixed at 64; ld_qkv=%d ld_out=%d inconsistent with H=%d", ld_qkv, ld_out, H);
  GVK_LAUNCH(attn_f32_fwd_kernel, dim3((T + kRowsPerWG - 1) / kRowsPerWG, H, B), dim3(kRowsPerWG), 0, (hipStream_t)stream, qkv, out, lse, T, H,
             ld_qkv, ld_out, scale, dr);
  return check_launch("attention_fwd_f32");
}

extern "C" int gvk_attention_fwd_f32(const float* qkv, float* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale,
                                     void* stream) {
  return gvk_attention_fwd_f32_dropout(qkv, out, lse, B, T, H, ld_qkv, ld_out, scale, 0.f, 0, nullptr, stream);
}

extern "C" int gvk_attention_bwd_f32_dropout(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv,
                                             int B, int T, int H, int ld_qkv, int ld_out, float scale, float drop_p, uint64_t seed,
                                             const void* seed_ptr, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed_ptr != nullptr), "gvk_attention_bwd_f32: drop_p in [0,1) and a seed word");
  GVK_REQUIRE(drop_p == 0.f || (int64_t)T * T < (int64_t)1 << 32, "gvk_attention_bwd_f32: the dropout mask index (query*T + key) is 32-bit");
  const AttnDropF dr{seed, (const unsigned long long*)seed_ptr, drop_threshold_u32(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f};
  GVK_REQUIRE(qkv && out && dout && lse && delta && dqkv, "gvk_attention_bwd_f32: null pointer");
  GVK_REQUIRE(B > 0 && T > 0 && H > 0, "gvk_attention_bwd_f32: empty shape");
  GVK_REQUIRE(ld_qkv >= 3 * H * 64 && ld_qkv % 4 == 0 && ld_out >= H * 64 && ld_out % 4 == 0,
              "gvk_attention_bwd_f32: head dim is fixed at 64; ld_qkv=%d ld_out=%d inconsistent with H=%d", ld_qkv, ld_out, H);
  const dim3 grid((T + kRowsPerWG - 1) / kRowsPerWG, H, B), block(kRowsPerWG);
  hipStream_t s = (hipStream_t)stream;
  GVK_LAUNCH(attn_f32_bwd_dq_kernel, grid, block, 0, s, qkv, out, dout, lse, delta, dqkv, T, H, ld_qkv, ld_out, scale, dr);
  int rc = check_launch("attention_bwd_f32/dq");
  if (rc) return rc;
  GVK_LAUNCH((attn_f32_bwd_kv_kernel<0>), grid, block, 0, s, qkv, dout, lse, (const float*)delta, dqkv, T, H, ld_qkv, ld_out, scale, dr);
  rc = check_launch("attention_bwd_f32/dv");
  if (rc) return rc;
  GVK_LAUNCH((attn_f32_bwd_kv_kernel<1>), grid, block, 0, s, qkv, dout, lse, (const float*)delta, dqkv, T, H, ld_qkv, ld_out, scale, dr);
  return check_launch("attention_bwd_f32/dk");
}

extern "C" int gvk_attention_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv,
                                     int B, int T, int H, int ld_qkv, int ld_out, float scale, void* stream) {
  return gvk_attention_bwd_f32_dropout(qkv, out, dout, lse, delta, dqkv, B, T, H, ld_qkv, ld_out, scale, 0.f, 0, nullptr, stream);
}
