// fp32 multi-head self-attention (head dim 64) for the reference's fp32 configurations (BASELINE cfg4, tolerance 1e-5):
// same interface and data layout as the bf16 kernels (qkv [B*T][3*H*64] in, ctx [B*T][H*64] out, lse saved), plain fp32
// VALU arithmetic, flash-style (no [T][T] score matrix in memory).  One thread owns one query row (forward, dQ) or one key
// row (dK, dV) with that row in registers; the other side streams through LDS in 32-row tiles read as broadcasts.  This path
// exists for parity, not for the headline metric (fp32 has no fast matrix path on CDNA4: 157 TFLOP/s peak, 1/16 of bf16).
#include "common.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

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
                                                                  int T, int H, int ld_qkv, int ld_out, float scale) {
  __shared__ __attribute__((aligned(16))) float sK[kTile * 64], sV[kTile * 64];
  const int b = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * kRowsPerWG + threadIdx.x;
  const int inner = H * 64;
  const float* base = qkv + (size_t)b * T * ld_qkv + h * 64;
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
      axpy64(o, e, sV + j * 64);
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
                                                                     int ld_qkv, int ld_out, float scale) {
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
      const float ds = pr * (dot64(dor, sV + j * 64) - dl);
      axpy64(dq, ds, sK + j * 64);
    }
  }
  if (q < T) store64(dqkv + ((size_t)b * T + q) * ld_qkv + h * 64, dq, scale);
}

// ---- backward, key side.  WHICH = 0: dV_j = sum_i P_ij dO_i ;  WHICH = 1: dK_j = scale * sum_i P_ij (dO_i . V_j - delta_i) Q_i
template <int WHICH>
__global__ __launch_bounds__(kRowsPerWG) void attn_f32_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                     const float* __restrict__ lse, const float* __restrict__ delta,
                                                                     float* __restrict__ dqkv, int T, int H, int ld_qkv, int ld_out, float scale) {
  __shared__ __attribute__((aligned(16))) float sQ[kTile * 64], sD[kTile * 64];
  __shared__ float sL[kTile], sDl[kTile];
  const int b = blockIdx.z, h = blockIdx.y;
  const int k = blockIdx.x * kRowsPerWG + threadIdx.x;
  const int kc = min(k, T - 1);
  const int inner = H * 64;
  const float* base = qkv + (size_t)b * T * ld_qkv + h * 64;
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
      if constexpr (WHICH == 0) {
        axpy64(acc, pr, sD + i * 64);
      } else {
        const float ds = pr * (dot64(vr, sD + i * 64) - sDl[i]);
        axpy64(acc, ds, sQ + i * 64);
      }
    }
  }
  if (k < T) store64(dqkv + ((size_t)b * T + k) * ld_qkv + (WHICH == 0 ? 2 : 1) * inner + h * 64, acc, WHICH == 0 ? 1.0f : scale);
}

}  // namespace gvk

extern "C" int gvk_attention_fwd_f32(const float* qkv, float* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale,
                                     void* stream) {
  using namespace gvk;
  GVK_REQUIRE(qkv && out, "gvk_attention_fwd_f32: null pointer");
  GVK_REQUIRE(B > 0 && T > 0 && H > 0, "gvk_attention_fwd_f32: empty shape");
  GVK_REQUIRE(ld_qkv >= 3 * H * 64 && ld_qkv % 4 == 0 && ld_out >= H * 64 && ld_out % 4 == 0,
              "gvk_attention_fwd_f32: head dim is fixed at 64; ld_qkv=%d ld_out=%d inconsistent with H=%d", ld_qkv, ld_out, H);
  GVK_LAUNCH(attn_f32_fwd_kernel, dim3((T + kRowsPerWG - 1) / kRowsPerWG, H, B), dim3(kRowsPerWG), 0, (hipStream_t)stream, qkv, out, lse, T, H,
             ld_qkv, ld_out, scale);
  return check_launch("attention_fwd_f32");
}

extern "C" int gvk_attention_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv,
                                     int B, int T, int H, int ld_qkv, int ld_out, float scale, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(qkv && out && dout && lse && delta && dqkv, "gvk_attention_bwd_f32: null pointer");
  GVK_REQUIRE(B > 0 && T > 0 && H > 0, "gvk_attention_bwd_f32: empty shape");
  GVK_REQUIRE(ld_qkv >= 3 * H * 64 && ld_qkv % 4 == 0 && ld_out >= H * 64 && ld_out % 4 == 0,
              "gvk_attention_bwd_f32: head dim is fixed at 64; ld_qkv=%d ld_out=%d inconsistent with H=%d", ld_qkv, ld_out, H);
  const dim3 grid((T + kRowsPerWG - 1) / kRowsPerWG, H, B), block(kRowsPerWG);
  hipStream_t s = (hipStream_t)stream;
  GVK_LAUNCH(attn_f32_bwd_dq_kernel, grid, block, 0, s, qkv, out, dout, lse, delta, dqkv, T, H, ld_qkv, ld_out, scale);
  int rc = check_launch("attention_bwd_f32/dq");
  if (rc) return rc;
  GVK_LAUNCH((attn_f32_bwd_kv_kernel<0>), grid, block, 0, s, qkv, dout, lse, (const float*)delta, dqkv, T, H, ld_qkv, ld_out, scale);
  rc = check_launch("attention_bwd_f32/dv");
  if (rc) return rc;
  GVK_LAUNCH((attn_f32_bwd_kv_kernel<1>), grid, block, 0, s, qkv, dout, lse, (const float*)delta, dqkv, T, H, ld_qkv, ld_out, scale);
  return check_launch("attention_bwd_f32/dk");
}
