// Shared pieces of the bf16 MFMA GEMM kernels (gemm_bf16.hip: 4-wave tiles; gemm8p_bf16.hip: the 256 x 256 eight-phase kernel):
// argument block, LDS swizzles and the fused epilogue table.  Accumulator convention of both kernels: v_mfma_f32_16x16x32_bf16 with the
// WEIGHT fragment as the A operand, so a lane holds 4 consecutive n of row m = lane & 15 per 16x16 tile; the weight rows of tiles 2p and
// 2p+1 are interleaved so that across the pair a lane owns EIGHT consecutive n (one 16-byte bf16 / 32-byte fp32 piece per store).
#pragma once
#include "common.hpp"
#include "dropout.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

struct GemmArgs {
  const bf16* A;
  const bf16* W;
  void* out0;
  void* out1;
  const float* bias;
  const float* res;
  const bf16* aux;
  const float* pos;
  int M, N, K, lda, ldw, ldo, ldres, ldaux;
  int rows_in, rows_out, row_off;
  int nbm, nbn;
  int group_m;                                           // row panels per rasterisation group (4-wave kernel; set by launch_gemm)
  int xcd_panels;                                        // > 0: panel-major tile order inside each XCD's run (launch_gemm: one-round launches)
  int a_rows;                                            // rows the A buffer really has (M padded to 128): the 256-row tile clamps to it
  int m_stride;                                          // first row of row tile t = t * m_stride (launch_gemm: the tile's rows, or gvk_gemm_desc.m_stride)
  // SPLITK instantiations only (strided row panels): the K loop of a tile is cut into `ksplit` consecutive pieces of `kt_per` k-tiles, one
  // workgroup each; partial tiles meet in `sk_part` (f32, [tile][piece][wave][register][lane][4]) and the workgroup that arrives last at
  // the tile's ticket word sums them in piece order and runs the epilogue
  int ksplit, kt_per;
  float* sk_part;
  int* sk_tick;
  size_t sk_bytes;                                       // (host side: bytes behind gvk_gemm_desc.splitk_ws)
  // DROP instantiations only (nn.Dropout behind a Linear of the unfrozen-backbone methods): mask index m * N + n
  unsigned long long seed; const unsigned long long* seed_ptr; unsigned int drop_thresh; float inv_keep;
  // STORE_BF16 only: columns n < scale_cols leave as (acc + bias) * col_scale (the q block of a qkv projection, pre-scaled for the attention kernels)
  int scale_cols; float col_scale;
  // LayerNorm folded into the consumer GEMM (STORE_BF16): A holds the RAW rows x (bf16), W the rows gamma o W; the epilogue applies
  //   y = rstd[m] * (acc - mean[m] * c1[n]) + bias[n]      c1[n] = sum_c (gamma o W)[n][c] (of the bf16 operand), bias[n] = sum_c beta[c] W[n][c]
  const float* ln_mean; const float* ln_rstd; const float* ln_c1;
  // BIAS_RES_F32_BF16: per-row partial (sum, sum of squares) of the fp32 output over this wave's columns, written to
  // stat_part[(column group)][m] (float2; group = column / (16 * NT)); a later kernel turns the groups of a row into mean / rstd
  float* stat_part;
  const float* stat_pivot;                               // the partials are sums of (x - stat_pivot[m]) and its square (nullptr: pivot 0)
  // BIAS_GELU_BF16: out0 = GELU'(pre) instead of pre;  GELU_BWD_BF16: aux holds that derivative (gvk_gemm_desc.aux_is_grad)
  int aux_grad;
};

// LDS swizzles (applied to the 16-byte chunk index of a 128-byte tile row; conflict-free for the ds_read_b128 lane groups)
__device__ __forceinline__ int swz_a128(int row) { return (row >> 1) & 7; }
// Weight tile: a fragment read touches rows base + 8q + 4b + r (q, r = 0..3, b fixed); the key takes row bits 1, 3, 4
// so those 16 rows again hit 16 distinct (parity, slot) pairs.
__device__ __forceinline__ int swz_w(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }

// Epilogue of one wave: acc[i][j] is the 16x16 tile at rows mbase + 16 i, columns nbase + 32 (j >> 1) (+ the interleave above).
// Loads and stores retire through ONE in-order counter (vmcnt), so a side load issued behind the previous row's stores would wait for
// them: the column-only bias is fetched once, and the per-row operands (residual / GELU' input / position rows) of row i+1 are requested
// before row i is stored.
template <int EPI, bool DROP, int MT, int NT>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[MT][NT], const int mbase, const int nbase, const int l15, const int lq) {
  static_assert(NT % 2 == 0, "tile pairs");
  constexpr int NP = NT / 2;
  constexpr bool kRes = EPI == GVK_EPI_BIAS_RES_F32 || EPI == GVK_EPI_BIAS_RES_F32_BF16;
  constexpr bool kAux = EPI == GVK_EPI_GELU_BWD_BF16 || EPI == GVK_EPI_RELU_BWD_BF16;
  constexpr bool kPos = EPI == GVK_EPI_PATCH_F32;
  constexpr bool kFold = EPI == GVK_EPI_STORE_BF16;          // optional LayerNorm fold (p.ln_mean != nullptr)
  constexpr bool kStat = EPI == GVK_EPI_BIAS_RES_F32_BF16;   // optional row-statistic partials (p.stat_part != nullptr)
  f32x4 bv[NP][2];
  [[maybe_unused]] f32x4 c1v[NP][2];
#pragma unroll
  for (int jp = 0; jp < NP; ++jp) {
    bv[jp][0] = bv[jp][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int n = nbase + 32 * jp + 8 * lq;
    if (p.bias != nullptr) {
      bv[jp][0] = *(const f32x4*)(p.bias + n);
      bv[jp][1] = *(const f32x4*)(p.bias + n + 4);
    }
    if constexpr (kFold) {
      c1v[jp][0] = c1v[jp][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.ln_mean != nullptr) {
        c1v[jp][0] = *(const f32x4*)(p.ln_c1 + n);
        c1v[jp][1] = *(const f32x4*)(p.ln_c1 + n + 4);
      }
    }
  }
  struct Side { f32x4 r0[NP], r1[NP]; bf16x8 a8[NP]; float mu, rs, pv; };
  auto fetch = [&](const int i, Side& sd) {
    const int m = mbase + i * 16 + l15;
    if (m >= p.M) return;
    if constexpr (kFold) {
      if (p.ln_mean != nullptr) { sd.mu = p.ln_mean[m]; sd.rs = p.ln_rstd[m]; }
    }
    if constexpr (kStat) sd.pv = p.stat_pivot != nullptr ? p.stat_pivot[m] : 0.f;
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      const int n = nbase + 32 * jp + 8 * lq;
      if constexpr (kRes) {
        const float* rp = p.res + (size_t)m * p.ldres + n;
        sd.r0[jp] = *(const f32x4*)rp;
        sd.r1[jp] = *(const f32x4*)(rp + 4);
      } else if constexpr (kPos) {
        const float* pp = p.pos + (size_t)(m % p.rows_in) * p.N + n;
        sd.r0[jp] = *(const f32x4*)pp;
        sd.r1[jp] = *(const f32x4*)(pp + 4);
      } else if constexpr (kAux) {
        sd.a8[jp] = *(const bf16x8*)(p.aux + (size_t)m * p.ldaux + n);
      }
    }
  };
  Side cur, nxt;
  fetch(0, cur);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if (i + 1 < MT) fetch(i + 1, nxt);
    const int m = mbase + i * 16 + l15;
    [[maybe_unused]] float ps1 = 0.f, ps2 = 0.f;             // kStat: this lane's share of the row's (sum, sum of squares)
    if (m < p.M) {
      size_t orow = (size_t)m;
      if constexpr (kPos) {
        const int s = m / p.rows_in;
        orow = (size_t)s * p.rows_out + p.row_off + (m - s * p.rows_in);
      }
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        const int n = nbase + 32 * jp + 8 * lq;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * jp][e] + bv[jp][0][e]; v[4 + e] = acc[i][2 * jp + 1][e] + bv[jp][1][e]; }
        if constexpr (kFold) {
          if (p.ln_mean != nullptr) {                        // y = rstd * (acc - mean * c1) + bias
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = __builtin_fmaf(__builtin_fmaf(-cur.mu, c1v[jp][0][e], acc[i][2 * jp][e]), cur.rs, bv[jp][0][e]);
              v[4 + e] = __builtin_fmaf(__builtin_fmaf(-cur.mu, c1v[jp][1][e], acc[i][2 * jp + 1][e]), cur.rs, bv[jp][1][e]);
            }
          }
        }
        auto store_f32 = [&](float* dst) {
          *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};
          *(f32x4*)(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
        };
        auto store_bf16 = [&](bf16* dst) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
          *(bf16x8*)dst = o;
        };
        [[maybe_unused]] auto drop8 = [&]() {               // v *= mask / keep, element (m, n + e)
          const unsigned long long sd = p.seed + *p.seed_ptr;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= drop_scale(sd, (unsigned long long)m * p.N + n + e, p.drop_thresh, p.inv_keep);
        };
        if constexpr (EPI == GVK_EPI_STORE_BF16) {
          if (n < p.scale_cols) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= p.col_scale;
          }
          store_bf16((bf16*)p.out0 + (size_t)m * p.ldo + n);
        } else if constexpr (kRes) {
          if constexpr (DROP) drop8();                       // out = res + dropout(acc + bias): vision_transformer.py:34,54
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += cur.r0[jp][e]; v[4 + e] += cur.r1[jp][e]; }
          store_f32((float*)p.out0 + (size_t)m * p.ldo + n);
          if constexpr (EPI == GVK_EPI_BIAS_RES_F32_BF16) {
            store_bf16((bf16*)p.out1 + (size_t)m * p.ldo + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dv = v[e] - cur.pv; ps1 += dv; ps2 = __builtin_fmaf(dv, dv, ps2); }
          }
        } else if constexpr (EPI == GVK_EPI_BIAS_GELU_BF16) {
          if (p.aux_grad != 0) {                             // out0 = GELU'(pre): what the fc2 dgrad multiplies by, from the exponential GELU needs anyway
            bf16x8 g8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float y, dy;
              gelu_fast_both(v[e], y, dy);
              v[e] = y;
              g8[e] = (bf16)dy;
            }
            *(bf16x8*)((bf16*)p.out0 + (size_t)m * p.ldo + n) = g8;
          } else {
            if (p.out0 != nullptr) store_bf16((bf16*)p.out0 + (size_t)m * p.ldo + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
          }
          if constexpr (DROP) drop8();                       // out1 = dropout(GELU(pre)): vision_transformer.py:32-33
          store_bf16((bf16*)p.out1 + (size_t)m * p.ldo + n);
        } else if constexpr (kPos) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += cur.r0[jp][e]; v[4 + e] += cur.r1[jp][e]; }
          store_f32((float*)p.out0 + orow * p.ldo + n);
          if (p.out1 != nullptr) store_f32((float*)p.out1 + (size_t)m * p.ldo + n);
        } else if constexpr (EPI == GVK_EPI_GELU_BWD_BF16) {
          if constexpr (DROP) drop8();                       // gradient through that dropout, same mask
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.aux_grad != 0 ? (float)cur.a8[jp][e] : gelu_fast_grad((float)cur.a8[jp][e]);
          store_bf16((bf16*)p.out0 + (size_t)m * p.ldo + n);
        } else if constexpr (EPI == GVK_EPI_STORE_F32) {
          store_f32((float*)p.out0 + (size_t)m * p.ldo + n);
        } else if constexpr (EPI == GVK_EPI_BIAS_RELU_BF16) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          store_bf16((bf16*)p.out0 + (size_t)m * p.ldo + n);
        } else if constexpr (EPI == GVK_EPI_RELU_BWD_BF16) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (float)cur.a8[jp][e] > 0.f ? v[e] : 0.f;
          store_bf16((bf16*)p.out0 + (size_t)m * p.ldo + n);
        }
      }
    }
    if constexpr (kStat) {
      if (p.stat_part != nullptr) {                          // wave-uniform: the four lanes lq = 0..3 of a row hold its column quarters
        ps1 += __shfl_xor(ps1, 16, 64); ps2 += __shfl_xor(ps2, 16, 64);
        ps1 += __shfl_xor(ps1, 32, 64); ps2 += __shfl_xor(ps2, 32, 64);
        if (lq == 0 && m < p.M) *(f32x2*)(p.stat_part + ((size_t)(nbase / (16 * NT)) * p.M + m) * 2) = f32x2{ps1, ps2};
      }
    }
    cur = nxt;
  }
}

// gemm8p_bf16.hip: 256 x 256 tile, eight waves in two groups staggered by one barrier (ping-pong on each SIMD's matrix pipe)
int launch_gemm8p(const GemmArgs& a, int epilogue, int variant, hipStream_t stream);
bool gemm8p_supports(int epilogue);

}  // namespace gvk
