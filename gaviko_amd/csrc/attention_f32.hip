// fp32 multi-head self-attention (head dim 64) for the reference's fp32 configurations (BASELINE cfg4, tolerance 1e-5):
// same interface and data layout as the bf16 kernels (qkv [B*T][3*H*64] in, ctx [B*T][H*64] out, lse saved), flash-style (no [T][T]
// score matrix in memory), every product on the fp32 matrix cores: v_mfma_f32_32x32x2_f32 multiplies exact fp32 products into an fp32
// accumulator, so the path keeps fp32 accuracy (1e-5 against float64) and runs at the fp32 MATRIX rate instead of one LDS broadcast
// read per four VALU FMAs (round 4: the row-per-thread form this file held before took 5.5 ms per layer at cfg4, 68 % of its step).
//
// One workgroup = 4 waves = 128 rows of one (batch, head); a wave owns 32 rows whose operand side stays in registers, the other side
// streams through LDS in 32-row tiles.  A 32x32x2 step contracts TWO k indices, one per 32-lane half (hh = lane >> 5); the contraction
// order is free, so
//   * a score block S^T[key][query] = K . Q^T runs over d in the order d = 8j + 4hh + t (j = 0..7, t = 0..3): a lane reads its four
//     values of a step group as ONE 16-byte LDS read, and the register side holds the row's 32 values of its half in that order;
//   * the accumulator of that block (lane = column, register r = row (r & 3) + 8 (r >> 2) + 4 hh) IS the register side of the next
//     product over those rows (O^T += V^T . P^T, dQ^T += K^T . dS^T, dV^T += dO^T . P, dK^T += Q^T . dS): step r contracts the two rows
//     that register r holds in the two lane halves, and the streamed side is read from the row-major LDS tile at exactly those rows --
//     probabilities never move between lanes.
// Per 32 x 32 block a wave issues 64 (forward), 96 (dQ pass) or 128 (dK / dV pass) MFMAs of 64 cycles each beside ~50 LDS reads and
// ~100 VALU instructions: the kernels are bound by the matrix pipe.
#include "common.hpp"
#include "dropout.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

// attention-probability dropout (thresh = 0: off): same mask function as the bf16 kernels, element (b*H + h, query, key)
struct AttnDropF { unsigned long long seed; const unsigned long long* seed_ptr; unsigned int thresh; float inv_keep; };

constexpr int kRows = 128;        // rows per workgroup (4 waves x 32)
constexpr int kTile = 32;         // rows of the streamed operand per LDS tile
constexpr int kLd = 68;           // LDS row pitch in floats (64 + 4: rows 4 banks apart)

// stage rows [r0, r0 + 32) x 64 floats of `src` (row stride ld) into dst[32][kLd]; rows past the sequence re-read row T-1 (finite data; the
// callers mask those rows / columns)
__device__ __forceinline__ void stage32(float* dst, const float* __restrict__ src, int r0, int T, int ld) {
  for (int i = threadIdx.x; i < kTile * 16; i += 2 * kRows) {
    const int r = i >> 4, c = (i & 15) * 4;
    *(f32x4*)(dst + r * kLd + c) = *(const f32x4*)(src + (size_t)min(r0 + r, T - 1) * ld + c);
  }
}
// this lane's half of one row as the register side of a product over d: v[4j + t] = row[8j + 4hh + t] * mul
__device__ __forceinline__ void load_half_row(float (&v)[32], const float* __restrict__ row, int hh, float mul) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const f32x4 x = *(const f32x4*)(row + 8 * j + 4 * hh);
    v[4 * j] = x[0] * mul; v[4 * j + 1] = x[1] * mul; v[4 * j + 2] = x[2] * mul; v[4 * j + 3] = x[3] * mul;
  }
}
// acc[row r31 of the tile][column = this lane's register-side row] += sum_d tile[r31][d] * reg[d]
__device__ __forceinline__ f32x16 rows_dot(const float* __restrict__ tile, const float (&reg)[32], int r31, int hh) {
  f32x16 acc = {};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const f32x4 a = *(const f32x4*)(tile + r31 * kLd + 8 * j + 4 * hh);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], reg[4 * j + t], acc, 0, 0, 0);
  }
  return acc;
}
// out^T[d][column] += sum over the tile's 32 rows of tile[row][d] * w[row][column], w given as an accumulator (register r = row rr(r, hh))
__device__ __forceinline__ int rr(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }
__device__ __forceinline__ void cols_axpy(f32x16 (&out)[2], const float* __restrict__ tile, const f32x16& w, int r31, int hh) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* row = tile + rr(r, hh) * kLd + r31;
    out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(row[0], w[r], out[0], 0, 0, 0);
    out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(row[32], w[r], out[1], 0, 0, 0);
  }
}
// out[row][d] = acc^T * mul: lane (row = r31, hh) holds d = 32 db + 8 g + 4 hh + (0..3) in registers 4g .. 4g+3 of acc[db]
__device__ __forceinline__ void store_t(float* __restrict__ dst_row, const f32x16 (&acc)[2], float mul, int hh) {
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(f32x4*)(dst_row + 32 * db + 8 * g + 4 * hh) =
          f32x4{acc[db][4 * g] * mul, acc[db][4 * g + 1] * mul, acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul};
}
// lanes l and l + 32 hold the two halves of one column's data: maximum / sum over the pair (symmetric in the two results of the swap)
__device__ __forceinline__ float pair_max(float v) {
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned int)r[0]), __builtin_bit_cast(float, (unsigned int)r[1]));
}
__device__ __forceinline__ float pair_sum(float v) {
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned int)r[0]) + __builtin_bit_cast(float, (unsigned int)r[1]);
}

// ---- forward: vision_transformer.py:63-71
template <bool DROP>
__global__ __launch_bounds__(2 * kRows) void attn_f32_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse,
                                                                 int T, int H, int ld_qkv, int ld_out, float scale, AttnDropF dr) {
  __shared__ __attribute__((aligned(16))) float sK[kTile * kLd], sV[kTile * kLd];
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * kRows;
  const int lane = lane_id(), wave = wave_id(), r31 = lane & 31, hh = lane >> 5;
  const int inner = H * 64;
  const float* base = qkv + (size_t)b * T * ld_qkv + h * 64;
  const int q = q0 + wave * 32 + r31;
  const bool active = q0 + wave * 32 < T;          // (wave-uniform) a wave whose rows all lie past the sequence only helps staging
  float qf[32];
  load_half_row(qf, base + (size_t)min(q, T - 1) * ld_qkv, hh, scale);
  [[maybe_unused]] unsigned int akey = 0u;
  if constexpr (DROP) akey = attn_key(dr.seed + *dr.seed_ptr, b * H + h);
  f32x16 ot[2] = {};
  float m = -INFINITY, l = 0.f;                    // l: this half's share of the row sum (its 16 keys of every tile)
  for (int k0 = 0; k0 < T; k0 += kTile) {
    __syncthreads();
    stage32(sK, base + inner, k0, T, ld_qkv);
    stage32(sV, base + 2 * inner, k0, T, ld_qkv);
    __syncthreads();
    if (!active) continue;
    f32x16 st = rows_dot(sK, qf, r31, hh);         // S^T[key][query] * scale
    if (k0 + kTile > T) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (k0 + rr(r, hh) >= T) st[r] = -INFINITY;
    }
    float mx = st[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
    mx = pair_max(mx);
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(st[r] - mn);
      ps += p;                                     // the softmax statistics stay those of the undropped scores
      st[r] = p;
      if constexpr (DROP) st[r] = p * attn_drop_scale(akey, (unsigned int)q * (unsigned int)T + (unsigned int)(k0 + rr(r, hh)), dr.thresh, dr.inv_keep);
    }
    l = l * alpha + ps;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ot[0][r] *= alpha; ot[1][r] *= alpha; }
    cols_axpy(ot, sV, st, r31, hh);                // O^T[d][query] += V^T . P^T
  }
  if (!active) return;
  const float lt = pair_sum(l);
  if (q < T) {
    store_t(out + ((size_t)b * T + q) * ld_out + h * 64, ot, 1.0f / lt, hh);
    if (lse != nullptr && hh == 0) lse[((size_t)b * H + h) * T + q] = m + __logf(lt);
  }
}

// ---- backward, query side: delta_i = dO_i . O_i ; dQ_i = scale * sum_j P_ij (M_ij dO_i . V_j - delta_i) K_j
template <bool DROP>
__global__ __launch_bounds__(2 * kRows) void attn_f32_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                                    const float* __restrict__ dout, const float* __restrict__ lse,
                                                                    float* __restrict__ delta, float* __restrict__ dqkv, int T, int H,
                                                                    int ld_qkv, int ld_out, float scale, AttnDropF dr) {
  __shared__ __attribute__((aligned(16))) float sK[kTile * kLd], sV[kTile * kLd];
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * kRows;
  const int lane = lane_id(), wave = wave_id(), r31 = lane & 31, hh = lane >> 5;
  const int inner = H * 64;
  const float* base = qkv + (size_t)b * T * ld_qkv + h * 64;
  const int q = q0 + wave * 32 + r31, qc = min(q, T - 1);
  const bool active = q0 + wave * 32 < T;
  float qf[32], dof[32];
  load_half_row(qf, base + (size_t)qc * ld_qkv, hh, scale);
  load_half_row(dof, dout + ((size_t)b * T + qc) * ld_out + h * 64, hh, 1.f);
  float dl = 0.f;
  {
    float of[32];
    load_half_row(of, o + ((size_t)b * T + qc) * ld_out + h * 64, hh, 1.f);
#pragma unroll
    for (int i = 0; i < 32; ++i) dl = __builtin_fmaf(dof[i], of[i], dl);
    dl = pair_sum(dl);
  }
  const float ls = lse[((size_t)b * H + h) * T + qc];
  if (q < T && hh == 0) delta[((size_t)b * H + h) * T + q] = dl;
  [[maybe_unused]] unsigned int akey = 0u;
  if constexpr (DROP) akey = attn_key(dr.seed + *dr.seed_ptr, b * H + h);
  f32x16 dqt[2] = {};
  for (int k0 = 0; k0 < T; k0 += kTile) {
    __syncthreads();
    stage32(sK, base + inner, k0, T, ld_qkv);
    stage32(sV, base + 2 * inner, k0, T, ld_qkv);
    __syncthreads();
    if (!active) continue;
    f32x16 st = rows_dot(sK, qf, r31, hh);         // S^T * scale
    const f32x16 dp = rows_dot(sV, dof, r31, hh);  // dP^T[key][query] = V . dO^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + rr(r, hh);
      const float p = key < T ? __expf(st[r] - ls) : 0.f;
      float dpv = dp[r];
      if constexpr (DROP) dpv *= attn_drop_scale(akey, (unsigned int)q * (unsigned int)T + (unsigned int)key, dr.thresh, dr.inv_keep);
      st[r] = p * (dpv - dl);
    }
    cols_axpy(dqt, sK, st, r31, hh);               // dQ^T[d][query] += K^T . dS^T
  }
  if (active && q < T) store_t(dqkv + ((size_t)b * T + q) * ld_qkv + h * 64, dqt, scale, hh);
}

// ---- backward, key side: dV_j = sum_i (P M)_ij dO_i ;  dK_j = scale * sum_i P_ij (M_ij dO_i . V_j - delta_i) Q_i
template <bool DROP>
__global__ __launch_bounds__(2 * kRows) void attn_f32_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                                    float* __restrict__ dqkv, int T, int H, int ld_qkv, int ld_out, float scale,
                                                                    AttnDropF dr) {
  __shared__ __attribute__((aligned(16))) float sQ[kTile * kLd], sD[kTile * kLd];
  __shared__ float sL[kTile], sDl[kTile];
  const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * kRows;
  const int lane = lane_id(), wave = wave_id(), r31 = lane & 31, hh = lane >> 5;
  const int inner = H * 64;
  const float* base = qkv + (size_t)b * T * ld_qkv + h * 64;
  const int key = k0 + wave * 32 + r31, kc = min(key, T - 1);
  const bool active = k0 + wave * 32 < T;
  float kf[32], vf[32];
  load_half_row(kf, base + inner + (size_t)kc * ld_qkv, hh, scale);          // (K * scale: the score operand)
  load_half_row(vf, base + 2 * inner + (size_t)kc * ld_qkv, hh, 1.f);
  [[maybe_unused]] unsigned int akey = 0u;
  if constexpr (DROP) akey = attn_key(dr.seed + *dr.seed_ptr, b * H + h);
  f32x16 dkt[2] = {}, dvt[2] = {};
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
    if (!active) continue;
    f32x16 s = rows_dot(sQ, kf, r31, hh);          // S[query][key] * scale
    f32x16 dp = rows_dot(sD, vf, r31, hh);         // dP[query][key] = dO . V^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = rr(r, hh);
      const float p = q0 + qi < T ? __expf(s[r] - sL[qi]) : 0.f;
      float pm = p, dpv = dp[r];
      if constexpr (DROP) {
        const float mm = attn_drop_scale(akey, (unsigned int)(q0 + qi) * (unsigned int)T + (unsigned int)key, dr.thresh, dr.inv_keep);
        pm = p * mm;
        dpv *= mm;
      }
      s[r] = pm;                                   // (P o M)[query][key]
      dp[r] = p * (dpv - sDl[qi]);                 // dS[query][key]
    }
    cols_axpy(dvt, sD, s, r31, hh);                // dV^T[d][key] += dO^T . (P o M)
    cols_axpy(dkt, sQ, dp, r31, hh);               // dK^T[d][key] += Q^T . dS
  }
  if (active && key < T) {
    float* row = dqkv + ((size_t)b * T + key) * ld_qkv + inner + h * 64;
    store_t(row, dkt, scale, hh);
    store_t(row + inner, dvt, 1.0f, hh);
  }
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
  const dim3 grid((T + kRows - 1) / kRows, H, B), block(2 * kRows);
  if (drop_p > 0.f) GVK_LAUNCH((attn_f32_fwd_kernel<true>), grid, block, 0, (hipStream_t)stream, qkv, out, lse, T, H, ld_qkv, ld_out, scale, dr);
  else GVK_LAUNCH((attn_f32_fwd_kernel<false>), grid, block, 0, (hipStream_t)stream, qkv, out, lse, T, H, ld_qkv, ld_out, scale, dr);
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
  const dim3 grid((T + kRows - 1) / kRows, H, B), block(2 * kRows);
  hipStream_t s = (hipStream_t)stream;
  if (drop_p > 0.f) GVK_LAUNCH((attn_f32_bwd_dq_kernel<true>), grid, block, 0, s, qkv, out, dout, lse, delta, dqkv, T, H, ld_qkv, ld_out, scale, dr);
  else GVK_LAUNCH((attn_f32_bwd_dq_kernel<false>), grid, block, 0, s, qkv, out, dout, lse, delta, dqkv, T, H, ld_qkv, ld_out, scale, dr);
  int rc = check_launch("attention_bwd_f32/dq");
  if (rc) return rc;
  if (drop_p > 0.f) GVK_LAUNCH((attn_f32_bwd_kv_kernel<true>), grid, block, 0, s, qkv, dout, lse, (const float*)delta, dqkv, T, H, ld_qkv, ld_out, scale, dr);
  else GVK_LAUNCH((attn_f32_bwd_kv_kernel<false>), grid, block, 0, s, qkv, dout, lse, (const float*)delta, dqkv, T, H, ld_qkv, ld_out, scale, dr);
  return check_launch("attention_bwd_f32/dkdv");
}

extern "C" int gvk_attention_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv,
                                     int B, int T, int H, int ld_qkv, int ld_out, float scale, void* stream) {
  return gvk_attention_bwd_f32_dropout(qkv, out, dout, lse, delta, dqkv, B, T, H, ld_qkv, ld_out, scale, 0.f, 0, nullptr, stream);
}
