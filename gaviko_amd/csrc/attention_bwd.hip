// Flash-style attention backward for gfx950, head dim 64, bf16 operands / fp32 accumulate.
// Autograd of vision_transformer.py:63-71.  P is recomputed from Q, K and the forward's log-sum-exp; nothing N x N is
// stored.  Two passes, no atomics, bitwise reproducible (dq pass first: it also leaves delta = rowsum(dO * O) for the other):
//   dkdv pass: workgroup = 128 keys (4 waves x 32) of one (batch, head), sweeping 32-query slices.  S[q][key] and
//              dP[q][key] are computed with the KEY on the MFMA lane, so their accumulators are directly the B operands
//              of dV^T += dO^T.P and dK^T += Q^T.dS (Q / dO tiles are read row-wise for S, dP and 4x16-transposed
//              (ds_read_b64_tr_b16) for the two gradient products -- one LDS image serves both).
//   dq pass:   workgroup = 128 queries, sweeping 64-key tiles exactly like the forward: S^T, dP^T with the QUERY on the
//              lane, dQ^T += K^T.dS^T with K^T gathered by transposed reads of the row-major K tile.
// The row constants (-lse/scale) are preloaded into the S accumulators, so P = exp2(c * S') needs no subtraction.
#include "common.hpp"
#include "dropout.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

// A/B switch (compile-time, tools/gpu experiments): raise the wave's priority around its MFMA clusters so that, of the two waves a SIMD
// hosts (two workgroups per CU), the one in a matrix phase issues first and the other fills the gaps with its softmax VALU work
#ifdef GVK_ATTN_PRIO
#define GVK_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define GVK_PRIO(x)
#endif

__device__ __forceinline__ int swz_b(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

constexpr int kQT = 64;                 // query rows staged per barrier pair (two 32-row MFMA sub-blocks)
constexpr int kTileQ = kQT * 128;       // bytes of a [kQT][64] bf16 tile

// With attention-probability dropout (mask M, 1/keep folded in): O = (P.M) V, so dV^T += dO^T.(P.M), dS = P.(M.dP - delta) and
// delta = rowsum(dO.O) is unchanged; the mask is regenerated from (b*H + head, query, key) exactly as the forward drew it.
struct AttnDrop { unsigned long long seed; const unsigned long long* seed_ptr; unsigned int thresh; float inv_keep; };

// ------------------------------------------------------------------------------------------------ dK, dV
template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            bf16* __restrict__ dqkv, int T, int H, int ld_qkv, int ld_o, float scale,
                                                            float scale_log2e, AttnDrop dr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 bufs][Q tile | dO tile | lse' kQT f32 | delta kQT f32]
  constexpr int kBuf = 2 * kTileQ + 2 * kQT * 4;
  int bh, kblk;
  xcd_group_block(blockIdx.x, (T + 127) / 128, gridDim.x / ((T + 127) / 128), bh, kblk);   // all key blocks of a (batch, head) on one XCD
  const int b = bh / H, head = bh - b * H, k0 = kblk * 128;
  const int lane = lane_id(), wave = wave_id();
  const int r31 = lane & 31, hh = lane >> 5;
  const int inner = H * 64;
  const bf16* qbase = qkv + (size_t)b * T * ld_qkv + head * 64;
  const bf16* dobase = d_o + (size_t)b * T * ld_o + head * 64;
  const float* lse_b = lse + ((size_t)b * H + head) * T;
  const float* del_b = delta + ((size_t)b * H + head) * T;

  // K, V fragments of this wave's 32 keys: B operands (col = key, k = d)
  const int key = k0 + wave * 32 + r31;
  const int keyc = min(key, T - 1);
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = *(const bf16x8*)(qbase + inner + (size_t)keyc * ld_qkv + 16 * ks + 8 * hh);
    vf[ks] = *(const bf16x8*)(qbase + 2 * inner + (size_t)keyc * ld_qkv + 16 * ks + 8 * hh);
  }

  [[maybe_unused]] unsigned int akey = 0u;
  if constexpr (DROP) akey = attn_key(dr.seed + *dr.seed_ptr, b * H + head);
  const float inv_scale = 1.0f / scale;
  auto stage = [&](int buf, int qt) {
    char* sQ = smem + buf * kBuf;
    char* sD = sQ + kTileQ;
    float* sL = (float*)(sD + kTileQ);
#pragma unroll
    for (int r = 0; r < kQT / 32; ++r) {
      const int row = r * 32 + wave * 8 + (lane >> 3), slot = lane & 7;
      const int q = min(qt * kQT + row, T - 1);
      const int chunk = slot ^ swz_b(row);
      glds16(qbase + (size_t)q * ld_qkv + chunk * 8, sQ + (r * 32 + wave * 8) * 128);
      glds16(dobase + (size_t)q * ld_o + chunk * 8, sD + (r * 32 + wave * 8) * 128);
    }
    // The row constants ride on the LDS-DMA too (4-byte form, 64 rows per instruction; every wave writes the same 256 bytes: benign, and it
    // keeps the per-wave vmcnt identical).  As ORDINARY loads they made hipcc wait vmcnt(0) at their first use -- right here, draining the
    // Q / dO requests issued two lines above: one exposed L2 round trip per 64-row tile (SQ_WAIT_ANY 0.41 of the wave cycles).
    {
      const int qq = min(qt * kQT + lane, T - 1);
      __builtin_amdgcn_global_load_lds((const GVK_GLOBAL void*)(lse_b + qq), (GVK_LDS void*)sL, 4, 0, 0);
      __builtin_amdgcn_global_load_lds((const GVK_GLOBAL void*)(del_b + qq), (GVK_LDS void*)(sL + kQT), 4, 0, 0);
    }
  };

  f32x16 dkt[2], dvt[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { dkt[i] = f32x16{}; dvt[i] = f32x16{}; }

  const int nqt = (T + kQT - 1) / kQT;
  const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  stage(0, 0);
  __syncthreads();
  for (int qt = 0; qt < nqt; ++qt) {
    const int buf = qt & 1;
    if (qt + 1 < nqt) stage(buf ^ 1, qt + 1);
    const char* sQ0 = smem + buf * kBuf;
    const char* sD0 = sQ0 + kTileQ;
    const float* sL0 = (const float*)(sD0 + kTileQ);
#pragma unroll
    for (int sub = 0; sub < kQT / 32; ++sub) {
      if (qt * kQT + sub * 32 >= T) break;                 // wave-uniform: sub-block entirely past the sequence
      const char* sQ = sQ0 + sub * 32 * 128;               // (32 rows = a multiple of the swizzle period 16)
      const char* sD = sD0 + sub * 32 * 128;
      const float* sL = sL0 + sub * 32;
      // S'[q][key] = Q.K^T - lse/scale ;  dP[q][key] = dO.V^T
      f32x16 s, dp;
      const int rows_left = T - (qt * kQT + sub * 32);       // query rows of this sub-block inside the sequence (wave-uniform)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 l4 = *(const f32x4*)(sL + 8 * g4 + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[4 * g4 + e] = -l4[e] * inv_scale;
      }
      if (rows_left < 32) {                                  // rows >= T: P = exp2(-inf) = 0
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (8 * g4 + 4 * hh + e >= rows_left) s[4 * g4 + e] = -INFINITY;
      }
      dp = f32x16{};
      GVK_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int chunk = 2 * ks + hh;
        const bf16x8 qa = *(const bf16x8*)(sQ + r31 * 128 + ((chunk ^ swz_b(r31)) << 4));
        const bf16x8 da = *(const bf16x8*)(sD + r31 * 128 + ((chunk ^ swz_b(r31)) << 4));
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
      }
      GVK_PRIO(0);
      // P = exp2(c * S');  dS = P * (dP - delta[q]) -- two scores per packed instruction (these loops, not the MFMAs, fill the SIMD)
      const f32x2 sc2 = {scale_log2e, scale_log2e};
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 d4 = *(const f32x4*)(sL + kQT + 8 * g4 + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const f32x2 a = f32x2{s[4 * g4 + e], s[4 * g4 + e + 1]} * sc2;
          f32x2 pr = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
          f32x2 dpv = {dp[4 * g4 + e], dp[4 * g4 + e + 1]};
          f32x2 pd = pr;                                      // the P that multiplies dO in dV: dropped and rescaled under DROP
          if constexpr (DROP) {
            const unsigned int qq = (unsigned int)(qt * kQT + sub * 32 + 8 * g4 + 4 * hh + e);
            const f32x2 mm = {attn_drop_scale(akey, qq * (unsigned int)T + (unsigned int)key, dr.thresh, dr.inv_keep),
                              attn_drop_scale(akey, (qq + 1u) * (unsigned int)T + (unsigned int)key, dr.thresh, dr.inv_keep)};
            pd = pr * mm;
            dpv = dpv * mm;
          }
          const f32x2 ds = pr * (dpv - f32x2{d4[e], d4[e + 1]});
          s[4 * g4 + e] = pd[0]; s[4 * g4 + e + 1] = pd[1];
          dp[4 * g4 + e] = ds[0]; dp[4 * g4 + e + 1] = ds[1];
        }
      }
      // dV^T[d][key] += dO^T[d][q] . P[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key]   (k = q, accumulator row order)
      GVK_PRIO(1);
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        bf16x8 pf, dsf;
#pragma unroll
        for (int j = 0; j < 8; ++j) { pf[j] = (bf16)s[8 * sk + j]; dsf[j] = (bf16)dp[8 * sk + j]; }
        const int q0r = 16 * sk + 4 * (g >> 1);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int chunk = db * 4 + 2 * (g & 1) + (tp >> 1);
          const int ra = q0r + tq, rb = q0r + 8 + tq;
          const int oa = ra * 128 + ((chunk ^ swz_b(ra)) << 4) + (tp & 1) * 8;
          const int ob = rb * 128 + ((chunk ^ swz_b(rb)) << 4) + (tp & 1) * 8;
          const bf16x4 da0 = lds_read_tr16(sD + oa), da1 = lds_read_tr16(sD + ob);
          const bf16x4 qa0 = lds_read_tr16(sQ + oa), qa1 = lds_read_tr16(sQ + ob);
          const bf16x8 dof = {da0[0], da0[1], da0[2], da0[3], da1[0], da1[1], da1[2], da1[3]};
          const bf16x8 qf = {qa0[0], qa0[1], qa0[2], qa0[3], qa1[0], qa1[1], qa1[2], qa1[3]};
          dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof, pf, dvt[db], 0, 0, 0);
          dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, dsf, dkt[db], 0, 0, 0);
        }
      }
      GVK_PRIO(0);
    }
    __syncthreads();
  }
  if (key < T) {
    bf16* dk_row = dqkv + ((size_t)b * T + key) * ld_qkv + inner + head * 64;
    bf16* dv_row = dk_row + inner;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d = db * 32 + 8 * g4 + 4 * hh;
        bf16x4 ok = {(bf16)(dkt[db][4 * g4] * scale), (bf16)(dkt[db][4 * g4 + 1] * scale), (bf16)(dkt[db][4 * g4 + 2] * scale),
                     (bf16)(dkt[db][4 * g4 + 3] * scale)};
        bf16x4 ov = {(bf16)dvt[db][4 * g4], (bf16)dvt[db][4 * g4 + 1], (bf16)dvt[db][4 * g4 + 2], (bf16)dvt[db][4 * g4 + 3]};
        *(bf16x4*)(dk_row + d) = ok;
        *(bf16x4*)(dv_row + d) = ov;
      }
  }
}

// ------------------------------------------------------------------------------------------------ dQ
constexpr int kKB2 = 64;
constexpr int kTile64 = 64 * 128;

// Runs FIRST: it also produces delta[b][h][q] = sum_d dO * O for its own queries (the rows are in its registers anyway) and leaves
// it in memory for the dK/dV pass, so no separate row-sum kernel sits on the critical path.
template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o_fwd, const bf16* __restrict__ d_o,
                                                          const float* __restrict__ lse, float* __restrict__ delta,
                                                          bf16* __restrict__ dqkv, int T, int H, int ld_qkv, int ld_o, float scale,
                                                          float scale_log2e, AttnDrop dr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 bufs][K tile | V tile]
  int bh, qblk;
  xcd_group_block(blockIdx.x, (T + 127) / 128, gridDim.x / ((T + 127) / 128), bh, qblk);
  const int b = bh / H, head = bh - b * H, q0 = qblk * 128;
  const int lane = lane_id(), wave = wave_id();
  const int r31 = lane & 31, hh = lane >> 5;
  const int inner = H * 64;
  const bf16* base = qkv + (size_t)b * T * ld_qkv + head * 64;
  const bf16* kbase = base + inner;
  const bf16* vbase = base + 2 * inner;
  const int q = q0 + wave * 32 + r31;
  const int qc = min(q, T - 1);
  bf16x8 qf[4], dof[4];
  float del = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = *(const bf16x8*)(base + (size_t)qc * ld_qkv + 16 * ks + 8 * hh);
    dof[ks] = *(const bf16x8*)(d_o + ((size_t)b * T + qc) * ld_o + head * 64 + 16 * ks + 8 * hh);
    const bf16x8 of = *(const bf16x8*)(o_fwd + ((size_t)b * T + qc) * ld_o + head * 64 + 16 * ks + 8 * hh);
#pragma unroll
    for (int j = 0; j < 8; ++j) del += (float)of[j] * (float)dof[ks][j];
  }
  del += __shfl_xor(del, 32, 64);                         // the two half-waves hold the two halves of the 64-wide row
  if (hh == 0 && q < T) delta[((size_t)b * H + head) * T + q] = del;
  [[maybe_unused]] unsigned int akey = 0u, qoff = 0u;
  if constexpr (DROP) {
    akey = attn_key(dr.seed + *dr.seed_ptr, b * H + head);
    qoff = (unsigned int)q * (unsigned int)T;
  }
  const float sinit = -lse[((size_t)b * H + head) * T + qc] / scale;

  auto stage = [&](int buf, int kt) {
    char* sK = smem + buf * 2 * kTile64;
    char* sV = sK + kTile64;
    const int rsub = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = r * 32 + wave * 8 + rsub;
      const int key = min(kt * kKB2 + row, T - 1);
      const int chunk = slot ^ swz_b(row);
      glds16(kbase + (size_t)key * ld_qkv + chunk * 8, sK + (r * 32 + wave * 8) * 128);
      glds16(vbase + (size_t)key * ld_qkv + chunk * 8, sV + (r * 32 + wave * 8) * 128);
    }
  };

  f32x16 dqt[2];
  dqt[0] = f32x16{};
  dqt[1] = f32x16{};
  const int nkt = (T + kKB2 - 1) / kKB2;
  const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) stage(buf ^ 1, kt + 1);
    const char* sK = smem + buf * 2 * kTile64;
    const char* sV = sK + kTile64;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = sinit;
      dpt = f32x16{};
      const int row = kb * 32 + r31;
      GVK_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int chunk = 2 * ks + hh;
        const int off = row * 128 + ((chunk ^ swz_b(row)) << 4);
        const bf16x8 ka = *(const bf16x8*)(sK + off);
        const bf16x8 va = *(const bf16x8*)(sV + off);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[ks], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[ks], dpt, 0, 0, 0);
      }
      GVK_PRIO(0);
      if (kt == nkt - 1) {                                 // wave-uniform: only the last tile holds keys >= T
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * kKB2 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          st[r] = (key < T) ? st[r] : -INFINITY;
        }
      }
      const f32x2 sc2 = {scale_log2e, scale_log2e}, del2 = {del, del};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {                                                                     // dS^T, two scores per packed op
        f32x2 dpv = {dpt[r], dpt[r + 1]};
        if constexpr (DROP) {
          const int key = kt * kKB2 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;          // r even: r + 1 is the next key
          dpv = dpv * f32x2{attn_drop_scale(akey, qoff + (unsigned int)key, dr.thresh, dr.inv_keep),
                            attn_drop_scale(akey, qoff + (unsigned int)key + 1u, dr.thresh, dr.inv_keep)};
        }
        const f32x2 a = f32x2{st[r], st[r + 1]} * sc2;
        const f32x2 ds = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} * (dpv - del2);
        st[r] = ds[0];
        st[r + 1] = ds[1];
      }
      GVK_PRIO(1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 dsf;
#pragma unroll
        for (int j = 0; j < 8; ++j) dsf[j] = (bf16)st[8 * s + j];
        const int key0 = kb * 32 + 16 * s + 4 * (g >> 1);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int chunk = db * 4 + 2 * (g & 1) + (tp >> 1);
          const int ra = key0 + tq, rb = key0 + 8 + tq;
          const bf16x4 ka0 = lds_read_tr16(sK + ra * 128 + ((chunk ^ swz_b(ra)) << 4) + (tp & 1) * 8);
          const bf16x4 ka1 = lds_read_tr16(sK + rb * 128 + ((chunk ^ swz_b(rb)) << 4) + (tp & 1) * 8);
          const bf16x8 kt8 = {ka0[0], ka0[1], ka0[2], ka0[3], ka1[0], ka1[1], ka1[2], ka1[3]};
          dqt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt8, dsf, dqt[db], 0, 0, 0);
        }
      }
      GVK_PRIO(0);
    }
    __syncthreads();
  }
  if (q < T) {
    bf16* dq_row = dqkv + ((size_t)b * T + q) * ld_qkv + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 o = {(bf16)(dqt[db][4 * g4] * scale), (bf16)(dqt[db][4 * g4 + 1] * scale), (bf16)(dqt[db][4 * g4 + 2] * scale),
                    (bf16)(dqt[db][4 * g4 + 3] * scale)};
        *(bf16x4*)(dq_row + db * 32 + 8 * g4 + 4 * hh) = o;
      }
  }
}

}  // namespace gvk

extern "C" int gvk_attention_bwd_bf16_dropout(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B,
                                              int T, int H, int ld_qkv, int ld_out, float scale, float drop_p, uint64_t seed, const void* seed_ptr,
                                              void* stream) {
  using namespace gvk;
  GVK_REQUIRE(qkv && out && dout && lse && delta && dqkv, "gvk_attention_bwd_bf16: null pointer");
  GVK_REQUIRE(B > 0 && T > 0 && H > 0, "gvk_attention_bwd_bf16: empty shape");
  GVK_REQUIRE(ld_qkv >= 3 * H * 64 && ld_qkv % 8 == 0 && ld_out >= H * 64 && ld_out % 8 == 0,
              "gvk_attention_bwd_bf16: head dim is fixed at 64; ld_qkv=%d ld_out=%d inconsistent with H=%d", ld_qkv, ld_out, H);
  GVK_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed_ptr != nullptr), "gvk_attention_bwd_bf16: drop_p in [0,1) and a seed word");
  GVK_REQUIRE(drop_p == 0.f || (int64_t)T * T < (int64_t)1 << 32, "gvk_attention_bwd_bf16: the dropout mask index (query*T + key) is 32-bit");
  const AttnDrop dr{seed, (const unsigned long long*)seed_ptr, drop_threshold_u32(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f};
  hipStream_t s = (hipStream_t)stream;
  const float sl2 = scale * 1.44269504088896340736f;
  const dim3 grid(((T + 127) / 128) * H * B);
  const unsigned lds_kv = 2 * (2 * kTileQ + 2 * kQT * 4), lds_q = 2 * 2 * kTile64;
  int rc;
  if (drop_p > 0.f) {
    GVK_LAUNCH(attn_bwd_dq_kernel<true>, grid, dim3(256), lds_q, s, (const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, delta, (bf16*)dqkv, T, H,
               ld_qkv, ld_out, scale, sl2, dr);
    rc = check_launch("attention_bwd/dq");
    if (rc) return rc;
    GVK_LAUNCH(attn_bwd_dkdv_kernel<true>, grid, dim3(256), lds_kv, s, (const bf16*)qkv, (const bf16*)dout, lse, delta, (bf16*)dqkv, T, H, ld_qkv,
               ld_out, scale, sl2, dr);
    return check_launch("attention_bwd/dkdv");
  }
  GVK_LAUNCH(attn_bwd_dq_kernel<false>, grid, dim3(256), lds_q, s, (const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, delta, (bf16*)dqkv, T, H,
             ld_qkv, ld_out, scale, sl2, dr);
  rc = check_launch("attention_bwd/dq");
  if (rc) return rc;
  GVK_LAUNCH(attn_bwd_dkdv_kernel<false>, grid, dim3(256), lds_kv, s, (const bf16*)qkv, (const bf16*)dout, lse, delta, (bf16*)dqkv, T, H, ld_qkv,
             ld_out, scale, sl2, dr);
  return check_launch("attention_bwd/dkdv");
}

extern "C" int gvk_attention_bwd_bf16(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B,
                                      int T, int H, int ld_qkv, int ld_out, float scale, void* stream) {
  return gvk_attention_bwd_bf16_dropout(qkv, out, dout, lse, delta, dqkv, B, T, H, ld_qkv, ld_out, scale, 0.f, 0, nullptr, stream);
}
