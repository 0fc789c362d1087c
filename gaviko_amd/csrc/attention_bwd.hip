// Flash-style attention backward for gfx950, head dim 64, bf16 operands / fp32 accumulate.
// Autograd of vision_transformer.py:63-71.  P is recomputed from Q, K and the forward's log-sum-exp; nothing N x N is
// stored.  Two passes, no atomics, bitwise reproducible (dq pass first: it also leaves delta = rowsum(dO * O) for the other):
//   dkdv pass: workgroup = 128 keys (4 waves x 32) of one (batch, head), sweeping 32-query slices.  S[q][key] and
//              dP[q][key] are computed with the KEY on the MFMA lane, so their accumulators are directly the B operands
//              of dV^T += dO^T.P and dK^T += Q^T.dS (Q / dO tiles are read row-wise for S, dP and 4x16-transposed
//              (ds_read_b64_tr_b16) for the two gradient products -- one LDS image serves both).
//   dq pass:   workgroup = 128 queries, sweeping key tiles exactly like the forward: S^T, dP^T with the QUERY on the
//              lane, dQ^T += K^T.dS^T with K^T gathered by transposed reads of the row-major K tile.
// Round 3 (same diet as the forward, attention_fwd.hip): both kernels were bound by vector issue -- per 32 x 32 score block ~100-120
// VALU instructions beside 12-16 MFMAs, half of them the row constants (S - lse, dP - delta), the x scale and key / row masks that
// hipcc had hoisted into per-tile v_cmp / v_cndmask chains.  Now
//   * the q block ARRIVES pre-scaled by scale*log2(e) (the qkv projection's epilogue, gvk_gemm_desc.scale_cols), exactly as the forward
//     read it: P = exp2(S') with no multiply, and the three kernels recompute bit-identical scores -- an in-kernel pre-scale of whichever
//     operand sits in registers (Q in one pass, K in the other) made P inconsistent with the forward's lse and cost 8x on the deepest
//     gradients of the adaptformer fixture;  dQ = scale . dS.K is the gradient of the UNSCALED q, dK = dS^T.Q' / log2(e);
//   * S' = S - lse*log2(e) (+ the mask as -3e38) and dP' = dP - delta come out of the matrix pipe: one extra MFMA each over an
//     augmented contraction (attention_common.hpp), no v_fma / v_sub / v_cndmask per score: a score costs v_exp, v_mul, and its
//     share of two v_cvt_pk;
//   * key / query tiles of 96 rows when they pad the sequence less than 128 (T = 1033: 1056 instead of 1152 / 1088), waves whose 32
//     rows lie wholly past the sequence only help staging;
//   * staging by LDS-DMA through buffer resources with SCALAR tile offsets (no per-tile vector address arithmetic);
//   * dQ / dK / dV leave through LDS as whole 128-byte rows.
#include "attention_common.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

// ------------------------------------------------------------------------------------------------ dK, dV
// QT: query rows staged per barrier pair (3 or 4 sub-blocks of 32)
template <int QT, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            bf16* __restrict__ dqkv, int T, int H, int ld_qkv, int ld_o, float dk_scale,
                                                            AttnDrop dr) {
  constexpr int NSB = QT / 32;
  constexpr int kTileQ = QT * 128;                // bytes of a [QT][64] bf16 tile
  constexpr int kBuf = 2 * kTileQ + 2 * 128 * 4;  // Q tile | dO tile | lse 128 f32 | delta 128 f32
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 bufs][kBuf]
  int bh, kblk;
  xcd_group_block(blockIdx.x, (T + 127) / 128, gridDim.x / ((T + 127) / 128), bh, kblk);   // all key blocks of a (batch, head) on one XCD
  const int b = bh / H, head = bh - b * H, k0 = kblk * 128;
  const int lane = lane_id(), wave = wave_id();
  const int r31 = lane & 31, hh = lane >> 5;
  const int inner = H * 64;
  const bf16* qbase = qkv + (size_t)b * T * ld_qkv + head * 64;
  const bool active = k0 + wave * 32 < T;          // a wave whose 32 keys all lie past the sequence only stages tiles

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

  // staging: Q rows (from qkv), dO rows, lse / delta (one 4-byte LDS-DMA per wave: waves 0,1 the two 64-row halves of lse, waves 2,3 of delta)
  const int nqt = (T + QT - 1) / QT;
  const int nB = (int)gridDim.x / (((T + 127) / 128) * H);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, nB * T * ld_qkv * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)d_o, 0, nB * T * ld_o * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(wave < 2 ? lse : delta), 0, nB * H * T * 4, 0x00020000);
  const int rsub = lane >> 3, slot = lane & 7;
  int voq[NSB], vod[NSB];
#pragma unroll
  for (int r = 0; r < NSB; ++r) {
    const int row = r * 32 + wave * 8 + rsub;
    const int col = head * 64 + ((slot ^ attn_swz(row)) << 3);
    voq[r] = ((b * T + row) * ld_qkv + col) * 2;
    vod[r] = ((b * T + row) * ld_o + col) * 2;
  }
  const int lrow = (wave & 1) * 64 + lane;                         // row of the tile whose constant this lane fetches
  const int vol = ((b * H + head) * T + lrow) * 4;
  auto stage = [&](int buf, int qt) {
    char* sQ = smem + buf * kBuf;
    char* sD = sQ + kTileQ;
    char* sL = sD + kTileQ;
    const bool last = qt == nqt - 1;
    // (the scalar offsets go through plain ints: with the template parameter inside the builtin's argument list hipcc's HOST pass dropped
    //  the whole kernel stub without a diagnostic -- an undefined symbol at load time)
    const int soq = qt * QT * ld_qkv * 2, sod = qt * QT * ld_o * 2, sol = qt * QT * 4;
#pragma unroll
    for (int r = 0; r < NSB; ++r) {
      // only the last tile can reach past the sequence: its rows step back to row T-1 (finite; masked by the row flag), recomputed here
      // rather than kept in registers
      const int over = last ? max(qt * QT + r * 32 + wave * 8 + rsub - (T - 1), 0) : 0;
      const int vq = voq[r] - over * ld_qkv * 2, vd = vod[r] - over * ld_o * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (GVK_LDS void*)(sQ + (r * 32 + wave * 8) * 128), 16, vq, soq, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (GVK_LDS void*)(sD + (r * 32 + wave * 8) * 128), 16, vd, sod, 0, 0);
    }
    const int overl = last ? max(qt * QT + lrow - (T - 1), 0) : 0;
    const int vl = vol - overl * 4;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (GVK_LDS void*)(sL + (wave >> 1) * 512 + (wave & 1) * 256), 4, vl, sol, 0, 0);
  };

  f32x16 dkt[2], dvt[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { dkt[i] = f32x16{}; dvt[i] = f32x16{}; }
  const bf16x8 sel_s = aug_sel_first(true, hh);      // [1, 1, 1, 1, 0...]: the query side carries -3e38 only in rows past the sequence
  [[maybe_unused]] const bf16x8 sel_d = aug_sel_second(hh);           // [0, 0, 0, 0, 1, 1, 1, 0]
  const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  stage(0, 0);
  __syncthreads();
  for (int qt = 0; qt < nqt; ++qt) {
    const int buf = qt & 1;
    if (qt + 1 < nqt) stage(buf ^ 1, qt + 1);
    if (active) {
      const char* sQ0 = smem + buf * kBuf;
      const char* sD0 = sQ0 + kTileQ;
      const float* sL0 = (const float*)(sD0 + kTileQ);
      // Three stages per 32-query sub-block, software-pipelined inside the wave so that the matrix pipe never waits for the VALU:
      //   B(sub): S', dP' (10 MFMAs)   C(sub): exp2, multiply, bf16 conversion (VALU)   D(sub): dV^T, dK^T (8 MFMAs + transposed reads)
      // issue order  B(0) | B(1) C(0) D(0) | B(2) C(1) D(1) | ... : C(sub) runs while B(sub+1) executes, B(sub+2) is issued behind D(sub).
      auto scores = [&](int sub, f32x16& s, f32x16& dp) {
        const int qrow0 = qt * QT + sub * 32;
        const char* sQ = sQ0 + sub * 32 * 128;               // (32 rows = a multiple of the swizzle period 16)
        const char* sD = sD0 + sub * 32 * 128;
        // constant side of the augmented MFMAs: this lane's query row r31 -> [-lse*log2e pieces, row >= T ? -3e38 : 0, -delta pieces, 0]
        const float l2 = sL0[sub * 32 + r31] * 1.44269504088896340736f;
        const float dl = DROP ? 0.f : sL0[128 + sub * 32 + r31];   // with dropout delta is subtracted after the mask (dS = P.(M.dP - delta))
        const bf16x8 qaug = aug_const(l2, qrow0 + r31 >= T, dl, hh);
        // S'[q][key] = Q'.K^T - lse2 ;  dP'[q][key] = dO.V^T - delta
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qaug, sel_s, f32x16{}, 0, 0, 0);
        if constexpr (DROP) dp = f32x16{};                   // (delta is subtracted behind the mask)
        else dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qaug, sel_d, f32x16{}, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int chunk = 2 * ks + hh;
          const bf16x8 qa = *(const bf16x8*)(sQ + r31 * 128 + ((chunk ^ attn_swz(r31)) << 4));
          const bf16x8 da = *(const bf16x8*)(sD + r31 * 128 + ((chunk ^ attn_swz(r31)) << 4));
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
        }
      };
      // P = exp2(S');  dS = P * dP'  ->  bf16 B operands of the gradient products
      auto soft = [&](int sub, f32x16& s, f32x16& dp) {
        [[maybe_unused]] const int qrow0 = qt * QT + sub * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pr = __builtin_amdgcn_exp2f(s[r]);
          if constexpr (DROP) {
            const unsigned int qq = (unsigned int)(qrow0 + (r & 3) + 8 * (r >> 2) + 4 * hh);
            const float mm = attn_drop_scale(akey, qq * (unsigned int)T + (unsigned int)key, dr.thresh, dr.inv_keep);
            const float dlr = sL0[128 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
            s[r] = pr * mm;                                  // the P that multiplies dO in dV: dropped and rescaled
            dp[r] = pr * (dp[r] * mm - dlr);
          } else {
            s[r] = pr;
            dp[r] = pr * dp[r];
          }
        }
      };
      // dV^T[d][key] += dO^T[d][q] . P[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key]   (k = q, accumulator row order)
      auto grads = [&](int sub, const f32x16& s, const f32x16& dp) {
        const char* sQ = sQ0 + sub * 32 * 128;
        const char* sD = sD0 + sub * 32 * 128;
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
            const int oa = ra * 128 + ((chunk ^ attn_swz(ra)) << 4) + (tp & 1) * 8;
            const int ob = rb * 128 + ((chunk ^ attn_swz(rb)) << 4) + (tp & 1) * 8;
            const bf16x4 da0 = lds_read_tr16(sD + oa), da1 = lds_read_tr16(sD + ob);
            const bf16x4 qa0 = lds_read_tr16(sQ + oa), qa1 = lds_read_tr16(sQ + ob);
            const bf16x8 dof = {da0[0], da0[1], da0[2], da0[3], da1[0], da1[1], da1[2], da1[3]};
            const bf16x8 qf = {qa0[0], qa0[1], qa0[2], qa0[3], qa1[0], qa1[1], qa1[2], qa1[3]};
            dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof, pf, dvt[db], 0, 0, 0);
            dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, dsf, dkt[db], 0, 0, 0);
          }
        }
      };
      constexpr bool PIPE = !DROP;                           // (the dropout variant's mask arithmetic leaves no registers for two score tiles)
      f32x16 sc[PIPE ? 2 : 1], dpc[PIPE ? 2 : 1];
      if constexpr (PIPE) scores(0, sc[0], dpc[0]);
#pragma unroll
      for (int sub = 0; sub < NSB; ++sub) {
        const bool more = sub + 1 < NSB && qt * QT + (sub + 1) * 32 < T;     // wave-uniform: the next sub-block holds rows of the sequence
        constexpr int kCurMask = PIPE ? 1 : 0;
        const int cur = sub & kCurMask;
        if constexpr (PIPE) {
          if (more) scores(sub + 1, sc[(sub + 1) & kCurMask], dpc[(sub + 1) & kCurMask]);
        } else {
          scores(sub, sc[0], dpc[0]);
        }
        if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
        soft(sub, sc[cur], dpc[cur]);
        if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
        grads(sub, sc[cur], dpc[cur]);
        if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
        if (!more) break;
      }
    }
    __syncthreads();
  }
  if (!active) return;
  const int kw = k0 + wave * 32;
  bf16* dk_rows = dqkv + ((size_t)b * T + kw) * ld_qkv + inner + head * 64;
  store_rows_t(dkt, dk_scale, smem + wave * 8192, dk_rows, (size_t)ld_qkv, T - kw, lane);
  store_rows_t(dvt, 1.0f, smem + wave * 8192 + 4096, dk_rows + inner, (size_t)ld_qkv, T - kw, lane);
}

// ------------------------------------------------------------------------------------------------ dQ
// Runs FIRST: it also produces delta[b][h][q] = sum_d dO * O for its own queries (the rows are in its registers anyway) and leaves
// it in memory for the dK/dV pass, so no separate row-sum kernel sits on the critical path.
template <int KB, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o_fwd, const bf16* __restrict__ d_o,
                                                          const float* __restrict__ lse, float* __restrict__ delta,
                                                          bf16* __restrict__ dqkv, int T, int H, int ld_qkv, int ld_o, float scale,
                                                          AttnDrop dr) {
  constexpr int NKB = KB / 32;
  constexpr int kTileBytes = KB * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 bufs][K tile | V tile]
  int bh, qblk;
  xcd_group_block(blockIdx.x, (T + 127) / 128, gridDim.x / ((T + 127) / 128), bh, qblk);
  const int b = bh / H, head = bh - b * H, q0 = qblk * 128;
  const int lane = lane_id(), wave = wave_id();
  const int r31 = lane & 31, hh = lane >> 5;
  const int inner = H * 64;
  const bf16* base = qkv + (size_t)b * T * ld_qkv + head * 64;
  const bool active = q0 + wave * 32 < T;
  const int q = q0 + wave * 32 + r31;
  const int qc = min(q, T - 1);
  bf16x8 qf[4], dof[4];
  float del = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = *(const bf16x8*)(base + (size_t)qc * ld_qkv + 16 * ks + 8 * hh);           // Q' = q * scale * log2(e)
    dof[ks] = *(const bf16x8*)(d_o + ((size_t)b * T + qc) * ld_o + head * 64 + 16 * ks + 8 * hh);
    const bf16x8 of = *(const bf16x8*)(o_fwd + ((size_t)b * T + qc) * ld_o + head * 64 + 16 * ks + 8 * hh);
#pragma unroll
    for (int j = 0; j < 8; ++j) del += (float)of[j] * (float)dof[ks][j];
  }
  del = half_sum(del);                                    // the two half-waves hold the two halves of the 64-wide row
  if (hh == 0 && q < T) delta[((size_t)b * H + head) * T + q] = del;
  [[maybe_unused]] unsigned int akey = 0u, qoff = 0u;
  if constexpr (DROP) {
    akey = attn_key(dr.seed + *dr.seed_ptr, b * H + head);
    qoff = (unsigned int)q * (unsigned int)T;
  }
  // constant side of the augmented MFMAs (this lane's query): [-lse*log2e pieces, -3e38, -delta pieces, 0]
  const float l2 = lse[((size_t)b * H + head) * T + qc] * 1.44269504088896340736f;
  const bf16x8 qaug = aug_const(l2, true, DROP ? 0.f : del, hh);
  [[maybe_unused]] const bf16x8 sel_d = aug_sel_second(hh);

  const int nkt = (T + KB - 1) / KB;
  const int nB = (int)gridDim.x / (((T + 127) / 128) * H);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, nB * T * ld_qkv * 2, 0x00020000);
  const int rsub = lane >> 3, slot = lane & 7;
  int vo[NKB], vo_last[NKB];
#pragma unroll
  for (int r = 0; r < NKB; ++r) {
    const int row = r * 32 + wave * 8 + rsub;
    const int col = inner + head * 64 + ((slot ^ attn_swz(row)) << 3);
    vo[r] = ((b * T + row) * ld_qkv + col) * 2;
    vo_last[r] = ((b * T + row - max((nkt - 1) * KB + row - (T - 1), 0)) * ld_qkv + col) * 2;
  }
  auto stage = [&](int buf, int kt) {
    char* sK = smem + buf * 2 * kTileBytes;
    char* sV = sK + kTileBytes;
#pragma unroll
    for (int r = 0; r < NKB; ++r) {
      const int v = (kt == nkt - 1) ? vo_last[r] : vo[r];
      const int so = kt * KB * ld_qkv * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (GVK_LDS void*)(sK + (r * 32 + wave * 8) * 128), 16, v, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (GVK_LDS void*)(sV + (r * 32 + wave * 8) * 128), 16, v, so + inner * 2, 0, 0);
    }
  };

  f32x16 dqt[2];
  dqt[0] = f32x16{};
  dqt[1] = f32x16{};
  const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) stage(buf ^ 1, kt + 1);
    if (active) {
      const char* sK = smem + buf * 2 * kTileBytes;
      const char* sV = sK + kTileBytes;
      // the same three-stage pipeline as the dK/dV pass, over the 32-key blocks of the tile:
      //   B(kb): S'^T, dP'^T (10 MFMAs)   C(kb): dS^T = exp2(S'^T) * dP'^T, bf16 (VALU)   D(kb): dQ^T += K^T.dS^T (4 MFMAs + transposed reads)
      auto scores = [&](int kb, f32x16& st, f32x16& dpt) {
        const int row = kb * 32 + r31;
        const bf16x8 sel_s = aug_sel_first(kt * KB + row >= T, hh);        // [1, 1, 1, key >= T, 0...]
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel_s, qaug, f32x16{}, 0, 0, 0);
        if constexpr (DROP) dpt = f32x16{};
        else dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sel_d, qaug, f32x16{}, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int chunk = 2 * ks + hh;
          const int off = row * 128 + ((chunk ^ attn_swz(row)) << 4);
          const bf16x8 ka = *(const bf16x8*)(sK + off);
          const bf16x8 va = *(const bf16x8*)(sV + off);
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[ks], st, 0, 0, 0);
          dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[ks], dpt, 0, 0, 0);
        }
      };
      auto soft = [&](int kb, f32x16& st, f32x16& dpt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float dpv = dpt[r];
          if constexpr (DROP) {
            const int key = kt * KB + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            dpv = dpv * attn_drop_scale(akey, qoff + (unsigned int)key, dr.thresh, dr.inv_keep) - del;
          }
          st[r] = __builtin_amdgcn_exp2f(st[r]) * dpv;
        }
      };
      auto grads = [&](int kb, const f32x16& st) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 dsf;
#pragma unroll
          for (int j = 0; j < 8; ++j) dsf[j] = (bf16)st[8 * s2 + j];
          const int key0 = kb * 32 + 16 * s2 + 4 * (g >> 1);
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const int chunk = db * 4 + 2 * (g & 1) + (tp >> 1);
            const int ra = key0 + tq, rb = key0 + 8 + tq;
            const bf16x4 ka0 = lds_read_tr16(sK + ra * 128 + ((chunk ^ attn_swz(ra)) << 4) + (tp & 1) * 8);
            const bf16x4 ka1 = lds_read_tr16(sK + rb * 128 + ((chunk ^ attn_swz(rb)) << 4) + (tp & 1) * 8);
            const bf16x8 kt8 = {ka0[0], ka0[1], ka0[2], ka0[3], ka1[0], ka1[1], ka1[2], ka1[3]};
            dqt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt8, dsf, dqt[db], 0, 0, 0);
          }
        }
      };
      f32x16 sc[2], dpc[2];
      scores(0, sc[0], dpc[0]);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) scores(kb + 1, sc[(kb + 1) & 1], dpc[(kb + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        soft(kb, sc[kb & 1], dpc[kb & 1]);
        __builtin_amdgcn_sched_barrier(0);
        grads(kb, sc[kb & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  if (!active) return;
  const int qw = q0 + wave * 32;
  store_rows_t(dqt, scale, smem + wave * 4096, dqkv + ((size_t)b * T + qw) * ld_qkv + head * 64, (size_t)ld_qkv, T - qw, lane);
}

template <int KB, bool DROP>
static int launch_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int T, int H,
                           int ld_qkv, int ld_out, float scale, AttnDrop dr, hipStream_t s) {
  const float dk_scale = 0.69314718055994530942f;      // dK = scale . dS^T.Q = dS^T.Q' / log2(e)
  const dim3 grid(((T + 127) / 128) * H * B);
  constexpr unsigned lds_kv = 2 * (2 * KB * 128 + 2 * 128 * 4), lds_q = 2 * 2 * KB * 128;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<KB, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_kernel<KB, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
    if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(attn_bwd): %s", hipGetErrorString(e));
    attr = true;
  }
  GVK_LAUNCH((attn_bwd_dq_kernel<KB, DROP>), grid, dim3(256), lds_q, s, (const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, delta, (bf16*)dqkv, T, H,
             ld_qkv, ld_out, scale, dr);
  int rc = check_launch("attention_bwd/dq");
  if (rc) return rc;
  GVK_LAUNCH((attn_bwd_dkdv_kernel<KB, DROP>), grid, dim3(256), lds_kv, s, (const bf16*)qkv, (const bf16*)dout, lse, (const float*)delta, (bf16*)dqkv, T, H,
             ld_qkv, ld_out, dk_scale, dr);
  return check_launch("attention_bwd/dkdv");
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
  GVK_REQUIRE((int64_t)B * T * ld_qkv * 2 < (int64_t)1 << 31, "gvk_attention_bwd_bf16: the qkv tensor must stay below 2 GiB (32-bit buffer offsets)");
  const AttnDrop dr{seed, (const unsigned long long*)seed_ptr, drop_threshold_u32(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f};
  hipStream_t s = (hipStream_t)stream;
  // tile of 96 rows when it pads the sequence less than 128 does (T = 1033: 1056 against 1152)
  int kb = ((T + 95) / 96 * 96 < (T + 127) / 128 * 128) ? 96 : 128;
  if (getenv("GAVIKO_HIP_ATTN_KB") && (atoi(getenv("GAVIKO_HIP_ATTN_KB")) == 96 || atoi(getenv("GAVIKO_HIP_ATTN_KB")) == 128)) kb = atoi(getenv("GAVIKO_HIP_ATTN_KB"));
  if (drop_p > 0.f)      // the dropout variants carry the mask arithmetic: 96-row tiles only (the 128-row form would spill registers)
    return launch_attn_bwd<96, true>(qkv, out, dout, lse, delta, dqkv, B, T, H, ld_qkv, ld_out, scale, dr, s);
  return kb == 96 ? launch_attn_bwd<96, false>(qkv, out, dout, lse, delta, dqkv, B, T, H, ld_qkv, ld_out, scale, dr, s)
                  : launch_attn_bwd<128, false>(qkv, out, dout, lse, delta, dqkv, B, T, H, ld_qkv, ld_out, scale, dr, s);
}

extern "C" int gvk_attention_bwd_bf16(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B,
                                      int T, int H, int ld_qkv, int ld_out, float scale, void* stream) {
  return gvk_attention_bwd_bf16_dropout(qkv, out, dout, lse, delta, dqkv, B, T, H, ld_qkv, ld_out, scale, 0.f, 0, nullptr, stream);
}
