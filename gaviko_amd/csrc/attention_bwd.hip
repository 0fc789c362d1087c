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
                                                            AttnDrop dr, int need_rows) {
  constexpr int NSB = QT / 32;
  constexpr int kTileQ = QT * 128;                // bytes of a [QT][64] bf16 tile
  constexpr int kBuf = 2 * kTileQ + 2 * 128 * 4;  // Q tile | dO tile | lse 128 f32 | delta 128 f32
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 bufs][kBuf]
  int bh, kblk;
  xcd_group_block(blockIdx.x, (T + 127) / 128, gridDim.x / ((T + 127) / 128), bh, kblk);   // all key blocks of a (batch, head) on one XCD
  const int b = bh / H, head = bh - b * H, k0 = kblk * 128;
  if (k0 >= need_rows) return;                     // (gvk_attention_bwd_bf16_rows: dk, dv of the first need_rows tokens only -- the whole workgroup leaves)
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
          }
        }
        if constexpr (!DROP) {
          // dS = P * dP' as explicit even-aligned pairs (v_pk_mul_f32), P first written in place: left to the SLP vectorizer, the dQ pass's
          // form `exp2(s) * dp` was paired across registers hipcc had allocated one off an even boundary -- 36 v_mov + 24 v_alignbit /
          // v_perm per 96-key tile to realign them, a quarter of that loop's vector instructions (203 -> 128 per tile and wave)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            typedef float f32x2_ __attribute__((ext_vector_type(2)));
            const f32x2_ pp = {s[r], s[r + 1]}, dd = {dp[r], dp[r + 1]};
            const f32x2_ o = pp * dd;
            dp[r] = o[0]; dp[r + 1] = o[1];
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
                                                          AttnDrop dr, int need_rows) {
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
  if (q0 >= need_rows) return;                            // (gvk_attention_bwd_bf16_rows: delta of every row, dq of the first need_rows tokens only)
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
          if constexpr (DROP) st[r] = __builtin_amdgcn_exp2f(st[r]) * dpv;
          else st[r] = __builtin_amdgcn_exp2f(st[r]);
        }
        if constexpr (!DROP) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            typedef float f32x2_ __attribute__((ext_vector_type(2)));
            const f32x2_ pp = {st[r], st[r + 1]}, dd = {dpt[r], dpt[r + 1]};
            const f32x2_ o = pp * dd;
            st[r] = o[0]; st[r + 1] = o[1];
          }
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

// ------------------------------------------------------------------------------------------------ one pass: dQ, dK, dV (round 5)
// The two passes above compute S and dP twice (seven MFMA products, every score exponentiated twice).  This kernel is the dK/dV pass
// extended by the fifth product: five products per (query, key) block, one exponential per score.
//   * workgroup = 128 keys of one (batch, head), exactly as the dK/dV pass (key on the lane; dK^T, dV^T of the wave's 32 keys stay in its
//     accumulators for the whole sweep);
//   * dS crosses LDS once per 32-query sub-block: every lane packs its accumulator registers 4g..4g+3 (queries 8g + 4h ..+3 of its key)
//     into one 8-byte unit of a [key][query] image (64-byte rows, units swizzled so that the writes and the transposed reads below are
//     conflict-free); after ONE barrier per sub-block the four waves each take a 16-wide d slab of dQ^T[d][q] = K^T.dS^T over all 128 keys:
//     eight v_mfma_f32_16x16x32_bf16 whose A operand (K^T of the slab, 16 registers) was gathered once and whose B operand comes
//     straight out of the image by ds_read_b64_tr_b16;
//   * dQ is summed over the key blocks of a (batch, head) by an ORDERED hand-off, no float atomics, bitwise reproducible: workgroup k
//     starts its sweep at query tile floor(k.nqt/nkb) and walks the tiles cyclically, so for every tile the nkb workgroups arrive at
//     distinct steps, one tile-time apart; the order of arrival IS the summation order (a function of (tile, k) alone).  The running
//     sum of a sub-block lives in one 8-KB scratch slab: a member waits until the slab's progress word equals its position, adds its
//     own 16 x 32 slab piece (16-byte sc1 loads, write-through sc1 stores), and the word is advanced behind the NEXT sub-block's
//     barrier, when every wave has drained its stores (s_waitcnt vmcnt(0)); the last member writes bf16 dQ and leaves the word at zero
//     for the next launch.  A member only ever waits for members at EARLIER steps, the workgroups of a group are consecutive in their
//     XCD's dispatch order, and a spin is bounded (status word), so a group that is only partly resident cannot hang the chip.
// delta = rowsum(dO o O) comes from a small kernel in front (attn_delta_kernel); the dropout variants keep the two-pass kernels.
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ o_fwd, const bf16* __restrict__ d_o, float* __restrict__ delta, int M,
                                                         int T, int H, int ld_o) {
  const int gid = blockIdx.x * 256 + threadIdx.x;          // ((row * H + head) * 8 + chunk)
  const int c = gid & 7, rh = gid >> 3;
  const int row = min(rh / H, M - 1), head = rh - (rh / H) * H;
  const size_t off = (size_t)row * ld_o + head * 64 + c * 8;
  const bf16x8 a = *(const bf16x8*)(o_fwd + off), b = *(const bf16x8*)(d_o + off);
  float v = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) v += (float)a[j] * (float)b[j];
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  if (c == 0 && rh < M * H) {
    const int bb = row / T, t = row - bb * T;
    delta[((size_t)bb * H + head) * T + t] = v;
  }
}

// byte offset of the 8-byte unit (key row 0..127, g = query group 0..3, h = half 0..1) of a dS^T image
__device__ __forceinline__ int ds_unit(int keyl, int g, int h) {
  return keyl * 64 + ((g ^ ((keyl >> 2) & 3)) << 4) + ((h ^ ((keyl >> 4) & 1)) << 3);
}

// VAR (measurement build only; results wrong unless 0): bit 0 = no waiting / no running-sum loads, bit 1 = no running-sum stores,
// bit 2 = no dQ product at all, bit 3 = write-through stores whatever the successor's XCD, bit 4 = no polling (sums loaded whatever their
// state), bit 5 = dQ product kept but nothing loaded or stored, bit 6 = shader-clock totals of the loop's phases, bit 7 = barrier without
// a drain of the vector-memory queue
// QT = queries per STEP (one staged tile, one barrier, one hand-off): 64 = two 32-query sub-blocks
template <int QT, int VAR>
__global__ __launch_bounds__(256, 2) void attn_bwd_fused_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                             const float* __restrict__ delta, bf16* __restrict__ dqkv, float* dq_acc, int* prog,
                                                             int* xcc_tab, int* status, int T, int H, int ld_qkv, int ld_o, float scale, float dk_scale) {
  constexpr int NSB = QT / 32;
  constexpr int kTileQ = QT * 128;                // bytes of a [QT][64] bf16 tile
  constexpr int kBuf = 2 * kTileQ + 2 * 128 * 4;  // Q tile | dO tile | lse 128 f32 | delta 128 f32
  constexpr int kDS = 128 * 64;                   // one dS^T image: [128 keys][32 queries] bf16
  constexpr int kImg = NSB * kDS;                 // the images of one step
  static_assert(2 * kImg >= 128 * 128, "the dS^T images first hold the K tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][kBuf] | [2][NSB][kDS] | start tiles of the key blocks (int [256])
  char* const sDSbase = smem + 2 * kBuf;      // (the images first hold the K tile, once)
  int* const stab = (int*)(sDSbase + 2 * kImg);
  const int nkb = (T + 127) / 128, nqt = (T + QT - 1) / QT;
  int bh, kblk;
  xcd_group_block(blockIdx.x, nkb, gridDim.x / nkb, bh, kblk);   // all key blocks of a (batch, head) on one XCD, consecutive in its dispatch order
  const int b = bh / H, head = bh - b * H, k0 = kblk * 128;
  const int lane = lane_id(), wave = wave_id();
  const int r31 = lane & 31, hh = lane >> 5;
  const int inner = H * 64;
  const bf16* qbase = qkv + (size_t)b * T * ld_qkv + head * 64;
  const bool active = k0 + wave * 32 < T;          // a wave whose 32 keys all lie past the sequence computes on copies of the last key and stores nothing

  // where this workgroup runs: the hand-off keeps a sum inside the XCD's L2 when producer and consumer share it (speed only; see finish)
  int my_xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(my_xcc));
  int* const xrow = xcc_tab + bh * nkb;
  if (threadIdx.x == 0) __hip_atomic_store((GVK_GLOBAL int*)(xrow + kblk), my_xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // K, V fragments of this wave's 32 keys: B operands (col = key, k = d)
  const int key = k0 + wave * 32 + r31;
  const int keyc = min(key, T - 1);
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = *(const bf16x8*)(qbase + inner + (size_t)keyc * ld_qkv + 16 * ks + 8 * hh);
    vf[ks] = *(const bf16x8*)(qbase + 2 * inner + (size_t)keyc * ld_qkv + 16 * ks + 8 * hh);
  }

  // staging: Q rows (from qkv), dO rows, lse / delta (one 4-byte LDS-DMA per wave: waves 0,1 the two 64-row halves of lse, waves 2,3 of delta)
  const int nB = (int)gridDim.x / (nkb * H);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, nB * T * ld_qkv * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)d_o, 0, nB * T * ld_o * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(wave < 2 ? lse : delta), 0, nB * H * T * 4, 0x00020000);
  // per (batch, head): one running-sum slab (NSB x 8 KB) and one progress word (on a 128-byte line of its own) per step
  const __amdgpu_buffer_rsrc_t racc = __builtin_amdgcn_make_buffer_rsrc((void*)(dq_acc + (size_t)bh * nqt * NSB * 2048), 0, nqt * NSB * 8192, 0x00020000);
  float* const accb = dq_acc + (size_t)bh * nqt * NSB * 2048;
  int* const progb = prog + (size_t)bh * nqt * 32;
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
    const int soq = qt * QT * ld_qkv * 2, sod = qt * QT * ld_o * 2, sol = qt * QT * 4;
#pragma unroll
    for (int r = 0; r < NSB; ++r) {
      const int over = max(qt * QT + r * 32 + wave * 8 + rsub - (T - 1), 0);          // rows past the sequence: copies of the last row (masked below)
      const int vq = voq[r] - over * ld_qkv * 2, vd = vod[r] - over * ld_o * 2;
      lds_dma16(rq, sQ + (r * 32 + wave * 8) * 128, vq, soq);
      lds_dma16(rd, sD + (r * 32 + wave * 8) * 128, vd, sod);
    }
    const int overl = max(qt * QT + lrow - (T - 1), 0);
    lds_dma4(rl, sL + (wave >> 1) * 512 + (wave & 1) * 256, vol - overl * 4, sol);
  };

  const int start = (kblk * nqt) / nkb;            // first tile of this workgroup's cyclic sweep
  if ((int)threadIdx.x < nkb) stab[threadIdx.x] = ((int)threadIdx.x * nqt) / nkb;     // every member's (one division each, here, instead of nkb per tile)
  // ---- prologue: the K tile of the workgroup through LDS (once), the first two query tiles
  GVK_LOADS_LANDED();
#pragma unroll
  for (int r = 0; r < 4; ++r) {                    // 128 key rows x 128 B into the (not yet used) dS^T images, rows past the sequence = copies of the last key
    const int row = r * 32 + wave * 8 + rsub;
    const int rowc = min(k0 + row, T - 1);
    lds_dma16(rq, sDSbase + (r * 32 + wave * 8) * 128, ((b * T + rowc) * ld_qkv + inner + head * 64 + ((slot ^ attn_swz(row)) << 3)) * 2, 0);
  }
  stage(0, start);
  GVK_DMA_DRAIN();
  __syncthreads();
  // K^T of this wave's d slab (16 wide) over all 128 keys: A operands of dQ^T = K^T.dS^T (row = d, k = key), by transposed reads of the
  // row-major tile; keys past the sequence are ZERO here, which is what keeps their (finite) dS out of dQ
  const int G = lane >> 4, li = lane & 15, tq = (lane & 15) >> 2, tp = lane & 3;
  bf16x8 ktf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int chunk = 2 * wave + (tp >> 1);
    const int ra = 32 * kk + 8 * G + tq, rb = ra + 4;
    const bf16x4 a0 = lds_read_tr16(sDSbase + ra * 128 + ((chunk ^ attn_swz(ra)) << 4) + (tp & 1) * 8);
    const bf16x4 a1 = lds_read_tr16(sDSbase + rb * 128 + ((chunk ^ attn_swz(rb)) << 4) + (tp & 1) * 8);
    ktf[kk] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  }
  if (k0 + 128 > T) {                               // (last key block only)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + 32 * kk + 8 * G + j >= T) ktf[kk][j] = (bf16)0.f;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): the fragments are in registers ...
  __syncthreads();                                 // ... in every wave, before the first dS^T image overwrites the K tile

  f32x16 dkt[2], dvt[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { dkt[i] = f32x16{}; dvt[i] = f32x16{}; }
  const bf16x8 sel_s = aug_sel_first(true, hh);
  const bf16x8 sel_d = aug_sel_second(hh);
  const int g = lane >> 4;
  const int keyl = wave * 32 + r31;                // this lane's key row in the dS^T images
  // transposed-read addresses of the dQ product inside an image: lane 4q'+p of group G supplies key row 32kk + 8G + 4rd + q', unit (2qh + (p>>1), p&1)
  int dsr[2][2];                                   // [qh][rd] for kk = 0; kk adds 32 rows = 2048 bytes (the swizzle terms repeat every 32 rows)
#pragma unroll
  for (int qh = 0; qh < 2; ++qh)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) dsr[qh][rd] = ds_unit(8 * G + 4 * rd + tq, 2 * qh + (tp >> 1), tp & 1);

  // ---- the sweep.  One iteration = one STEP of QT queries from barrier to barrier, software-pipelined so that no global latency and no
  // dependent MFMA chain sits alone on a wave's critical path:
  //     dK, dV of sub-block 0 (its scores came with the previous step); scores, dK, dV of the others; dS^T -> images   [drain, poll, barrier]
  //     [word of the previous step advanced]  [this step's sum and the next step's poll word requested]
  //     dQ products of the step  ||  S, dP of the next step's first sub-block (one basic block)   [add the sum, pass it on]   exponentials
  // position of this workgroup in a tile's summation chain = members that reach the tile at an earlier step; its successor = the next one
  auto chain = [&](int qt, int& pos, int& succ) {
    int mine = qt - start;
    if (mine < 0) mine += nqt;
    pos = 0;
    succ = -1;
    int best = 1 << 30;
    for (int k2 = 0; k2 < nkb; ++k2) {
      int st = qt - stab[k2];
      if (st < 0) st += nqt;
      pos += st < mine ? 1 : 0;
      if (st > mine && st < best) { best = st; succ = k2; }
    }
    pos = __builtin_amdgcn_readfirstlane(pos);
    succ = __builtin_amdgcn_readfirstlane(succ);
  };
  // S' = Q'.K^T - lse2 and dP' = dO.V^T - delta of sub-block (tile buffer, sub); then (soft) P = exp2(S'), dS = P.dP'
  f32x16 s, dp;
  auto scores = [&](int buf, int sub, int qrow0) {
    const char* sQ = smem + buf * kBuf + sub * 32 * 128;
    const char* sD = sQ + kTileQ;
    const float* sL0 = (const float*)(smem + buf * kBuf + 2 * kTileQ);
    const float l2 = sL0[sub * 32 + r31] * 1.44269504088896340736f;
    const float dl = sL0[128 + sub * 32 + r31];
    const bf16x8 qaug = aug_const(l2, qrow0 + r31 >= T, dl, hh);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qaug, sel_s, f32x16{}, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qaug, sel_d, f32x16{}, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = 2 * ks + hh;
      const bf16x8 qa = *(const bf16x8*)(sQ + r31 * 128 + ((chunk ^ attn_swz(r31)) << 4));
      const bf16x8 da = *(const bf16x8*)(sD + r31 * 128 + ((chunk ^ attn_swz(r31)) << 4));
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
    }
  };
  auto soft = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = __builtin_amdgcn_exp2f(s[r]);
      s[r] = pr;
      dp[r] = pr * dp[r];
    }
  };
  // dV^T[d][key] += dO^T[d][q] . P[q][key] ;  dK^T[d][key] += Q^T[d][q] . dS[q][key];  dS^T -> the sub-block's image
  auto grads = [&](int buf, int sub, char* sDS) {
    const char* sQ = smem + buf * kBuf + sub * 32 * 128;
    const char* sD = sQ + kTileQ;
#pragma unroll
    for (int sk = 0; sk < 2; ++sk) {
      bf16x8 pf, dsf;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) { pf[jj] = (bf16)s[8 * sk + jj]; dsf[jj] = (bf16)dp[8 * sk + jj]; }
      const u32x4 dsw = __builtin_bit_cast(u32x4, dsf);
      *(u32x2*)(sDS + ds_unit(keyl, 2 * sk, hh)) = u32x2{dsw[0], dsw[1]};
      *(u32x2*)(sDS + ds_unit(keyl, 2 * sk + 1, hh)) = u32x2{dsw[2], dsw[3]};
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

  int pend_word = -1, pend_val = 0;                // progress word to advance behind the next barrier
  // a step's share of dQ (NSB sub-blocks x 2 query halves of this wave's 16-d slab): add the predecessor's sum (requested behind the
  // barrier, consumed here a whole product later), pass it on
  auto finish = [&](f32x4 (&hdq)[NSB][2], const u32x4 (&hld)[NSB][2], int qt, int ppos, bool add, bool last, bool plain) {
    const int aoff = qt * NSB * 8192 + wave * 2048 + lane * 16;
    {   // (a select, not a branch: the requested sum is always consumed, so its registers are never overwritten while in flight -- the
        //  compiler answers that hazard with a full s_waitcnt vmcnt(0), which would also wait for the stores below)
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) hdq[sb][qh] += add ? __builtin_bit_cast(f32x4, hld[sb][qh]) : z;
    }
    if (last) {
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
          const int q = qt * QT + sb * 32 + 16 * qh + li;
          const bf16x4 o = {(bf16)(hdq[sb][qh][0] * scale), (bf16)(hdq[sb][qh][1] * scale), (bf16)(hdq[sb][qh][2] * scale), (bf16)(hdq[sb][qh][3] * scale)};
          if (q < T) *(bf16x4*)(dqkv + ((size_t)b * T + q) * ld_qkv + head * 64 + 16 * wave + 4 * G) = o;
        }
    } else if (plain) {
      // the successor runs on this XCD: plain stores leave the sum in the shared L2, where its L1-bypassing loads find it (a write-through
      // store would drop the lines from the L2 and send every one of those loads to the fabric)
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) *(f32x4*)((char*)accb + aoff + sb * 8192 + qh * 1024) = hdq[sb][qh];
    } else {
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hdq[sb][qh]), racc, aoff + sb * 8192 + qh * 1024, 0, 16);   // aux 16 = sc1
    }
    pend_word = qt;
    pend_val = (last || (VAR & 39) != 0) ? 0 : ppos + 1;      // (the timing ablations never leave a progress word set)
  };

  int qt = start, pos, succ;
  chain(qt, pos, succ);
  if constexpr ((VAR & 1) != 0) pos = 0;
  // every member's XCD + 1, lane k2 = member k2 (0: had not started when this was read, a few microseconds into the kernel -> write-through
  // towards it).  Read ONCE: a load inside the loop whose result is carried around it makes the compiler drain the queue (WAW on the
  // carried register) in every iteration.
  const int xall = __hip_atomic_load((GVK_GLOBAL int*)(xrow + min(lane, nkb - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0x0F70);              // (vmcnt(0), visible to the compiler: otherwise its wait for xall lands inside the loop)
  int xs = succ >= 0 && succ < 64 ? __builtin_amdgcn_readlane(xall, succ & 63) : 0;      // the successor's (key blocks past 64: unknown)
  int pv = 0;                                      // progress word of the current step (wave 0 only), requested one step ahead
  if (nqt > 1) stage(1, start + 1 < nqt ? start + 1 : 0);
  scores(0, 0, start * QT);
  soft();
  int par = 0;
  [[maybe_unused]] unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;     // VAR bit 6: shader-clock totals of the loop's phases (a few workgroups)
#define GVK_PH(k)                                                             \
  if constexpr ((VAR & 64) != 0) {                                            \
    const unsigned long long tnow = __builtin_amdgcn_s_memtime();             \
    if (j >= 2 && j < nqt - 2) ph[k] += tnow - tprev;                         \
    tprev = tnow;                                                             \
  }
  for (int j = 0; j < nqt; ++j) {
    const int buf = j & 1;
    const bool lastm = (VAR & 2) ? true : pos == nkb - 1;
    char* sDS = sDSbase + par * kImg;
    GVK_PH(5)
    grads(buf, 0, sDS);
#pragma unroll
    for (int sb = 1; sb < NSB; ++sb) {
      scores(buf, sb, qt * QT + sb * 32);
      soft();
      grads(buf, sb, sDS + sb * kDS);
    }
    GVK_PH(1)
    // drain: this wave's stores of the previous step have been acknowledged, the poll word and the next tile's rows have arrived
    if constexpr ((VAR & 128) == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv) : : "memory");
    GVK_PH(2)
    if constexpr ((VAR & 16) == 0) {
      if (wave == 0 && pos > 0 && pv != pos) {     // (rare) the predecessor's sum is not complete yet: wave 0 waits in front of the barrier for all
        for (unsigned spins = 0;; ++spins) {
          __builtin_amdgcn_s_sleep(4);
          pv = __hip_atomic_load((GVK_GLOBAL int*)(progb + qt * 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (pv == pos) break;
          if (spins > (1u << 16)) { if (lane == 0) atomicAdd(status, 1); break; }     // (never seen: bounded so that a broken chain ends the launch)
        }
#ifdef GVK_DIAG
        if (lane == 0) atomicAdd(status + 1, 1);            // waits that were not satisfied by the early read
#endif
      }
    }
    if constexpr ((VAR & 128) != 0) {             // (timing ablation: a barrier that does not wait for the vector-memory queue)
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();
    }
    GVK_PH(3)
    if (pend_word >= 0 && threadIdx.x == 0)
      __hip_atomic_store((GVK_GLOBAL int*)(progb + pend_word * 32), pend_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pend_word = -1;
    // this step's place in its chain; the predecessor's sum is requested here and added behind the products below
    const int h_qt = qt, h_pos = pos;
    const bool h_last = lastm, h_add = pos > 0, h_plain = (VAR & 8) == 0 && xs == my_xcc + 1;
    f32x4 hdq[NSB][2];
    u32x4 hld[NSB][2];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) hld[sb][qh] = u32x4{0u, 0u, 0u, 0u};
    if constexpr ((VAR & 37) == 0) {               // (unconditional: a load under a branch would leave registers in flight on the other path)
      const int aoff = qt * NSB * 8192 + wave * 2048 + lane * 16;
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh)
          hld[sb][qh] = __builtin_amdgcn_raw_buffer_load_b128(racc, aoff + sb * 8192 + qh * 1024, 0, 16);     // aux 16 = sc1: past this CU's L1, served by the L2
    }
    // the next step's tile (cyclic); the tile after it is requested into the buffer just left
    int qtn = qt + 1 < nqt ? qt + 1 : 0;
    if (j + 1 < nqt) {
      if (j + 2 < nqt) stage(buf, qtn + 1 < nqt ? qtn + 1 : 0);
      chain(qtn, pos, succ);
      if constexpr ((VAR & 1) != 0) pos = 0;
      xs = succ >= 0 && succ < 64 ? __builtin_amdgcn_readlane(xall, succ & 63) : 0;
    } else {
      qtn = qt;                                    // (last step: the scores below are recomputed and unused)
    }
    const int bufn = j + 1 < nqt ? buf ^ 1 : buf;
    if constexpr ((VAR & 16) == 0) {
      if (wave == 0) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(pv) : "v"(0), "s"(progb + qtn * 32) : "memory");
    }
    GVK_PH(4)
    // dQ^T[16 d of this wave][QT q] of THIS step over the workgroup's 128 keys, in one block with the first scores of the NEXT one
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      hdq[sb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      hdq[sb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr ((VAR & 4) == 0) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int qh = 0; qh < 2; ++qh) {
            const char* im = sDS + sb * kDS + kk * 2048;
            const bf16x4 b0 = lds_read_tr16(im + dsr[qh][0]), b1 = lds_read_tr16(im + dsr[qh][1]);
            const bf16x8 bb = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            hdq[sb][qh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[kk], bb, hdq[sb][qh], 0, 0, 0);
          }
      }
    }
    scores(bufn, 0, qtn * QT);
    GVK_PH(0)
    // between the two halves of the next scores: the requested sum has had the products above to arrive, the stores have the exponentials
    // and the next step's dK / dV products to be acknowledged before the drain in front of its barrier
    if constexpr ((VAR & 4) == 0 && (VAR & 32) == 0) finish(hdq, hld, h_qt, h_pos, h_add, h_last, h_plain);
    if constexpr ((VAR & 32) != 0) {               // (dQ product kept alive, nothing loaded or stored)
      if (hdq[0][0][0] + hdq[NSB - 1][1][3] == 123.456f) atomicAdd(status + 3, 1);
    }
    soft();
    qt = qtn;
    par ^= 1;
  }
#undef GVK_PH
  if constexpr ((VAR & 64) != 0) {
    if (lane == 0 && (blockIdx.x == 3 || blockIdx.x == 100 || blockIdx.x == 259))
      for (int k = 0; k < 6; ++k) ((unsigned long long*)(status + 16))[(blockIdx.x == 3 ? 0 : blockIdx.x == 100 ? 1 : 2) * 32 + wave * 8 + k] = ph[k];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (pend_word >= 0) __hip_atomic_store((GVK_GLOBAL int*)(progb + pend_word * 32), pend_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((GVK_GLOBAL int*)(xrow + kblk), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // nobody asks for this workgroup's XCD any more
  }
  if (!active) return;
  const int kw = k0 + wave * 32;
  bf16* dk_rows = dqkv + ((size_t)b * T + kw) * ld_qkv + inner + head * 64;
  store_rows_t(dkt, dk_scale, smem + wave * 8192, dk_rows, (size_t)ld_qkv, T - kw, lane);
  store_rows_t(dvt, 1.0f, smem + wave * 8192 + 4096, dk_rows + inner, (size_t)ld_qkv, T - kw, lane);
}

template <int KB, bool DROP>
static int launch_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int T, int H,
                           int ld_qkv, int ld_out, float scale, AttnDrop dr, hipStream_t s, int need_rows = 1 << 30) {
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
             ld_qkv, ld_out, scale, dr, need_rows);
  int rc = check_launch("attention_bwd/dq");
  if (rc) return rc;
  GVK_LAUNCH((attn_bwd_dkdv_kernel<KB, DROP>), grid, dim3(256), lds_kv, s, (const bf16*)qkv, (const bf16*)dout, lse, (const float*)delta, (bf16*)dqkv, T, H,
             ld_qkv, ld_out, dk_scale, dr, need_rows);
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

// workspace of the one-pass backward: [progress words: cap lines of 128 B, one word each | XCD table: cap int32, together padded to 256 B | status word, 256 B | running
// dQ sums: cap slabs of 8 KB].
// The layout follows from the workspace SIZE alone (cap = slabs it can hold), never from T: a model whose layers run different sequence
// lengths through one workspace (deep VPT) then keeps its progress words in one place, where every launch leaves them zero.
static constexpr int kFusedQT = 64;                 // queries per step of the sweep: one staged tile, one barrier, one hand-off (two 32-query sub-blocks)
static size_t fused_slabs(int B, int T, int H) { return (size_t)B * H * ((T + kFusedQT - 1) / kFusedQT) * (kFusedQT / 32); }
static size_t fused_ws_cap(size_t ws_bytes) { return ws_bytes < 1024 ? 0 : (ws_bytes - 1024) / (8192 + 128 + 4); }
static size_t fused_status_off(size_t ws_bytes) { return (fused_ws_cap(ws_bytes) * 132 + 255) / 256 * 256; }

extern "C" size_t gvk_attention_bwd_ws_bytes(int B, int T, int H) {
  if (B <= 0 || T <= 0 || H <= 0) return 0;
  return fused_slabs(B, T, H) * (8192 + 128 + 4) + 1024;
}
extern "C" size_t gvk_attention_bwd_status_offset(size_t ws_bytes) { return fused_status_off(ws_bytes); }

extern "C" int gvk_attention_bwd_bf16_fused(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, void* ws,
                                            size_t ws_bytes, int B, int T, int H, int ld_qkv, int ld_out, float scale, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(qkv && out && dout && lse && delta && dqkv && ws, "gvk_attention_bwd_bf16_fused: null pointer");
  GVK_REQUIRE(B > 0 && T > 0 && H > 0, "gvk_attention_bwd_bf16_fused: empty shape");
  GVK_REQUIRE(ld_qkv >= 3 * H * 64 && ld_qkv % 8 == 0 && ld_out >= H * 64 && ld_out % 8 == 0,
              "gvk_attention_bwd_bf16_fused: head dim is fixed at 64; ld_qkv=%d ld_out=%d inconsistent with H=%d", ld_qkv, ld_out, H);
  GVK_REQUIRE((int64_t)B * T * ld_qkv * 2 < (int64_t)1 << 31, "gvk_attention_bwd_bf16_fused: the qkv tensor must stay below 2 GiB (32-bit buffer offsets)");
  const size_t need = gvk_attention_bwd_ws_bytes(B, T, H);
  GVK_REQUIRE(ws_bytes >= need && ((uintptr_t)ws & 255) == 0, "gvk_attention_bwd_bf16_fused: workspace of %zu bytes (256-byte aligned) needed, %zu given", need, ws_bytes);
  GVK_REQUIRE(fused_ws_cap(ws_bytes) >= fused_slabs(B, T, H), "gvk_attention_bwd_bf16_fused: workspace layout cannot hold %zu slabs", fused_slabs(B, T, H));
  const size_t off_status = fused_status_off(ws_bytes), off_acc = off_status + 256;
  const size_t nsub = (size_t)((T + kFusedQT - 1) / kFusedQT) * (kFusedQT / 32);
  GVK_REQUIRE(nsub * 8192 < ((size_t)1 << 31) && (T + 127) / 128 <= 256, "gvk_attention_bwd_bf16_fused: sequence too long (32-bit slab offsets, 256 key blocks)");
  hipStream_t s = (hipStream_t)stream;
  constexpr unsigned lds = 2 * (2 * kFusedQT * 128 + 2 * 128 * 4) + 2 * (kFusedQT / 32) * 128 * 64 + 1024;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<kFusedQT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#ifdef GVK_DIAG
    auto set1 = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds); };
    set1(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<kFusedQT, 1>)); set1(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<kFusedQT, 3>));
    set1(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<kFusedQT, 7>)); set1(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<kFusedQT, 8>));
    set1(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<kFusedQT, 11>)); set1(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<kFusedQT, 15>));
    set1(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<kFusedQT, 16>)); set1(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<kFusedQT, 64>)); set1(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<kFusedQT, 33>)); set1(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<kFusedQT, 129>)); set1(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<kFusedQT, 144>));
#endif
    if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(attn_bwd_fused): %s", hipGetErrorString(e));
    attr = true;
  }
  const int M = B * T;
  GVK_LAUNCH(attn_delta_kernel, dim3((unsigned)(((size_t)M * H * 8 + 255) / 256)), dim3(256), 0, s, (const bf16*)out, (const bf16*)dout, delta, M, T, H, ld_out);
  int rc = check_launch("attention_bwd/delta");
  if (rc) return rc;
  const float dk_scale = 0.69314718055994530942f;      // dK = scale . dS^T.Q = dS^T.Q' / log2(e)
  char* w = (char*)ws;
  const dim3 grid(((T + 127) / 128) * H * B);
#define GVK_FUSED(V)                                                                                                                          \
  GVK_LAUNCH((attn_bwd_fused_kernel<kFusedQT, V>), grid, dim3(256), lds, s, (const bf16*)qkv, (const bf16*)dout, lse, (const float*)delta, \
             (bf16*)dqkv, (float*)(w + off_acc), (int*)w, (int*)w + fused_ws_cap(ws_bytes) * 32, (int*)(w + off_status), T, H, ld_qkv, ld_out, scale, dk_scale)
#ifdef GVK_DIAG
  const int var = diag_env("GAVIKO_HIP_ATTN_VAR") ? atoi(diag_env("GAVIKO_HIP_ATTN_VAR")) : 0;      // timing ablations (wrong results)
  if (var == 1) GVK_FUSED(1); else if (var == 3) GVK_FUSED(3); else if (var == 7) GVK_FUSED(7); else if (var == 8) GVK_FUSED(8);
  else if (var == 11) GVK_FUSED(11); else if (var == 15) GVK_FUSED(15); else if (var == 16) GVK_FUSED(16); else if (var == 64) GVK_FUSED(64); else if (var == 33) GVK_FUSED(33); else if (var == 129) GVK_FUSED(129); else if (var == 144) GVK_FUSED(144); else
#endif
  GVK_FUSED(0);
#undef GVK_FUSED
  return check_launch("attention_bwd/fused");
}

extern "C" int gvk_attention_bwd_bf16_rows(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B,
                                           int T, int H, int ld_qkv, int ld_out, float scale, int need_rows, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(qkv && out && dout && lse && delta && dqkv, "gvk_attention_bwd_bf16_rows: null pointer");
  GVK_REQUIRE(B > 0 && T > 0 && H > 0 && need_rows > 0, "gvk_attention_bwd_bf16_rows: empty shape");
  GVK_REQUIRE(ld_qkv >= 3 * H * 64 && ld_qkv % 8 == 0 && ld_out >= H * 64 && ld_out % 8 == 0,
              "gvk_attention_bwd_bf16_rows: head dim is fixed at 64; ld_qkv=%d ld_out=%d inconsistent with H=%d", ld_qkv, ld_out, H);
  GVK_REQUIRE((int64_t)B * T * ld_qkv * 2 < (int64_t)1 << 31, "gvk_attention_bwd_bf16_rows: the qkv tensor must stay below 2 GiB (32-bit buffer offsets)");
  const AttnDrop dr{0, nullptr, 0u, 1.f};
  const int kb = ((T + 95) / 96 * 96 < (T + 127) / 128 * 128) ? 96 : 128;        // the same tile choice as gvk_attention_bwd_bf16: the same bits
  return kb == 96 ? launch_attn_bwd<96, false>(qkv, out, dout, lse, delta, dqkv, B, T, H, ld_qkv, ld_out, scale, dr, (hipStream_t)stream, need_rows)
                  : launch_attn_bwd<128, false>(qkv, out, dout, lse, delta, dqkv, B, T, H, ld_qkv, ld_out, scale, dr, (hipStream_t)stream, need_rows);
}

extern "C" int gvk_attention_bwd_bf16(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B,
                                      int T, int H, int ld_qkv, int ld_out, float scale, void* stream) {
  return gvk_attention_bwd_bf16_dropout(qkv, out, dout, lse, delta, dqkv, B, T, H, ld_qkv, ld_out, scale, 0.f, 0, nullptr, stream);
}
