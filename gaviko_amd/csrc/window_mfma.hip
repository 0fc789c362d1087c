// MWSA window attention (gaviko.py:235-241) on the fp32 matrix cores, L = 20: exact fp32 products, fp32 accumulation.
//
// The row-per-wave kernels of window_attn.hip touch only the live keys of each query, but every (query, key) pair costs two 80-byte
// gathers from L2 and 40-60 scalar FMAs: 0.86 M pairs per sample and pass, ~12 TB/s of L2 traffic while they run -- traffic the backbone
// GEMMs and the flash-attention kernels running beside them on the other streams pay for (removing the window
// kernels alone returned 4.5 % of the step).  Here the window is a MASK over dense 16 x 16 tiles instead:
//
//   workgroup = 16 consecutive tokens (queries; keys in the key-side backward) of one sample, four waves; wave w takes every fourth
//   16-token partner block of the d-planes the block's windows reach (6-7 planes x 100 tokens at local_k = 6,6,6: ~40 blocks, of which
//   the 216-key windows fill about a third -- the matrix cores do not care);
//   S^T[key][query] = K . Q^T is five v_mfma_f32_16x16x4_f32 (L = 20 = 5 x 4) with both operands read straight from global memory as
//   one 16-byte + one 4-byte load per lane (the latent axis is the MFMA K axis, taken in the permuted order [4g+i | 16+g] so that a lane's
//   float4 IS its four k-steps); the window test is index arithmetic on the lane's four consecutive keys; the accumulator layout
//   (lane = query column, registers = four keys) is exactly the B-operand layout of the second product O^T = V^T . P^T, so the
//   probabilities never move -- online softmax per query column, one 2-step cross-group max per tile.
//   The waves' partial (max, sum, O^T) are merged through LDS in a fixed order: deterministic, no atomics.  The operands of the next
//   partner block are requested before the current one is worked on (two waves per SIMD at most: nothing else hides the L2 latency).
//
// Per pass each token row is now read once per partner BLOCK (16 queries share it): ~10x less L2 traffic, ~40x fewer VALU FMAs.
// Backward: a query-side kernel (delta, dq) and a key-side kernel over the reverse window (dk, dv), same tiling.
#include "common.hpp"
#include "window_args.hpp"

namespace gvk {
namespace {

constexpr int kL = 20;            // latent width this file is built for
constexpr int kRow = 3 * kL;      // floats per token row of q | k | v
constexpr int kSplit = 4;         // waves per 16-token block (in the step: 4 -> 714, 8 -> 710, 16 -> 702, 2 -> 668 volumes/s; isolated 8 is the fastest)

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// reductions over the four 16-lane groups of a wave (same lane % 16)
__device__ __forceinline__ float xsum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float xmax(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }

struct Tok { int d, h, w; };
__device__ __forceinline__ Tok decode(int idx, int H, int W) {
  Tok t;
  const int hw = H * W;
  t.d = idx / hw;
  const int r = idx - t.d * hw;
  t.h = r / W;
  t.w = r - t.h * W;
  return t;
}
__device__ __forceinline__ void next(Tok& t, int H, int W) {
  if (++t.w == W) { t.w = 0; if (++t.h == H) { t.h = 0; ++t.d; } }
}
struct Box {                      // half-open per axis
  int d0, d1, h0, h1, w0, w1;
  __device__ __forceinline__ bool has(const Tok& t) const { return t.d >= d0 && t.d < d1 && t.h >= h0 && t.h < h1 && t.w >= w0 && t.w < w1; }
};
__device__ __forceinline__ Box fwd_box(const Tok& q, const WinArgs& p) {          // keys a query sees
  Box b; int n;
  axis_fwd(q.d, p.kd, p.D, b.d0, n); b.d1 = b.d0 + n;
  axis_fwd(q.h, p.kh, p.H, b.h0, n); b.h1 = b.h0 + n;
  axis_fwd(q.w, p.kw, p.W, b.w0, n); b.w1 = b.w0 + n;
  return b;
}
__device__ __forceinline__ Box rev_box(const Tok& k, const WinArgs& p) {          // queries that see a key
  Box b; int n;
  axis_rev(k.d, p.kd, p.D, b.d0, n); b.d1 = b.d0 + n;
  axis_rev(k.h, p.kh, p.H, b.h0, n); b.h1 = b.h0 + n;
  axis_rev(k.w, p.kw, p.W, b.w0, n); b.w1 = b.w0 + n;
  return b;
}

// 16 tokens as the rows (A) or columns (B) of a tile, the latent axis as the MFMA K axis: k-step i < 4 carries latent 4g + i (g = lane / 16),
// step 4 carries latent 16 + g.  sec = the lane's token row + section offset (16-byte aligned).
struct RowOp { f32x4 v; float x; };
__device__ __forceinline__ RowOp load_rowop(const float* sec, int g) {
  RowOp r;
  r.v = *(const f32x4*)(sec + 4 * g);
  r.x = sec[16 + g];
  return r;
}
// C[m][n] = sum_l a(token m)[l] b(token n)[l]; the lane ends up with C[4g + i][lane % 16], i = 0..3
__device__ __forceinline__ f32x4 dot_tile(const RowOp& a, const RowOp& b) {
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) c = mfma4(a.v[i], b.v[i], c);
  return mfma4(a.x, b.x, c);
}
// 16 tokens as the K axis of the second product, latents as its rows: step i carries token tok0 + 4g + i; tile 0 = latents 0..15
// (row = lane % 16), tile 1 = latents 16..19 (rows >= 4 are zero).  `stride` floats between token rows, `n` = tokens of the sample.
struct ColOp { float t0[4], t1[4]; };
__device__ __forceinline__ ColOp load_colop(const float* sec0, int stride, int tok0, int n, int g, int l16) {
  ColOp c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float* r = sec0 + (size_t)min(tok0 + 4 * g + i, n - 1) * stride;
    c.t0[i] = r[l16];
    c.t1[i] = l16 < kL - 16 ? r[16 + l16] : 0.f;
  }
  return c;
}
// acc[latent][n] += sum_i col(token i)[latent] * w[i]  (w = the lane's four values of the first product: no data movement)
__device__ __forceinline__ void col_acc(const ColOp& c, const f32x4 w, f32x4& a0, f32x4& a1) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a0 = mfma4(c.t0[i], w[i], a0);
    a1 = mfma4(c.t1[i], w[i], a1);
  }
}

__device__ __forceinline__ float keep_scale(const WinArgs& p, unsigned long long row, int N, int j) {
  return hash_u32_w(p.seed, row * N + j) >= p.drop_thresh ? p.inv_keep : 0.f;
}

// partner-block range [b0, b1) of a 16-token block whose tokens span planes dfirst..dlast, given the per-axis reach below / above
__device__ __forceinline__ void block_range(int dfirst, int dlast, int below, int above, int D, int HW, int N, int& b0, int& b1) {
  const int dlo = max(0, dfirst - below), dhi = min(D - 1, dlast + above);
  b0 = (dlo * HW) >> 4;
  b1 = (min((dhi + 1) * HW, N) + 15) >> 4;
}

// ---- round 4.  (1) Token coordinates (d, h, w) come from a table in LDS, filled once per workgroup: decode() is two integer divisions
// by run-time values, ~50 VALU instructions per iteration.  (2) The key side of the backward no longer reads the query side's delta:
// delta_i = dctx_i . ctx_i comes from the rows it loads anyway, and the four (lse, delta) a lane needs are lane shuffles of its own
// row's values instead of eight more loads -- the two backward kernels are independent launches.  Together: 740 -> 747 volumes/s.
// Measured on the same box and NOT adopted (round 4, three interleaved runs each; the variants are no longer in the source):
//   * both backward sides in ONE launch of 504 workgroups: the kernels' summed time drops from 70 to 37 us per layer and the STEP gets
//     2 % slower (729 vs 747) -- twice the waves resident beside the flash-attention backward on the main stream;
//   * partner blocks prefetched three / two ahead instead of one (all loads unconditional with clamped indices so that vmcnt is counted
//     exactly): 741 vs 747 -- 146 registers instead of ~100 cost more than the hidden L2 latency returns.
// Once more: what the backbone pays for is the side kernels' resident registers x time, not their duration.
constexpr int kRing = 1;                // partner blocks in flight ahead of the one being multiplied

__device__ __forceinline__ int pack_tok(const Tok& t) { return t.d | (t.h << 10) | (t.w << 20); }
__device__ __forceinline__ Tok unpack_tok(int v) { Tok t; t.d = v & 1023; t.h = (v >> 10) & 1023; t.w = v >> 20; return t; }
// coordinates of every token of a sample, one packed word each; entries N.. (up to the next multiple of 16) repeat the last token
__device__ __forceinline__ void fill_tok_table(int* tab, int N, int H, int W, int nthreads) {
  const int n16 = (N + 15) & ~15;
  for (int i = threadIdx.x; i < n16; i += nthreads) tab[i] = pack_tok(decode(min(i, N - 1), H, W));
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(64 * kSplit) void win_mfma_fwd_kernel(WinArgs p) {
  extern __shared__ __attribute__((aligned(16))) int tokc[];
  __shared__ float sm_m[kSplit][16], sm_l[kSplit][16], sm_o[kSplit][kL][16];
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  const int HW = p.H * p.W, N = p.D * HW, nblk = (N + 15) >> 4;
  const int b = blockIdx.x / nblk, qb = blockIdx.x - b * nblk;
  const int wave = wave_id(), lane = lane_id(), g = lane >> 4, l16 = lane & 15;
  const float* base = p.qkv + (size_t)b * N * kRow;
  const int q = min(qb * 16 + l16, N - 1);
  const unsigned long long row = (unsigned long long)b * N + q;
  RowOp Q = load_rowop(base + (size_t)q * kRow, g);
  Q.v *= p.scale; Q.x *= p.scale;
  int kb0, kb1;
  block_range((qb * 16) / HW, min(qb * 16 + 15, N - 1) / HW, p.kd / 2, p.kd - 1 - p.kd / 2, p.D, HW, N, kb0, kb1);
  struct In { RowOp K; ColOp V; };
  auto fetch = [&](int kb) {
    In r;
    const int k0 = min(kb, kb1 - 1) * 16;                 // (past the range: re-reads the last block, never used)
    r.K = load_rowop(base + (size_t)min(k0 + l16, N - 1) * kRow + kL, g);
    r.V = load_colop(base + 2 * kL, kRow, k0, N, g, l16);
    return r;
  };
  In ring[kRing];
#pragma unroll
  for (int r = 0; r < kRing; ++r) ring[r] = fetch(kb0 + wave + r * kSplit);
  fill_tok_table(tokc, N, p.H, p.W, 64 * kSplit);
  const Box bx = fwd_box(unpack_tok(tokc[q]), p);
  float m = -INFINITY, l = 0.f;
  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
  auto step = [&](const In& in, int kb) {
    if (kb >= kb1) return;                                 // wave-uniform
    const int key0 = kb * 16;
    const f32x4 s = dot_tile(in.K, Q);                    // [key0 + 4g + i][query]
    const int4 tk4 = *(const int4*)(tokc + key0 + 4 * g);
    const int tkv[4] = {tk4.x, tk4.y, tk4.z, tk4.w};
    float sv[4], mb = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = key0 + 4 * g + i < N && bx.has(unpack_tok(tkv[i]));
      sv[i] = ok ? s[i] : -INFINITY;
      mb = fmaxf(mb, sv[i]);
    }
    const float mn = fmaxf(m, xmax(mb));
    const float ms = mn == -INFINITY ? 0.f : mn;         // a tile wholly outside this query's window leaves everything at zero
    const float corr = __expf(m - ms);
    l *= corr; o0 *= corr; o1 *= corr;
    f32x4 pv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float e = __expf(sv[i] - ms);
      l += e;
      pv[i] = p.drop_thresh != 0u ? e * keep_scale(p, row, N, key0 + 4 * g + i) : e;
    }
    m = mn;
    col_acc(in.V, pv, o0, o1);
  };
  for (int kb = kb0 + wave; kb < kb1; kb += kRing * kSplit) {
#pragma unroll
    for (int r = 0; r < kRing; ++r) {
      step(ring[r], kb + r * kSplit);
      ring[r] = fetch(kb + (kRing + r) * kSplit);
    }
  }
  l = xsum(l);
  if (g == 0) { sm_m[wave][l16] = m; sm_l[wave][l16] = l; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sm_o[wave][4 * g + i][l16] = o0[i];
    if (g == 0) sm_o[wave][16 + i][l16] = o1[i];
  }
  __syncthreads();
  if (wave != 0) return;
  float M = sm_m[0][l16];
#pragma unroll
  for (int w = 1; w < kSplit; ++w) M = fmaxf(M, sm_m[w][l16]);
  float f[kSplit], ls = 0.f;                              // the query's own key is always live: M is finite, ls > 0
#pragma unroll
  for (int w = 0; w < kSplit; ++w) { f[w] = __expf(sm_m[w][l16] - M); ls = __builtin_fmaf(sm_l[w][l16], f[w], ls); }
  const float inv = 1.f / ls;
  f32x4 c0, c1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int w = 0; w < kSplit; ++w) {
      a = __builtin_fmaf(sm_o[w][4 * g + i][l16], f[w], a);
      c = __builtin_fmaf(sm_o[w][16 + i][l16], f[w], c);
    }
    c0[i] = a * inv; c1[i] = c * inv;
  }
  if (qb * 16 + l16 < N) {
    *(f32x4*)(p.ctx + row * kL + 4 * g) = c0;
    if (g == 0) {
      *(f32x4*)(p.ctx + row * kL + 16) = c1;
      if (p.lse) p.lse[row] = M + __logf(ls);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ backward, query side
// delta_i = dctx_i . ctx_i ;  dq_i = scale * sum_j p_ij (mask_ij dctx_i . v_j - delta_i) k_j
__device__ __forceinline__ void win_bwd_q_body(const WinArgs& p, const int* tokc, float (*sm_o)[kL][16], const int b, const int qb) {
  const int HW = p.H * p.W, N = p.D * HW;
  const int wave = wave_id(), lane = lane_id(), g = lane >> 4, l16 = lane & 15;
  const float* base = p.qkv + (size_t)b * N * kRow;
  const int q = min(qb * 16 + l16, N - 1);
  const unsigned long long row = (unsigned long long)b * N + q;
  const Box bx = fwd_box(unpack_tok(tokc[q]), p);
  RowOp Q = load_rowop(base + (size_t)q * kRow, g);
  Q.v *= p.scale; Q.x *= p.scale;
  const RowOp Dc = load_rowop(p.dctx + row * kL, g), Cx = load_rowop(p.ctx + row * kL, g);
  const float delta = xsum(Dc.v[0] * Cx.v[0] + Dc.v[1] * Cx.v[1] + Dc.v[2] * Cx.v[2] + Dc.v[3] * Cx.v[3] + Dc.x * Cx.x);
  const float lse = p.lse[row];
  int kb0, kb1;
  block_range((qb * 16) / HW, min(qb * 16 + 15, N - 1) / HW, p.kd / 2, p.kd - 1 - p.kd / 2, p.D, HW, N, kb0, kb1);
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  struct In { RowOp K, V; ColOp Kc; };
  auto fetch = [&](int kb) {
    In r;
    const int k0 = min(kb, kb1 - 1) * 16;
    const float* krow = base + (size_t)min(k0 + l16, N - 1) * kRow;
    r.K = load_rowop(krow + kL, g);
    r.V = load_rowop(krow + 2 * kL, g);
    r.Kc = load_colop(base + kL, kRow, k0, N, g, l16);
    return r;
  };
  In ring[kRing];
#pragma unroll
  for (int r = 0; r < kRing; ++r) ring[r] = fetch(kb0 + wave + r * kSplit);
  auto step = [&](const In& in, int kb) {
    if (kb >= kb1) return;
    const int key0 = kb * 16;
    const f32x4 s = dot_tile(in.K, Q), dp = dot_tile(in.V, Dc);   // [key][query]
    const int4 tk4 = *(const int4*)(tokc + key0 + 4 * g);
    const int tkv[4] = {tk4.x, tk4.y, tk4.z, tk4.w};
    f32x4 ds;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = key0 + 4 * g + i < N && bx.has(unpack_tok(tkv[i]));
      const float pr = ok ? __expf(s[i] - lse) : 0.f;
      const float dpm = p.drop_thresh != 0u ? dp[i] * keep_scale(p, row, N, key0 + 4 * g + i) : dp[i];
      ds[i] = pr * (dpm - delta) * p.scale;
    }
    col_acc(in.Kc, ds, a0, a1);
  };
  for (int kb = kb0 + wave; kb < kb1; kb += kRing * kSplit) {
#pragma unroll
    for (int r = 0; r < kRing; ++r) {
      step(ring[r], kb + r * kSplit);
      ring[r] = fetch(kb + (kRing + r) * kSplit);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sm_o[wave][4 * g + i][l16] = a0[i];
    if (g == 0) sm_o[wave][16 + i][l16] = a1[i];
  }
  __syncthreads();
  if (wave != 0 || qb * 16 + l16 >= N) return;
  f32x4 d0, d1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int w = 0; w < kSplit; ++w) { a += sm_o[w][4 * g + i][l16]; c += sm_o[w][16 + i][l16]; }
    d0[i] = a; d1[i] = c;
  }
  *(f32x4*)(p.dqkv + row * kRow + 4 * g) = d0;
  if (g == 0) {
    *(f32x4*)(p.dqkv + row * kRow + 16) = d1;
    p.delta[row] = delta;
  }
}

// ------------------------------------------------------------------------------------------------------------------ backward, key side
// over the reverse window: dk_j = scale * sum_i p_ij (mask_ij dctx_i . v_j - delta_i) q_i ;  dv_j = sum_i p_ij mask_ij dctx_i
__device__ __forceinline__ void win_bwd_kv_body(const WinArgs& p, const int* tokc, float (*sm_o)[2][kL][16], const int b, const int kb) {
  const int HW = p.H * p.W, N = p.D * HW;
  const int wave = wave_id(), lane = lane_id(), g = lane >> 4, l16 = lane & 15;
  const float* base = p.qkv + (size_t)b * N * kRow;
  const float* dcb = p.dctx + (size_t)b * N * kL;
  const float* cxb = p.ctx + (size_t)b * N * kL;
  const int key = min(kb * 16 + l16, N - 1);
  const Box bx = rev_box(unpack_tok(tokc[key]), p);
  const RowOp K = load_rowop(base + (size_t)key * kRow + kL, g), V = load_rowop(base + (size_t)key * kRow + 2 * kL, g);
  int qb0, qb1;
  block_range((kb * 16) / HW, min(kb * 16 + 15, N - 1) / HW, p.kd - 1 - p.kd / 2, p.kd / 2, p.D, HW, N, qb0, qb1);
  f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0, v0 = k0, v1 = k0;
  // per query block: its rows as row operands (Q, dctx -- and ctx, for delta_i = dctx_i . ctx_i of the row this lane holds) and as column
  // operands; lse of the lane's row.  The four (lse, delta) a lane needs (rows 4g + i) are lane shuffles of those.
  struct In { RowOp Q, Dc, Cx; ColOp Qc, Dcc; float lse; };
  auto fetch = [&](int qb) {
    In r;
    const int q0 = min(qb, qb1 - 1) * 16;
    const int qr = min(q0 + l16, N - 1);
    r.Q = load_rowop(base + (size_t)qr * kRow, g);
    r.Dc = load_rowop(dcb + (size_t)qr * kL, g);
    r.Cx = load_rowop(cxb + (size_t)qr * kL, g);
    r.Qc = load_colop(base, kRow, q0, N, g, l16);
    r.Dcc = load_colop(dcb, kL, q0, N, g, l16);
    r.lse = p.lse[(size_t)b * N + qr];
    return r;
  };
  In ring[kRing];
#pragma unroll
  for (int r = 0; r < kRing; ++r) ring[r] = fetch(qb0 + wave + r * kSplit);
  auto step = [&](const In& in, int qb) {
    if (qb >= qb1) return;
    const int q0 = qb * 16;
    RowOp Q = in.Q;
    Q.v *= p.scale; Q.x *= p.scale;
    const float del_row = xsum(in.Dc.v[0] * in.Cx.v[0] + in.Dc.v[1] * in.Cx.v[1] + in.Dc.v[2] * in.Cx.v[2] + in.Dc.v[3] * in.Cx.v[3] +
                               in.Dc.x * in.Cx.x);          // delta of query q0 + l16, in all four lane groups
    const f32x4 s = dot_tile(Q, K), dp = dot_tile(in.Dc, V);   // [query q0 + 4g + i][key]
    const int4 tq4 = *(const int4*)(tokc + q0 + 4 * g);
    const int tqv[4] = {tq4.x, tq4.y, tq4.z, tq4.w};
    f32x4 ds, pt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qi = q0 + 4 * g + i;
      const bool ok = qi < N && bx.has(unpack_tok(tqv[i]));
      const unsigned long long gi = (unsigned long long)b * N + min(qi, N - 1);
      const float lse_i = __shfl(in.lse, 4 * g + i, 16), del_i = __shfl(del_row, 4 * g + i, 16);   // (width 16: lane 4g + i of this lane's own group)
      const float pr = ok ? __expf(s[i] - lse_i) : 0.f;
      const float msk = p.drop_thresh != 0u ? keep_scale(p, gi, N, key) : 1.f;
      ds[i] = pr * (dp[i] * msk - del_i) * p.scale;
      pt[i] = pr * msk;
    }
    col_acc(in.Qc, ds, k0, k1);
    col_acc(in.Dcc, pt, v0, v1);
  };
  for (int qb = qb0 + wave; qb < qb1; qb += kRing * kSplit) {
#pragma unroll
    for (int r = 0; r < kRing; ++r) {
      step(ring[r], qb + r * kSplit);
      ring[r] = fetch(qb + (kRing + r) * kSplit);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sm_o[wave][0][4 * g + i][l16] = k0[i];
    sm_o[wave][1][4 * g + i][l16] = v0[i];
    if (g == 0) { sm_o[wave][0][16 + i][l16] = k1[i]; sm_o[wave][1][16 + i][l16] = v1[i]; }
  }
  __syncthreads();
  if (wave != 0 || kb * 16 + l16 >= N) return;
  float* out = p.dqkv + ((size_t)b * N + key) * kRow;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f32x4 d0, d1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = 0.f, c = 0.f;
#pragma unroll
      for (int w = 0; w < kSplit; ++w) { a += sm_o[w][t][4 * g + i][l16]; c += sm_o[w][t][16 + i][l16]; }
      d0[i] = a; d1[i] = c;
    }
    *(f32x4*)(out + (1 + t) * kL + 4 * g) = d0;
    if (g == 0) *(f32x4*)(out + (1 + t) * kL + 16) = d1;
  }
}

// kv_side = 0: the query side (dq, delta);  1: the key side (dk, dv) -- one launch each, they share nothing
__global__ __launch_bounds__(64 * kSplit) void win_mfma_bwd_kernel(WinArgs p) {
  extern __shared__ __attribute__((aligned(16))) int tokc[];
  __shared__ float sm_o[kSplit][2][kL][16];
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  const int N = p.D * p.H * p.W, nblk = (N + 15) >> 4;
  fill_tok_table(tokc, N, p.H, p.W, 64 * kSplit);
  const int id = (int)blockIdx.x;
  const int b = id / nblk, blk = id - b * nblk;
  if (p.kv_side == 0) win_bwd_q_body(p, tokc, (float (*)[kL][16])sm_o, b, blk);
  else win_bwd_kv_body(p, tokc, sm_o, b, blk);
}

bool mfma_enabled() {
  static const bool on = [] { const char* e = diag_env("GAVIKO_HIP_WIN_MFMA"); return !(e && e[0] == '0'); }();   // A/B switch
  return on;
}

}  // namespace

int launch_win_mfma_fwd(const WinArgs& a, int L, hipStream_t s) {
  if (L != kL || !mfma_enabled()) return 1;
  const int N = a.D * a.H * a.W;
  if (a.D > 1023 || a.H > 1023 || a.W > 1023 || N > 12288) return 1;          // packed token coordinates, 48 KiB table
  GVK_LAUNCH(win_mfma_fwd_kernel, dim3(a.B * ((N + 15) / 16)), dim3(64 * kSplit), ((N + 15) & ~15) * 4, s, a);
  return check_launch("window_attn_fwd (mfma)");
}

int launch_win_mfma_bwd(const WinArgs& a, int L, hipStream_t s) {
  if (L != kL || !mfma_enabled()) return 1;
  const int N = a.D * a.H * a.W;
  if (a.D > 1023 || a.H > 1023 || a.W > 1023 || N > 12288) return 1;
  WinArgs m = a;                        // one launch per side: the query side (dq, delta), then the key side (dk, dv)
  m.kv_side = 0;
  GVK_LAUNCH(win_mfma_bwd_kernel, dim3(a.B * ((N + 15) / 16)), dim3(64 * kSplit), ((N + 15) & ~15) * 4, s, m);
  m.kv_side = 1;
  GVK_LAUNCH(win_mfma_bwd_kernel, dim3(a.B * ((N + 15) / 16)), dim3(64 * kSplit), ((N + 15) & ~15) * 4, s, m);
  return check_launch("window_attn_bwd (mfma)");
}

}  // namespace gvk
