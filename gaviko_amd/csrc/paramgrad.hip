// Every parameter gradient of one rank-L side-path module of one layer in ONE launch (round 4).
//
// Replaces, per layer and stream, two gvk_outer_reduce calls (partial + final each), one gvk_reduce_batch (partial + final) and, on the
// MWSA stream, gvk_ln_lowrank_affine: 12 launches per layer -> 2.  What the reference computes here is the autograd of
//   gaviko.py:155-156,187  (GPA: proj_down shared by the global and the local tokens, proj_up over every row)
//   gaviko.py:231-232,242  (MWSA: LayerNorm + proj_down, the qkv matrix, proj_up behind proj_drop)
// i.e. outer products  D[l][c] = sum_m narrow'[m][l] * wide'[m][c]  over M = 4..8 k token rows, plus a handful of tiny column sums /
// (J x L) products over the same rows.
//
// Structure.  The grid is a job list: [outer jobs | small jobs], sized to about one workgroup per CU (beside the backbone's GEMM
// workgroups a CU has registers for one more 4-wave workgroup at most, so a grid of several rounds only queues: the first form of this
// kernel, 700-1000 workgroups of 64 rows per wave, ran 46-72 us per launch).  An outer-product workgroup (4 waves) owns one 64-column tile
// and a contiguous range of rows, a quarter per wave; a wave streams its rows in chunks of 32 through two register buffers (the next
// chunk's loads are always in flight behind the current chunk's MFMAs; every load is unconditional, out-of-range rows are clamped and
// zeroed afterwards, so the compiler can count vmcnt exactly), multiplies on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32
// products) and keeps ONE accumulator tile for all its rows.  The four waves' tiles are summed through LDS and the workgroup writes ONE
// partial tile [(L+1)][64] (row L = column sum of wide').  The partial tiles of a column tile (7-14 of them) are then summed by the
// workgroup that arrives LAST at the tile's ticket counter -- in row-range order, so the result does not depend on who that is:
// deterministic, no second launch, no 5-us dependent hop.  The hand-off between workgroups is recipe R1 of cdna_hip_programming.md
// Guideline 16: the partial tiles are stored WRITE-THROUGH (sc1: they leave the XCD's L2 at once, so no release fence -- an agent-scope
// release is a write-back of every dirty line of that L2, and beside the backbone's GEMMs, whose fp32 outputs sit dirty in it, hundreds
// of workgroups doing that cost the step 12 %), every storing wave drains its stores, workgroup barrier, ONE lane takes the ticket
// (agent-scope atomic); the last arriver acquires at agent scope (drops its CU's stale L1 lines) before any wave of it loads a partial.
// The ticket words are zero at allocation and the last arriver writes its word back to zero, so a replayed plan needs no memset
// (launches that share ticket words are ordered by their stream).
#include "common.hpp"
#include "dropout.hpp"
#include "../../include/gaviko_hip.h"
#include <algorithm>

namespace gvk {

#ifndef GVK_PG_CH
#define GVK_PG_CH 16
#endif
constexpr int kPgCh = GVK_PG_CH;     // rows per chunk per wave (a multiple of 4; 32 and 16 are measured)
constexpr int kPgKS = kPgCh / 4;     // MFMA k-steps (four rows each) per chunk
constexpr int kPgMaxOuter = 3;
constexpr int kPgMaxSmall = 8;
constexpr int kPgMaxSg = 64;         // row ranges (workgroups) per column tile of an outer job
constexpr int kPgRedSlabs = 32;      // row slabs of a small job
#ifndef GVK_PG_WGS
#define GVK_PG_WGS 240
#endif
constexpr int kPgTargetWgs = GVK_PG_WGS;    // outer-product workgroups per launch: about one per CU
constexpr int kPgMinRows = 128;      // ... but never fewer rows per workgroup than this

struct PgOuter {
  const float* narrow; const float* wide; const float* narrow2; const float* wide2;   // rows M1.. come from (narrow2, wide2)
  const float* lat_override;                                                        // narrow rows t < P of a sample come from here [B*P][L]
  const float* mean; const float* rstd;                                             // wide' = (wide - mean[m]) * rstd[m]
  float* out; float* colsum;
  const float* aff_w; const float* aff_g; const float* aff_b; float* aff_dgamma; float* aff_dbeta; float* aff_dbias;
  unsigned long long seed; unsigned int drop_thresh; float inv_keep;                // wide' *= dropout mask of element (m, c)
  int M, M1, T, P, transposed, accumulate;
  int C, nct;                                                                       // columns (= row stride) of wide, 64-column tiles
  int wg0, nsg, tick0; long scr0;                                                    // first workgroup / row ranges / first ticket / scratch offset (floats)
};
struct PgSmall {
  const float* a; const float* b; const float* a2; float* out;                       // as gvk_reduce_job
  int M, M1, J, L, accumulate, wg0, nslab, tick0; long scr0;
};
struct PgArgs {
  PgOuter o[kPgMaxOuter];
  PgSmall s[kPgMaxSmall];
  float* scratch; int* tickets; const unsigned long long* seed_ptr;
  int nouter, nsmall, small_wg0, L;
};

// write-through (sc1) stores of a partial: 16 bytes through a buffer resource over the workgroup's slab, 4 bytes as an agent-scope atomic store
__device__ __forceinline__ void pg_store16_wt(__amdgpu_buffer_rsrc_t rsrc, int byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(*(u32x4*)&v, rsrc, byte_off, 0, 16);     // aux 16 = sc1
}
__device__ __forceinline__ void pg_store4_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The workgroup's partial stores (all sc1) are drained and ONE lane takes the ticket; returns (to every thread) whether this workgroup
// arrived last.
__device__ __forceinline__ bool pg_arrive_last(int* ticket, int n, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its write-through stores ...
  __syncthreads();                                           // ... before the one lane that signals for all of them
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == n - 1) ? 1 : 0;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // drop this CU's stale lines before any wave of it reads a partial
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this stream
    }
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}

template <int NT>
__global__ __launch_bounds__(256) void param_grads_kernel(PgArgs p) {
  constexpr int NW = NT * 16;                                // padded narrow width (L + 1 <= NW)
  constexpr int kNP = (kPgCh * 7 + 63) / 64;                 // narrow float4 pieces per lane and chunk (kPgCh rows x L / 4 <= 7 pieces)
  // [wave][buffer][row][NW] narrow rows, later the four waves' accumulator tiles [wave][16 NT registers][64 lanes] (the larger of the two)
  constexpr int kLdsFloats = (4 * 2 * kPgCh * 32 > 4 * 16 * NT * 64) ? 4 * 2 * kPgCh * 32 : 4 * 16 * NT * 64;
  __shared__ float lds[kLdsFloats];
  __shared__ float st[4][2][kPgCh][2];
  __shared__ float sw[4][32];
  __shared__ int s_flag;
  const int L = p.L;
  const int lane = lane_id(), wave = wave_id();
  const int bid = blockIdx.x;

  if (bid >= p.small_wg0) {
    // ------------------------------------------------------------------ small jobs: column sums / (J x L) products over the rows
    int ji = 0;
#pragma unroll
    for (int k = 1; k < kPgMaxSmall; ++k)
      if (k < p.nsmall && bid >= p.s[k].wg0) ji = k;
    const PgSmall jb = p.s[ji];
    const int local = bid - jb.wg0;
    const int slab = local % jb.nslab, chunk = local / jb.nslab;
    const int nout = jb.b ? jb.J * jb.L : jb.J;
    const int o0 = chunk * 64;
    const int no = min(64, nout - o0);
    const int rows_per = (jb.M + jb.nslab - 1) / jb.nslab;
    const int r0 = slab * rows_per, r1 = min(jb.M, r0 + rows_per);
    const int oi = threadIdx.x & 63, sl = threadIdx.x >> 6;     // 4 row slices
    float acc = 0.f;
    if (oi < no) {
      const int o = o0 + oi;
      if (jb.b != nullptr) {
        const int j = o / jb.L, l = o - j * jb.L;
        for (int m = r0 + sl; m < r1; m += 32) {                // eight rows' loads in flight per thread and pass
          float av[8], bv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int mm = m + u * 4;
            av[u] = mm < r1 ? jb.a[(size_t)mm * jb.J + j] : 0.f;
            bv[u] = mm < r1 ? jb.b[(size_t)mm * jb.L + l] : 0.f;
          }
          acc += ((av[0] * bv[0] + av[1] * bv[1]) + (av[2] * bv[2] + av[3] * bv[3])) + ((av[4] * bv[4] + av[5] * bv[5]) + (av[6] * bv[6] + av[7] * bv[7]));
        }
      } else {
        for (int m = r0 + sl; m < r1; m += 32) {
          float av[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int mm = m + u * 4;
            av[u] = mm < r1 ? (mm < jb.M1 ? jb.a[(size_t)mm * jb.J + o] : jb.a2[(size_t)(mm - jb.M1) * jb.J + o]) : 0.f;
          }
          acc += ((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7]));
        }
      }
    }
    lds[threadIdx.x] = acc;
    __syncthreads();
    float* part = p.scratch + jb.scr0 + (size_t)chunk * jb.nslab * 64;
    if (sl == 0 && oi < no) pg_store4_wt(part + slab * 64 + oi, (lds[oi] + lds[64 + oi]) + (lds[128 + oi] + lds[192 + oi]));
    if (!pg_arrive_last(p.tickets + jb.tick0 + chunk, jb.nslab, &s_flag)) return;
    if (threadIdx.x < no) {
      float a4[4] = {0.f, 0.f, 0.f, 0.f}, pv[kPgRedSlabs];
#pragma unroll
      for (int s = 0; s < kPgRedSlabs; ++s) pv[s] = part[min(s, jb.nslab - 1) * 64 + threadIdx.x];      // every slab's partial requested at once
#pragma unroll
      for (int s = 0; s < kPgRedSlabs; ++s) a4[s & 3] += (s < jb.nslab) ? pv[s] : 0.f;
      const float t = (a4[0] + a4[1]) + (a4[2] + a4[3]);
      float* dst = jb.out + o0 + threadIdx.x;
      *dst = jb.accumulate ? *dst + t : t;
    }
    return;
  }

  // -------------------------------------------------------------------- outer products
  int k = 0;
#pragma unroll
  for (int q = 1; q < kPgMaxOuter; ++q)
    if (q < p.nouter && bid >= p.o[q].wg0) k = q;
  const PgOuter J = p.o[k];                                   // BY VALUE: scalar loads into SGPRs (through a reference hipcc indexed the kernarg block per lane: a dependent vector load + vmcnt(0) in front of every data load)
  const int C = J.C, nct = J.nct;
  const int local = bid - J.wg0;
  const int ct = local % nct, sg = local / nct;
  const int rows_wg = (J.M + J.nsg - 1) / J.nsg;
  const int rw = (rows_wg + 3) >> 2;                           // rows per wave
  const int wg_lo = min(J.M, sg * rows_wg), wg_hi = min(J.M, wg_lo + rows_wg);
  const int r0 = min(wg_hi, wg_lo + wave * rw), r1 = min(wg_hi, r0 + rw);
  const int nr = r1 - r0;
  const int j = lane & 15, kq = lane >> 4;
  const int c = ct * 64 + 4 * j;
  const bool cok = c < C;
  const int cl = cok ? c : 0;                                  // clamped column of this lane's loads
  const int L4 = L >> 2;                                       // (the host admits L % 4 == 0 only)
  unsigned long long seed = J.seed;
  if (J.drop_thresh != 0u && p.seed_ptr != nullptr) seed += *p.seed_ptr;
  float (*nar)[kPgCh][NW] = (float (*)[kPgCh][NW])(lds + wave * 2 * kPgCh * 32);     // [buffer][row][NW]
  const int mlast = max(J.M - 1, 0);

  auto narrow_row = [&](int m) -> const float* {
    const float* src = (J.narrow2 != nullptr && m >= J.M1) ? J.narrow2 + (size_t)(m - J.M1) * L : J.narrow + (size_t)m * L;
    if (J.lat_override != nullptr) {
      const int s = m / J.T, t = m - s * J.T;
      if (t < J.P) src = J.lat_override + ((size_t)s * J.P + t) * L;
    }
    return src;
  };
  auto wide_row = [&](int m) -> const float* {
    return (J.narrow2 != nullptr && m >= J.M1) ? J.wide2 + (size_t)(m - J.M1) * C : J.wide + (size_t)m * C;
  };
  // one chunk's loads: 8 float4 of wide, up to 4 float4 of narrow, the row statistics.  UNCONDITIONAL (rows / columns clamped): the
  // waits in front of a chunk's consumers are then exact counts, and the chunk behind stays in flight
  auto load = [&](f32x4 (&x)[kPgKS], f32x4 (&nv)[kNP], float& mu, float& rs, int ch) {
    const int rb = r0 + kPgCh * ch;
#pragma unroll
    for (int t = 0; t < kNP; ++t) {
      const int e = lane + 64 * t;
      const int r = e / L4, pc = e - r * L4;
      nv[t] = *(const f32x4*)(narrow_row(min(rb + r, mlast)) + 4 * min(pc, L4 - 1));
    }
#pragma unroll
    for (int u = 0; u < kPgKS; ++u) x[u] = *(const f32x4*)(wide_row(min(rb + 4 * u + kq, mlast)) + cl);
    if (J.mean != nullptr) {
      const int m = min(rb + (lane & (kPgCh - 1)), mlast);
      mu = J.mean[m];
      rs = J.rstd[m];
    }
  };
  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s4[4] = {0.f, 0.f, 0.f, 0.f};                          // affine jobs: this lane's column (l = lane) of S = sum_m narrow[m][l]
  auto compute = [&](f32x4 (&x)[kPgKS], f32x4 (&nv)[kNP], float mu, float rs, const int buf, int ch) {
    const int rb = r0 + kPgCh * ch;
    if (rb >= r1) return;                                      // wave-uniform: a chunk wholly past this wave's rows (its loads were issued anyway)
#pragma unroll
    for (int t = 0; t < kNP; ++t) {
      const int e = lane + 64 * t;
      const int r = e / L4, pc = e - r * L4;
      if (r < kPgCh) *(f32x4*)(&nar[buf][r][4 * pc]) = (rb + r < r1) ? nv[t] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (J.mean != nullptr && lane < kPgCh) { st[wave][buf][lane][0] = mu; st[wave][buf][lane][1] = rs; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (wave-private LDS region: program order is enough, the compiler must keep it)
    __builtin_amdgcn_wave_barrier();
    if (J.aff_w != nullptr && lane < NW) {
#pragma unroll
      for (int r = 0; r < kPgCh; ++r) s4[r & 3] += nar[buf][r][lane];
    }
#pragma unroll
    for (int u = 0; u < kPgKS; ++u) {
      if (rb + 4 * u >= r1) break;                             // wave-uniform: k-steps wholly past the rows
      const int r = 4 * u + kq;
      const bool ok = (rb + r < r1) && cok;
      f32x4 xv = x[u];
      if (J.drop_thresh != 0u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] *= drop_scale(seed, (unsigned long long)(rb + r) * C + c + e, J.drop_thresh, J.inv_keep);
      }
      if (J.mean != nullptr) {
        const float m_ = st[wave][buf][r][0], r_ = st[wave][buf][r][1];
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] = (xv[e] - m_) * r_;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) xv[e] = ok ? xv[e] : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float a = nar[buf][r][t * 16 + j];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[e], acc[t][e], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  // columns L.. of every narrow row never change: [1, 0, 0, ...] (row L of the result is the column sum of wide')
  for (int i = lane; i < 2 * kPgCh * (NW - L); i += 64) {
    const int r = i / (NW - L), l = L + (i - r * (NW - L));
    nar[0][r][l] = (l == L) ? 1.f : 0.f;                       // (rows 32.. of "buffer 0" are buffer 1)
  }
  {
    const int nch = (nr + kPgCh - 1) / kPgCh;
    // four register buffers: three chunks' loads (12 KiB of wide rows per wave) stay in flight behind the chunk being multiplied -- the
    // launch is a few chunks long per wave, so what it costs is round trips, not bytes (two buffers: 1.2 us per chunk)
    f32x4 x0[kPgKS], x1[kPgKS], x2[kPgKS], x3[kPgKS], n0[kNP], n1[kNP], n2[kNP], n3[kNP];
    float mu0 = 0.f, rs0 = 0.f, mu1 = 0.f, rs1 = 0.f, mu2 = 0.f, rs2 = 0.f, mu3 = 0.f, rs3 = 0.f;
    load(x0, n0, mu0, rs0, 0);
    load(x1, n1, mu1, rs1, 1);
    load(x2, n2, mu2, rs2, 2);
    for (int ch = 0; ch < nch; ch += 4) {
      load(x3, n3, mu3, rs3, ch + 3);
      compute(x0, n0, mu0, rs0, 0, ch);
      load(x0, n0, mu0, rs0, ch + 4);
      compute(x1, n1, mu1, rs1, 1, ch + 1);
      load(x1, n1, mu1, rs1, ch + 5);
      compute(x2, n2, mu2, rs2, 0, ch + 2);
      load(x2, n2, mu2, rs2, ch + 6);
      compute(x3, n3, mu3, rs3, 1, ch + 3);
    }
  }
  if (J.aff_w != nullptr && lane < NW) sw[wave][lane] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  // the four waves' tiles -> one: every wave leaves its 16 NT accumulator registers in LDS, wave w then sums its share in wave order
  __syncthreads();                                             // (every wave is done with its narrow rows)
  float (*red)[16 * NT][64] = (float (*)[16 * NT][64])lds;     // [wave][register][lane]
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[wave][(t * 4 + e) * 4 + q][lane] = acc[t][e][q];
  __syncthreads();
  const int stride = (L + 2) * 64;
  float* slab = p.scratch + J.scr0 + ((size_t)ct * J.nsg + sg) * stride;
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, stride * 4, 0x00020000);
  // the (t, q) pairs are dealt to the waves; a lane then holds the four e of one (l, 4 columns)
  for (int tq = wave; tq < NT * 4; tq += 4) {
    const int t = tq >> 2, q = tq & 3;
    const int l = t * 16 + 4 * kq + q;                         // D row = 4 * (lane >> 4) + register
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rg = (t * 4 + e) * 4 + q;
      o[e] = (red[0][rg][lane] + red[1][rg][lane]) + (red[2][rg][lane] + red[3][rg][lane]);
    }
    if (l <= L) pg_store16_wt(srs, (l * 64 + 4 * j) * 4, o);
  }
  if (J.aff_w != nullptr && wave == 0 && lane < NW)
    pg_store4_wt(slab + (L + 1) * 64 + lane, (sw[0][lane] + sw[1][lane]) + (sw[2][lane] + sw[3][lane]));

  if (!pg_arrive_last(p.tickets + J.tick0 + ct, J.nsg, &s_flag)) return;

  // ---- last arriver of this column tile: sum the row ranges' tiles in range order, then the epilogue
  const float* tile0 = p.scratch + J.scr0 + (size_t)ct * J.nsg * stride;
  static_assert(kLdsFloats >= 30 * 64, "the affine epilogue stages [L + 2][64] floats");
  float (*qt)[64] = (float (*)[64])lds;                       // affine epilogue: the summed tile [L + 1][64] and S behind it
  const int ngran = (L + 1) * 16 + ((J.aff_w != nullptr) ? NW / 4 : 0);     // float4 granules: the tile (+ the S row)
  // The partials come from memory (they were stored write-through): up to 16 row ranges x both granules of a thread are requested before the
  // first is consumed (clamped, unconditional loads; one round trip instead of nsg / 4 dependent ones).  Same association as a plain
  // four-way interleaved sum over the ranges: a4[s & 3] += partial[s], in range order.
  f32x4 vsum[2];
  {
    f32x4 a4[2][4];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int u = 0; u < 4; ++u) a4[gi][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < J.nsg; s0 += 16) {
      f32x4 pb[2][16];
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        const int g = min((int)threadIdx.x + 256 * gi, ngran - 1);
#pragma unroll
        for (int u = 0; u < 16; ++u) pb[gi][u] = *(const f32x4*)(tile0 + (size_t)min(s0 + u, J.nsg - 1) * stride + 4 * g);
      }
#pragma unroll
      for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (s0 + u < J.nsg) a4[gi][u & 3] += pb[gi][u];
    }
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) vsum[gi] = (a4[gi][0] + a4[gi][1]) + (a4[gi][2] + a4[gi][3]);
  }
  static_assert((28 + 1) * 16 + 8 <= 512, "two granules per thread cover the tile");
#pragma unroll
  for (int gi = 0; gi < 2; ++gi) {
    const int g = threadIdx.x + 256 * gi;
    if (g >= ngran) continue;
    const f32x4 v = vsum[gi];
    const int l = g >> 4, cc = ct * 64 + 4 * (g & 15);
    if (J.aff_w != nullptr) {
      *(f32x4*)(&qt[0][0] + 4 * g) = v;                        // rows 0..L-1 = Q, row L = column sum (unused), row L+1.. = S
      continue;
    }
    if (cc >= C) continue;
    if (l < L) {
      if (J.out != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float* o = J.transposed ? J.out + (size_t)(cc + e) * L + l : J.out + (size_t)l * C + cc + e;
          *o = J.accumulate ? *o + v[e] : v[e];
        }
      }
    } else if (l == L && J.colsum != nullptr) {
      f32x4* o = (f32x4*)(J.colsum + cc);
      *o = J.accumulate ? *o + v : v;
    }
  }
  if (J.aff_w == nullptr) return;
  // Affine / weight gradients of  y = LN(x) . Wd^T  from Q[l][c] = sum_m dlat[m][l] xhat[m][c] and S[l] = sum_m dlat[m][l]:
  //   dWd[l][c] = g_c Q[l][c] + b_c S[l],  dgamma_c = sum_l Wd[l][c] Q[l][c],  dbeta_c = sum_l Wd[l][c] S[l],  dbias_l = S[l]
  __syncthreads();
  const float* S = &qt[L + 1][0];
  if (threadIdx.x < 64 && ct * 64 + (int)threadIdx.x < C) {
    const int cc = ct * 64 + threadIdx.x;
    const float gc = J.aff_g[cc], bc = J.aff_b[cc];
    float dg = 0.f, db = 0.f;
    for (int l = 0; l < L; ++l) {
      const float q = qt[l][threadIdx.x], sl = S[l], w = J.aff_w[(size_t)l * C + cc];
      float* o = J.out + (size_t)l * C + cc;
      const float d = gc * q + bc * sl;
      *o = J.accumulate ? *o + d : d;
      dg += w * q;
      db += w * sl;
    }
    J.aff_dgamma[cc] = J.accumulate ? J.aff_dgamma[cc] + dg : dg;
    J.aff_dbeta[cc] = J.accumulate ? J.aff_dbeta[cc] + db : db;
  }
  if (ct == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + L && J.aff_dbias != nullptr) {
    const int l = threadIdx.x - 64;
    J.aff_dbias[l] = J.accumulate ? J.aff_dbias[l] + S[l] : S[l];
  }
}

static unsigned int pg_drop_threshold(float p) { return drop_threshold_u32(p); }

// rows per outer-product workgroup: the jobs' (rows x column tiles) dealt to about kPgTargetWgs workgroups
static int pg_rows_target(const gvk_pgrad_outer* outer, int n_outer, int C) {
  long total = 0;
  for (int k = 0; k < n_outer; ++k) {
    const int Cj = outer[k].C > 0 ? outer[k].C : C;
    total += (long)(outer[k].M + (outer[k].narrow2 ? outer[k].M2 : 0)) * ((Cj + 63) / 64);
  }
  return (int)std::max<long>(kPgMinRows, (total + kPgTargetWgs - 1) / kPgTargetWgs);
}
static int pg_nsg(int rows, int target) { return std::min(kPgMaxSg, std::max(1, (rows + target - 1) / target)); }

}  // namespace gvk

extern "C" int64_t gvk_param_grads_scratch(const gvk_pgrad_outer* outer, int n_outer, const gvk_reduce_job* small, int n_small, int C, int L) {
  using namespace gvk;
  int64_t n = 0;
  const int target = n_outer > 0 ? pg_rows_target(outer, n_outer, C) : kPgMinRows;
  for (int k = 0; k < n_outer; ++k) {
    const int M = outer[k].M + (outer[k].narrow2 ? outer[k].M2 : 0);
    const int Cj = outer[k].C > 0 ? outer[k].C : C;
    n += (int64_t)((Cj + 63) / 64) * pg_nsg(M, target) * (L + 2) * 64;
  }
  for (int k = 0; k < n_small; ++k) {
    const int nout = small[k].b ? small[k].J * small[k].L : small[k].J;
    n += (int64_t)((nout + 63) / 64) * kPgRedSlabs * 64;
  }
  return n;
}

extern "C" int gvk_param_grads(const gvk_pgrad_outer* outer, int n_outer, const gvk_reduce_job* small, int n_small, float* scratch,
                               int64_t scratch_elems, int32_t* tickets, int n_tickets, const void* seed_ptr, int C, int L, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(scratch && tickets && n_outer >= 0 && n_outer <= kPgMaxOuter && n_small >= 0 && n_small <= kPgMaxSmall && n_outer + n_small > 0,
              "gvk_param_grads: up to %d outer and %d small jobs, scratch and tickets required", kPgMaxOuter, kPgMaxSmall);
  GVK_REQUIRE(C > 0 && C % 4 == 0 && L > 0 && L <= 28 && L % 4 == 0, "gvk_param_grads: C=%d must be a multiple of 4, L=%d a multiple of 4 up to 28", C, L);
  GVK_REQUIRE(n_outer == 0 || outer != nullptr, "gvk_param_grads: null outer job list");
  GVK_REQUIRE(n_small == 0 || small != nullptr, "gvk_param_grads: null small job list");
  PgArgs a{};
  a.scratch = scratch; a.tickets = tickets; a.seed_ptr = (const unsigned long long*)seed_ptr; a.L = L;
  a.nouter = n_outer; a.nsmall = n_small;
  int wg = 0, tick = 0;
  long scr = 0;
  const int target = n_outer > 0 ? pg_rows_target(outer, n_outer, C) : kPgMinRows;
  for (int k = 0; k < n_outer; ++k) {
    const gvk_pgrad_outer& d = outer[k];
    GVK_REQUIRE(d.narrow && d.wide && d.M > 0 && (d.out || d.colsum), "gvk_param_grads: outer job %d: null pointer / empty", k);
    GVK_REQUIRE(d.C >= 0 && d.C % 4 == 0, "gvk_param_grads: outer job %d: C=%d must be a multiple of 4 (0 = the call's C)", k, d.C);
    GVK_REQUIRE((d.mean == nullptr) == (d.rstd == nullptr), "gvk_param_grads: outer job %d: mean / rstd must come together", k);
    GVK_REQUIRE(d.narrow2 == nullptr || (d.wide2 != nullptr && d.M2 > 0 && d.mean == nullptr && d.lat_override == nullptr && d.drop_p <= 0.f),
                "gvk_param_grads: outer job %d: the second source takes plain rows only", k);
    GVK_REQUIRE(d.lat_override == nullptr || (d.T > 0 && d.P > 0 && d.P <= d.T), "gvk_param_grads: outer job %d: override needs 0 < P <= T", k);
    GVK_REQUIRE(d.drop_p >= 0.f && d.drop_p < 1.f && (d.drop_p == 0.f || seed_ptr != nullptr), "gvk_param_grads: outer job %d: drop_p in [0,1) and a seed word", k);
    GVK_REQUIRE(d.aff_w == nullptr || (d.aff_gamma && d.aff_beta && d.aff_dgamma && d.aff_dbeta && d.out && !d.transposed && d.colsum == nullptr),
                "gvk_param_grads: outer job %d: the affine epilogue needs gamma / beta / dgamma / dbeta, out [L][C] and takes no colsum", k);
    PgOuter& o = a.o[k];
    o.narrow = d.narrow; o.wide = d.wide; o.narrow2 = d.narrow2; o.wide2 = d.wide2; o.lat_override = d.lat_override;
    o.mean = d.mean; o.rstd = d.rstd; o.out = d.out; o.colsum = d.colsum;
    o.aff_w = d.aff_w; o.aff_g = d.aff_gamma; o.aff_b = d.aff_beta; o.aff_dgamma = d.aff_dgamma; o.aff_dbeta = d.aff_dbeta; o.aff_dbias = d.aff_dbias;
    o.seed = d.seed; o.drop_thresh = pg_drop_threshold(d.drop_p); o.inv_keep = d.drop_p > 0.f ? 1.f / (1.f - d.drop_p) : 1.f;
    o.M1 = d.M; o.M = d.M + (d.narrow2 ? d.M2 : 0); o.T = d.T; o.P = d.P; o.transposed = d.transposed; o.accumulate = d.accumulate;
    o.C = d.C > 0 ? d.C : C; o.nct = (o.C + 63) / 64;
    o.nsg = pg_nsg(o.M, target);
    o.wg0 = wg; o.tick0 = tick; o.scr0 = scr;
    wg += o.nct * o.nsg; tick += o.nct; scr += (long)o.nct * o.nsg * (L + 2) * 64;
  }
  a.small_wg0 = wg;
  for (int k = 0; k < n_small; ++k) {
    const gvk_reduce_job& j = small[k];
    GVK_REQUIRE(j.a && j.out && j.M > 0 && j.J > 0 && (j.b == nullptr || j.L > 0), "gvk_param_grads: small job %d malformed", k);
    GVK_REQUIRE(j.a2 == nullptr || (j.b == nullptr && j.M2 > 0), "gvk_param_grads: small job %d: a second source (a2, M2) goes with a column sum only", k);
    const int nout = j.b ? j.J * j.L : j.J;
    const int rows = j.M + (j.a2 ? j.M2 : 0);
    PgSmall& s = a.s[k];
    s.a = j.a; s.b = j.b; s.a2 = j.a2; s.out = j.out; s.M = rows; s.M1 = j.a2 ? j.M : 0x7fffffff; s.J = j.J; s.L = j.L; s.accumulate = j.accumulate;
    s.nslab = std::min(kPgRedSlabs, std::max(1, (rows + 255) / 256));
    const int nchunk = (nout + 63) / 64;
    s.wg0 = wg; s.tick0 = tick; s.scr0 = scr;
    wg += nchunk * s.nslab; tick += nchunk; scr += (long)nchunk * kPgRedSlabs * 64;
  }
  GVK_REQUIRE(scr <= scratch_elems, "gvk_param_grads: scratch holds %lld floats, %ld needed", (long long)scratch_elems, scr);
  GVK_REQUIRE(tick <= n_tickets, "gvk_param_grads: %d ticket words given, %d needed", n_tickets, tick);
  hipStream_t st = (hipStream_t)stream;
  if (L + 1 <= 16) GVK_LAUNCH(param_grads_kernel<1>, dim3(wg), dim3(256), 0, st, a);
  else GVK_LAUNCH(param_grads_kernel<2>, dim3(wg), dim3(256), 0, st, a);
  return check_launch("param_grads");
}
