// "Skinny" fp32 kernels for GAViKO's trainable rank-L side paths (L = latent dim, 20 by default):
//   down   y[m][0:L]   = act( LN?(x[m][:]) . W^T + b )          (+ optional second tiny matrix: y2 = y . W2^T)
//   up     out[m][:]   = res + drop( lat[m][0:L] . W^T + b )      (or accumulate into out)
//   outer  dW[l][c]    = sum_m narrow[m][l] * wide'[m][c]          (weight gradients of down / up projections)
//   small  dW[j][l]    = sum_m a[m][j] * b[m][l]                   (L x L sized weight gradients)
//   colsum db[c]       = sum_m x[m][c]
// All of them are HBM-bound streams over a [M][C] fp32 token matrix (C = 768/1024) with ~2*L flop per byte;
// fp32 VALU throughout because these feed trainable parameters.  Reductions are two-stage and deterministic.
// Replaces: gaviko.py:231 (norm + proj_down), :232 (qkv), :242 (proj_up), :155-156 (GPA proj_down + QuickGELU),
//           :187 (GPA proj_up) and the autograd wgrad/dgrad of each.
#include "common.hpp"
#include <algorithm>
#include <cstdlib>
#include "skinny_args.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

constexpr int kDownRows = 16;      // one 16-row MFMA tile per workgroup; the 16 waves split the C (reduction) dimension

// y[16 rows][L] = X[16][C] . W^T on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products and accumulation).
// Wave w owns the 16-column "super-steps" s = w, w+16, ... of the reduction: a lane (r = lane&15, kq = lane>>4) loads
// x[row0+r][16s + 4kq .. +3] as one float4 and uses element e as the A operand of MFMA e (k = kq), against
// W[j = lane&15][16s + 4kq + e] read as one float4 from the LDS copy of W.  Partial 16x32 tiles of the 16 waves are summed
// through LDS.  A fused LayerNorm is applied algebraically: y = rstd * (x.(g*W) - mean * sum(g*W)) + (beta.W + bias).
template <int L>
__global__ __launch_bounds__(1024) void skinny_down_kernel(DownArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = (L + 15) / 16;            // 16-column output tiles
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  const int C = p.C, Cp = C + 4;               // padded LDS row (keeps the b128 operand reads nearly conflict-free)
  float* cst = (float*)smem;                   // [2][32]  s1_j = sum_c g_c W_jc, s2_j = sum_c b_c W_jc + bias_j
  float* stat = cst + 64;                      // [16 waves][16 rows][2] partial sum / sum of squares
  float* ws = stat + 16 * 16 * 2;              // [L][Cp]  (gamma-folded when LN is fused); reused for the partial tiles
  const int lane = lane_id(), wave = wave_id();
  const int r = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * kDownRows, row = row0 + r;
  const bool live = row < p.M;
  const bool ln = p.ln_g != nullptr;
  const int nsteps = (C + 15) / 16;

  // ---- issue this wave's x loads first (latency hides behind the weight staging)
  constexpr int kMaxSteps = 4;                 // C <= 1024 -> at most 4 super-steps per wave
  f32x4 xv[kMaxSteps];
#pragma unroll
  for (int u = 0; u < kMaxSteps; ++u) {
    const int c = 16 * (wave + 16 * u) + 4 * kq;
    xv[u] = (live && c < C) ? *(const f32x4*)(p.x + (size_t)row * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // ---- stage W (optionally gamma-folded) into LDS
  for (int i = threadIdx.x; i < L * C; i += 1024) {
    int j, c;
    float v;
    if (p.w_layout == 0) { j = i / C; c = i - j * C; v = p.w[i]; }
    else { c = i / L; j = i - c * L; v = p.w[i]; }
    if (ln) v *= p.ln_g[c];
    ws[j * Cp + c] = v;
  }
  if (threadIdx.x < 64) cst[threadIdx.x] = 0.f;
  __syncthreads();
  // per-output constants (one wave per j): s1_j over the folded LDS copy, s2_j = beta.W_j + bias_j
  for (int j = wave; j < L; j += 16) {
    float a1 = 0.f, a2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      a1 += ws[j * Cp + c];
      if (ln) a2 += p.ln_b[c] * (p.w_layout == 0 ? p.w[(size_t)j * C + c] : p.w[(size_t)c * L + j]);
    }
    a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) { cst[j] = a1; cst[32 + j] = a2 + (p.bias ? p.bias[j] : 0.f); }
  }
  // ---- MFMA over this wave's super-steps, plus the row statistics of its column slice
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < kMaxSteps; ++u) {
    const int st = wave + 16 * u;
    if (st < nsteps) {
      const int c = 16 * st + 4 * kq;
      f32x4 x4 = xv[u];
      if (p.drop_thresh != 0u && live && c < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x4[e] *= drop_scale(p.seed, (unsigned long long)row * C + c + e, p.drop_thresh, p.inv_keep);
      }
      s1 += x4[0] + x4[1] + x4[2] + x4[3];
      s2 += x4[0] * x4[0] + x4[1] * x4[1] + x4[2] * x4[2] + x4[3] * x4[3];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int j = min(t * 16 + r, L - 1);                       // columns >= L re-read row L-1; their results are discarded
        const f32x4 w4 = (c < C) ? *(const f32x4*)(ws + j * Cp + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[e], w4[e], acc[t], 0, 0, 0);
      }
    }
  }
  if (ln) {
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    if (kq == 0) { stat[(wave * 16 + r) * 2] = s1; stat[(wave * 16 + r) * 2 + 1] = s2; }
  }
  __syncthreads();                              // all waves are done reading ws: reuse it for the partial tiles
  float* part = ws;                             // [16 waves][16 rows][32]
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[(wave * 16 + (4 * kq + e)) * 32 + t * 16 + r] = acc[t][e];   // D: row = 4*(lane>>4)+e, col = lane&15
  __syncthreads();
  float* yrow = part + 16 * 16 * 32;            // [16][32] finished activations (for the second stage)
  if (threadIdx.x < 16 * 32) {
    const int i = threadIdx.x >> 5, j = threadIdx.x & 31;
    float dot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) dot += part[(w * 16 + i) * 32 + j];
    float yv = 0.f;
    if (j < L && row0 + i < p.M) {
      float zz;
      if (ln) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) { a += stat[(w * 16 + i) * 2]; q += stat[(w * 16 + i) * 2 + 1]; }
        const float mean = a / (float)C;
        const float var = fmaxf(q / (float)C - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        if (j == 0) {
          if (p.mean) p.mean[row0 + i] = mean;
          if (p.rstd) p.rstd[row0 + i] = rstd;
        }
        zz = rstd * (dot - mean * cst[j]) + cst[32 + j];
      } else {
        zz = dot + cst[32 + j];
      }
      yv = p.act == 1 ? quick_gelu(zz) : zz;
      if (p.z) p.z[(size_t)(row0 + i) * L + j] = zz;
      if (p.y) p.y[(size_t)(row0 + i) * L + j] = yv;
    }
    yrow[i * 32 + j] = yv;
  }
  if (p.w2 != nullptr) {
    __syncthreads();
    for (int o = threadIdx.x; o < 16 * p.L2; o += 1024) {
      const int i = o / p.L2, j2 = o - i * p.L2;
      if (row0 + i < p.M) {
        float a = 0.f;
#pragma unroll
        for (int l = 0; l < L; ++l) a += yrow[i * 32 + l] * p.w2[j2 * L + l];
        p.y2[(size_t)(row0 + i) * p.L2 + j2] = a;
      }
    }
  }
}

constexpr int kUpRows = 16;     // one 16-row MFMA tile per workgroup; 4 waves x up to 4 chunks of 64 columns

// v[16 rows][C] = lat[16][L] . W^T on the fp32 matrix cores (K = L in steps of 4).  A wave owns column chunks ch = wave + 4u;
// with the interleaved column map c = 64 ch + 4 j + e (tile e, lane column j) a lane ends up with float4s of 4 consecutive
// columns for its 4 rows, so the read-modify-write of the [M][C] stream is fully coalesced 16-byte traffic.
// Epilogues: out = base + drop(v + bias)   or, with ln_x set,   out = base + LayerNorm'(v)  (dx of a LayerNorm whose output
// gradient is the rank-L product v: fuses `dn = dlat . Wd` with the LN backward of the MWSA branch, gaviko.py:231).
template <int L, int NW, int MAXU>
__global__ __launch_bounds__(NW * 64) void skinny_up_kernel(UpArgs p) {
  static_assert(L % 4 == 0, "latent width must be a multiple of 4");
  constexpr int KS = L / 4;
  __shared__ float red[NW][16][2];
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  const int C = p.C;
  const int lane = lane_id(), wave = wave_id();
  const int j = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * kUpRows;
  const int nch = (C + 63) / 64;
  const float* base = p.accumulate ? p.out : p.res;
  // ---- prefetch the stream operands of this lane: rows m0 + 4kq + q, columns 64 ch + 4 j .. +3
  f32x4 bs[MAXU][4], xs[MAXU][4];
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int c = (wave + NW * u) * 64 + 4 * j;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + 4 * kq + q;
      const bool ok = wave + NW * u < nch && c < C && m < p.M;
      bs[u][q] = (ok && base != nullptr) ? *(const f32x4*)(base + (size_t)m * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      xs[u][q] = (ok && p.ln_x != nullptr) ? *(const f32x4*)(p.ln_x + (size_t)m * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // ---- A operand: lat[m0 + j][4s + kq]
  float a[KS];
  {
    const int m = m0 + j;
    const float* src = p.lat + (size_t)min(m, p.M - 1) * L;
    if (p.lat_override != nullptr && m < p.M) {
      const int sidx = m / p.T, t = m - sidx * p.T;
      if (t < p.P) src = p.lat_override + ((size_t)sidx * p.P + t) * L;
    }
#pragma unroll
    for (int sk = 0; sk < KS; ++sk) a[sk] = (m < p.M) ? src[4 * sk + kq] : 0.f;
  }
  // ---- MFMA per chunk
  f32x4 acc[MAXU][4];
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int ch = wave + NW * u;
    const int c = ch * 64 + 4 * j;
    const bool cok = ch < nch && c < C;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[u][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ch < nch) {                                       // wave-uniform
#pragma unroll
      for (int sk = 0; sk < KS; ++sk) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (cok) {
          if (p.w_layout == 1) b4 = *(const f32x4*)(p.w + (size_t)(4 * sk + kq) * C + c);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) b4[e] = p.w[(size_t)(c + e) * L + 4 * sk + kq];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[u][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sk], b4[e], acc[u][e], 0, 0, 0);
      }
    }
  }
  if (p.ln_x == nullptr) {
    // ---- plain epilogue: out = base + drop(v + bias)
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      const int c = (wave + NW * u) * 64 + 4 * j;
      if (wave + NW * u < nch && c < C) {
        const f32x4 b4 = p.bias ? *(const f32x4*)(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = m0 + 4 * kq + q;
          if (m < p.M) {
            f32x4 v = {acc[u][0][q] + b4[0], acc[u][1][q] + b4[1], acc[u][2][q] + b4[2], acc[u][3][q] + b4[3]};
            if (p.drop_thresh != 0u) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= drop_scale(p.seed, (unsigned long long)m * C + c + e, p.drop_thresh, p.inv_keep);
            }
            v += bs[u][q];
            *(f32x4*)(p.out + (size_t)m * C + c) = v;
            if (p.out16 != nullptr) {
              bf16x4 h = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
              *(bf16x4*)(p.out16 + (size_t)m * C + c) = h;
            }
          }
        }
      }
    }
    return;
  }
  // ---- LayerNorm-backward epilogue: dx = base + rstd * (g*v - mean(g*v) - xhat * mean(g*v*xhat))
  float mu[4], rs[4], s1[4], s2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = min(m0 + 4 * kq + q, p.M - 1);
    mu[q] = p.ln_mean[m]; rs[q] = p.ln_rstd[m];
    s1[q] = 0.f; s2[q] = 0.f;
  }
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int c = (wave + NW * u) * 64 + 4 * j;
    if (wave + NW * u < nch && c < C) {
      const f32x4 g4 = *(const f32x4*)(p.ln_g + c);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dh = acc[u][e][q] * g4[e];
          const float xh = (xs[u][q][e] - mu[q]) * rs[q];
          acc[u][e][q] = dh;            // keep g*v
          xs[u][q][e] = xh;             // keep xhat
          s1[q] += dh; s2[q] += dh * xh;
        }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { s1[q] += __shfl_xor(s1[q], o, 64); s2[q] += __shfl_xor(s2[q], o, 64); }
    if (j == 0) { red[wave][4 * kq + q][0] = s1[q]; red[wave][4 * kq + q][1] = s2[q]; }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = 4 * kq + q;
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { t1 += red[w][i][0]; t2 += red[w][i][1]; }
    s1[q] = t1 / (float)C;
    s2[q] = t2 / (float)C;
  }
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int c = (wave + NW * u) * 64 + 4 * j;
    if (wave + NW * u < nch && c < C) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + 4 * kq + q;
        if (m < p.M) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs[q] * (acc[u][e][q] - s1[q] - xs[u][q][e] * s2[q]) + bs[u][q][e];
          *(f32x4*)(p.out + (size_t)m * C + c) = o;
        }
      }
    }
  }
}

struct OuterArgs {
  const float* narrow; const float* wide;                 // narrow [M][L], wide [M][C]
  const float* lat_override; int T, P;                    // as UpArgs (narrow side)
  const float* mean; const float* rstd; const float* ln_g; const float* ln_b;   // optional LN applied to `wide` on the fly
  float* scratch;                                         // [64][L+1][C]
  int M, C;
  unsigned long long seed; const unsigned long long* seed_ptr; unsigned int drop_thresh; float inv_keep;   // optional dropout mask on `wide`
  int wide_act;                                           // 1: QuickGELU applied to `wide` on the fly (DVPT: dWd = dz^T . QuickGELU(x))
  // optional second source: rows M1 .. M-1 of the reduction come from (narrow2, wide2) -- one weight fed by two token streams
  // (GPA proj_down: dW = dzx^T . G1 + dzl^T . L', gaviko.py:155-156) in ONE pass; plain rows only (no LN / dropout / override)
  const float* narrow2; const float* wide2; int M1;
};

constexpr int kSlabs = 64;          // row slabs of the small two-stage reductions (colsum, small_wgrad)
constexpr int kOuterSlabs = 64;     // row slabs of the outer-product reduction
constexpr int kOuterMaxRows = 160;  // rows of one slab staged in LDS (M <= kOuterSlabs * kOuterMaxRows)

// Partial outer products on the fp32 matrix cores: D[l][c] += sum over a slab of rows m of narrow'[m][l] * wide'[m][c], where
// narrow' has an extra column of ones (row L of D is the column sum of wide').  A wave owns 64 columns of one row slab:
// per k-step of 4 rows a lane (j = lane&15, kq = lane>>4) loads wide[m0+kq][c0 + 4j .. +3] as one float4 and feeds element e to
// MFMA e, so accumulator tile e holds the interleaved columns c0 + 4j + e.  The slab's narrow rows / LN statistics sit in LDS.
template <int L>
__global__ __launch_bounds__(256) void outer_partial_kernel(OuterArgs p) {
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  constexpr int NT = (L + 1 + 15) / 16;              // 16-row tiles of the (L+1)-row result
  __shared__ float nar[kOuterMaxRows][NT * 16];
  __shared__ float st[kOuterMaxRows][2];
  const int C = p.C, slab = blockIdx.y;
  const int lane = lane_id(), wave = wave_id();
  const int j = lane & 15, kq = lane >> 4;
  const int c0 = (blockIdx.x * 4 + wave) * 64, c = c0 + 4 * j;
  const int rows_per = (p.M + kOuterSlabs - 1) / kOuterSlabs;
  const int r0 = slab * rows_per, r1 = min(p.M, r0 + rows_per);
  const int nr = max(0, r1 - r0);
  // the slab's narrow rows -> LDS, four elements per thread requested before the first is stored (one element per pass was a chain of
  // nr / 8 dependent round trips in front of everything else the workgroup does)
  for (int i0 = threadIdx.x; i0 < nr * NT * 16; i0 += 4 * 256) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256;
      const int r = i / (NT * 16), l = i - r * (NT * 16), m = r0 + r;
      v[u] = (l == L) ? 1.f : 0.f;
      if (i < nr * NT * 16 && l < L) {
        const float* src = (p.narrow2 != nullptr && m >= p.M1) ? p.narrow2 + (size_t)(m - p.M1) * L : p.narrow + (size_t)m * L;
        if (p.lat_override != nullptr) {
          const int s = m / p.T, t = m - s * p.T;
          if (t < p.P) src = p.lat_override + ((size_t)s * p.P + t) * L;
        }
        v[u] = src[l];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256;
      if (i < nr * NT * 16) nar[i / (NT * 16)][i % (NT * 16)] = v[u];
    }
  }
  if (p.mean != nullptr && (int)threadIdx.x < nr) {
    st[threadIdx.x][0] = p.mean[r0 + threadIdx.x];
    st[threadIdx.x][1] = p.rstd[r0 + threadIdx.x];
  }
  __syncthreads();
  if (c0 >= C) return;
  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool cok = c < C;
  f32x4 g4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
  if (p.ln_g != nullptr && cok) { g4 = *(const f32x4*)(p.ln_g + c); b4 = *(const f32x4*)(p.ln_b + c); }
  // 16 rows (4 k-steps) per chunk, kPF chunks (64 rows) requested ahead of the MFMAs that consume them: one wave per SIMD has nothing
  // else to hide the HBM round trip behind, and a 65-row slab is then ONE round trip instead of five
  constexpr int kPF = 4;
  auto fetch = [&](int rb, f32x4 (&x)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = rb + 4 * u + kq;
      const int gm = r0 + r;
      const float* wrow = (p.narrow2 != nullptr && gm >= p.M1) ? p.wide2 + (size_t)(gm - p.M1) * C : p.wide + (size_t)gm * C;
      x[u] = (r < nr && cok) ? *(const f32x4*)(wrow + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x4 xr[kPF][4];
#pragma unroll
  for (int i = 0; i < kPF; ++i) fetch(16 * i, xr[i]);
  for (int rb0 = 0; rb0 < nr; rb0 += 16 * kPF) {
#pragma unroll
   for (int pi = 0; pi < kPF; ++pi) {
    const int rb = rb0 + 16 * pi;
    if (rb >= nr) break;                                 // wave-uniform
    f32x4 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = xr[pi][u];
    fetch(rb + 16 * kPF, xr[pi]);                        // refill this slot with the chunk kPF ahead (zeros past the slab)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = rb + 4 * u + kq;
      const bool ok = r < nr && cok;
      f32x4 xv = x[u];
      if (ok) {
        if (p.drop_thresh != 0u) {
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[e] *= drop_scale(p.seed, (unsigned long long)(r0 + r) * C + c + e, p.drop_thresh, p.inv_keep);
        }
        if (p.mean != nullptr) {
          const float mu = st[r][0], rs = st[r][1];
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[e] = (xv[e] - mu) * rs * g4[e] + b4[e];
        }
        if (p.wide_act == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[e] = quick_gelu(xv[e]);
        }
      }
      if (rb + 4 * u < nr) {                            // wave-uniform: skip k-steps entirely past the slab
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float a = (r < nr) ? nar[r][t * 16 + j] : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[e], acc[t][e], 0, 0, 0);
        }
      }
    }
   }
  }
  if (!cok) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int l = t * 16 + 4 * kq + q;               // D row = 4*(lane>>4) + reg
      if (l <= L) {
        f32x4 o = {acc[t][0][q], acc[t][1][q], acc[t][2][q], acc[t][3][q]};   // columns c .. c+3
        *(f32x4*)(p.scratch + ((size_t)slab * (L + 1) + l) * C + c) = o;
      }
    }
}

// out[l][c] (transposed=0) or out[c][l] (transposed=1); colsum[c] optional.  Block = 64 columns x 4 groups of 16 slabs (all 16 loads of a
// thread in flight; the serial 64-deep sum was a 6 us latency chain behind every outer product), LDS reduce in a fixed order: deterministic
__global__ __launch_bounds__(256) void outer_final_kernel(const float* __restrict__ scratch, float* __restrict__ out, float* __restrict__ colsum,
                                                          int L, int C, int transposed, int accumulate) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, l = blockIdx.y;
  float a = 0.f;
  if (c < C) {
    float v[kOuterSlabs / 4];
#pragma unroll
    for (int u = 0; u < kOuterSlabs / 4; ++u) v[u] = scratch[((size_t)(sg * (kOuterSlabs / 4) + u) * (L + 1) + l) * C + c];
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < kOuterSlabs / 4; ++u) a4[u & 3] += v[u];
    a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  }
  part[sg][cl] = a;
  __syncthreads();
  if (sg != 0 || c >= C) return;
  a = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
  if (l < L) {
    if (out == nullptr) return;
    float* o = transposed ? out + (size_t)c * L + l : out + (size_t)l * C + c;
    *o = accumulate ? *o + a : a;
  } else if (colsum != nullptr) {
    colsum[c] = accumulate ? colsum[c] + a : a;
  }
}

// dW[j][l] = sum_m a[m][j] * b[m][l], J, Lb <= 64: stage 1 per row slab, stage 2 sums slabs.
__global__ __launch_bounds__(256) void small_wgrad_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ scratch,
                                                                  int M, int J, int Lb) {
  const int slab = blockIdx.x, nout = J * Lb;
  const int rows_per = (M + kSlabs - 1) / kSlabs;
  const int r0 = slab * rows_per, r1 = min(M, r0 + rows_per);
  for (int o = threadIdx.x; o < nout; o += 256) {
    const int j = o / Lb, l = o - j * Lb;
    float acc = 0.f;
    for (int m = r0; m < r1; m += 8) {
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool ok = m + u < r1;
        av[u] = ok ? a[(size_t)(m + u) * J + j] : 0.f;
        bv[u] = ok ? b[(size_t)(m + u) * Lb + l] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += av[u] * bv[u];
    }
    scratch[(size_t)slab * nout + o] = acc;
  }
}
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ scratch, float* __restrict__ out, int n, int nslabs, int accumulate) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  float a4[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < nslabs; s += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) a4[u] += (s + u < nslabs) ? scratch[(size_t)(s + u) * n + o] : 0.f;
  }
  const float a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  out[o] = accumulate ? out[o] + a : a;
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ scratch, int M, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x, slab = blockIdx.y;
  if (c >= C) return;
  const int rows_per = (M + kSlabs - 1) / kSlabs;
  const int r0 = slab * rows_per, r1 = min(M, r0 + rows_per);
  float a = 0.f;
  for (int m = r0; m < r1; m += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (m + u < r1) ? x[(size_t)(m + u) * C + c] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  scratch[(size_t)slab * C + c] = a;
}

// ---- batched small reductions: several independent column sums / (J x L) weight-gradient products in TWO launches.
// Deterministic: stage 1 = (job, 64-output chunk, row slab) workgroups whose 256 threads are (output, row slice) pairs combined through
// LDS in a fixed order; stage 2 sums every output's partials.  The number of row slabs is PER JOB (one per 128 rows, at most kRedSlabs):
// the GPA's gate-parameter job sums 4 rows into 3525 outputs and used to launch 56 x 32 workgroups of 1024 threads for it, 28 of every 32
// empty -- 2336 workgroups (2.4 M threads, 15-17 us on the GPA stream beside the dgrad GEMMs) for 110 workgroups of work.
constexpr int kMaxJobs = 8;
constexpr int kRedSlabs = 32;
struct ReduceJob { const float* a; const float* b; float* out; int M, J, L, accumulate, wg0, o_base; const float* a2; int M1; int nslab; };   // a2: rows M1.. of a column sum
struct ReduceBatch { ReduceJob job[kMaxJobs]; int njobs; float* scratch; int total_out; };

__global__ __launch_bounds__(256) void reduce_batch_partial_kernel(ReduceBatch bt) {
  __shared__ float red[256];
  int ji = 0;
#pragma unroll
  for (int k = 1; k < kMaxJobs; ++k)
    if (k < bt.njobs && (int)blockIdx.x >= bt.job[k].wg0) ji = k;
  const ReduceJob jb = bt.job[ji];
  const int local = (int)blockIdx.x - jb.wg0;
  const int slab = local % jb.nslab;
  const int nout = jb.b ? jb.J * jb.L : jb.J;
  const int o0 = (local / jb.nslab) * 64;
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
  red[threadIdx.x] = acc;
  __syncthreads();
  if (sl == 0 && oi < no)
    bt.scratch[(size_t)slab * bt.total_out + jb.o_base + o0 + oi] = (red[oi] + red[64 + oi]) + (red[128 + oi] + red[192 + oi]);
}

__global__ __launch_bounds__(256) void reduce_batch_final_kernel(ReduceBatch bt) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= bt.total_out) return;
  int ji = 0;
#pragma unroll
  for (int k = 1; k < kMaxJobs; ++k)
    if (k < bt.njobs && g >= bt.job[k].o_base) ji = k;
  const int ns = bt.job[ji].nslab;
  float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < kRedSlabs; s += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) a4[u] += (s + u < ns) ? bt.scratch[(size_t)(s + u) * bt.total_out + g] : 0.f;
  }
  const float t = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  float* dst = bt.job[ji].out + (g - bt.job[ji].o_base);
  *dst = bt.job[ji].accumulate ? *dst + t : t;
}

// Affine / weight gradients of  y = LN(x) . Wd^T  from Q[l][c] = sum_m dlat[m][l] xhat[m][c] and S[l] = sum_m dlat[m][l]:
//   dWd[l][c] = g_c Q[l][c] + b_c S[l],  dgamma_c = sum_l Wd[l][c] Q[l][c],  dbeta_c = sum_l Wd[l][c] S[l],  dbias_l = S[l]
// (replaces materialising dn = dlat . Wd for a separate LayerNorm-affine reduction).
__global__ __launch_bounds__(256) void ln_lowrank_affine_kernel(const float* __restrict__ Q, const float* __restrict__ S, const float* __restrict__ W,
                                                                const float* __restrict__ g, const float* __restrict__ b, float* __restrict__ dW,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias,
                                                                int L, int C, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    const float gc = g[c], bc = b[c];
    float dg = 0.f, db = 0.f;
    for (int l0 = 0; l0 < L; l0 += 8) {                     // eight rows' loads in flight per round trip (one thread per column: the
      float q[8], w[8], o0[8];                              // row-by-row form was a 20-deep latency chain, 15 us on the MWSA stream)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool ok = l0 + u < L;
        q[u] = ok ? Q[(size_t)(l0 + u) * C + c] : 0.f;
        w[u] = ok ? W[(size_t)(l0 + u) * C + c] : 0.f;
        o0[u] = (ok && accumulate) ? dW[(size_t)(l0 + u) * C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (l0 + u < L) {
          const float sl = S[l0 + u];
          dW[(size_t)(l0 + u) * C + c] = o0[u] + (gc * q[u] + bc * sl);
          dg += w[u] * q[u];
          db += w[u] * sl;
        }
      }
    }
    dgamma[c] = accumulate ? dgamma[c] + dg : dg;
    dbeta[c] = accumulate ? dbeta[c] + db : db;
  }
  if (c < L && dbias != nullptr) dbias[c] = accumulate ? dbias[c] + S[c] : S[c];
}

template <int L>
static int launch_down(const DownArgs& a, hipStream_t s) {
  const int lds_w = L * (a.C + 4), lds_p = 16 * 16 * 32 + 16 * 32;      // floats: staged W  vs  partial tiles + finished rows
  const int lds = ((64 + 16 * 16 * 2) + (lds_w > lds_p ? lds_w : lds_p)) * 4;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_down_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(skinny_down): %s", hipGetErrorString(e));
    attr = true;
  }
  GVK_LAUNCH((skinny_down_kernel<L>), dim3((a.M + kDownRows - 1) / kDownRows), dim3(1024), lds, s, a);
  return check_launch("skinny_down");
}
template <int L>
static int launch_up(const UpArgs& a, hipStream_t s) {
  const int nch = (a.C + 63) / 64;
  const dim3 grid((a.M + kUpRows - 1) / kUpRows);
  if (nch > 12) GVK_LAUNCH((skinny_up_kernel<L, 8, 2>), grid, dim3(512), 0, s, a);        // C <= 1024
  else if (nch > 8) GVK_LAUNCH((skinny_up_kernel<L, 4, 3>), grid, dim3(256), 0, s, a);    // C <= 768
  else if (nch > 4) GVK_LAUNCH((skinny_up_kernel<L, 4, 2>), grid, dim3(256), 0, s, a);    // C <= 512
  else GVK_LAUNCH((skinny_up_kernel<L, 4, 1>), grid, dim3(256), 0, s, a);                 // C <= 256
  return check_launch("skinny_up");
}
template <int L>
static int launch_outer(const OuterArgs& a, hipStream_t s) {
  GVK_LAUNCH((outer_partial_kernel<L>), dim3((a.C + 255) / 256, kOuterSlabs), dim3(256), 0, s, a);   // 4 waves x 64 columns
  return check_launch("outer_partial");
}

static unsigned int drop_threshold(float p) {
  if (p <= 0.f) return 0u;
  double t = (double)p * 4294967296.0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (unsigned int)t;
}

}  // namespace gvk

extern "C" int gvk_skinny_down(const gvk_skinny_down_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->x && d->w && (d->y || d->z || d->y2), "gvk_skinny_down: null pointer");
  GVK_REQUIRE(d->M > 0 && d->C > 0 && d->C % 4 == 0 && d->C <= 1024, "gvk_skinny_down: C=%d must be a multiple of 4 and <= 1024", d->C);
  GVK_REQUIRE((d->L * (d->C + 4) + 576) * 4 <= 160 * 1024, "gvk_skinny_down: W does not fit the 160 KiB LDS");
  GVK_REQUIRE((d->ln_gamma == nullptr) == (d->ln_beta == nullptr), "gvk_skinny_down: LN gamma/beta must come together");
  GVK_REQUIRE(d->w2 == nullptr || (d->y2 != nullptr && d->L2 > 0), "gvk_skinny_down: second stage needs y2 and L2");
  DownArgs a{};
  a.x = d->x; a.w = d->w; a.bias = d->bias; a.ln_g = d->ln_gamma; a.ln_b = d->ln_beta; a.mean = d->mean; a.rstd = d->rstd;
  a.z = d->z; a.y = d->y; a.w2 = d->w2; a.y2 = d->y2; a.L2 = d->L2; a.M = d->M; a.C = d->C; a.act = d->act; a.w_layout = d->w_layout;
  a.eps = d->eps > 0.f ? d->eps : 1e-5f;
  a.seed = d->seed; a.seed_ptr = (const unsigned long long*)d->seed_ptr; a.drop_thresh = drop_threshold(d->drop_p); a.inv_keep = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
  hipStream_t s = (hipStream_t)stream;
  static const bool mfma_only = diag_env("GAVIKO_HIP_SKINNY_MFMA") != nullptr;     // A/B switch: force the MFMA-tile kernels
  if (d->act_in != 0) {
    GVK_REQUIRE(d->act_in == 1 && d->ln_gamma == nullptr && d->drop_p <= 0.f, "gvk_skinny_down: act_in=1 (QuickGELU on the input) takes no LN / dropout");
    a.mode = 3;
    int rc = launch_side_down(a, d->L, s);
    if (rc == 1) rc = launch_row_down(a, d->L, s);
    if (rc == 1) return set_error(-2, "gvk_skinny_down: act_in needs the row-per-wave kernel (L in {4,8,16,20}, C >= 128)");
    return rc;
  }
  if (!mfma_only) {
    int rc = launch_side_down(a, d->L, s);                                        // 16-row tiles on the fp32 matrix cores
    if (rc != 1) return rc;
    rc = launch_row_down(a, d->L, s);                                             // row-per-wave form for the wide shapes
    if (rc != 1) return rc;
  }
  switch (d->L) {
    case 4: return launch_down<4>(a, s);
    case 8: return launch_down<8>(a, s);
    case 16: return launch_down<16>(a, s);
    case 20: return launch_down<20>(a, s);
    case 32: return launch_down<32>(a, s);
    default: return set_error(-2, "gvk_skinny_down: L=%d unsupported (4, 8, 16, 20, 32)", d->L);
  }
}

namespace gvk {
// out[b*T + p][:] += (enh[b][p][:] - lat[b*T + p][:]) . W^T for the P prompt rows of every sample: when the up-projection of the plain
// latents rides a GEMM (K-concatenation), only the prompt rows -- whose latents the GPA replaces (gaviko.py:183-187) -- are left to fix.
__global__ __launch_bounds__(256) void prompt_up_fix_kernel(const float* __restrict__ enh, const float* __restrict__ lat, const float* __restrict__ w,
                                                            float* __restrict__ out, int T, int P, int C, int L) {
  __shared__ float dl[64];
  const int b = blockIdx.y, pi = blockIdx.x;
  const size_t row = (size_t)b * T + pi;
  if ((int)threadIdx.x < L) dl[threadIdx.x] = enh[((size_t)b * P + pi) * L + threadIdx.x] - lat[row * L + threadIdx.x];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int l = 0; l < L; ++l) a = __builtin_fmaf(dl[l], w[(size_t)c * L + l], a);
    out[row * C + c] += a;
  }
}
}  // namespace gvk

namespace gvk {
// gvk_prompt_up_fix for a layer whose first LayerNorm is folded into the qkv projection (gvk_gemm_desc.ln_mean): the fc2 GEMM in front of
// this kernel has left the fp32 rows, their bf16 copy and per-row (sum, sum of squares) partials over 64-column groups.
//   blocks [0, P*B):      fix one prompt row (as prompt_up_fix_kernel), rewrite its bf16 copy and compute its mean / rstd from the row itself
//   blocks [P*B, ...):    one thread per remaining row: mean / rstd from the GEMM's partials (prompt rows are left to the blocks above)
__global__ __launch_bounds__(256) void prompt_fix_stats_kernel(const float* __restrict__ enh, const float* __restrict__ lat, const float* __restrict__ w,
                                                               float* __restrict__ out, bf16* __restrict__ out16, const float* __restrict__ part,
                                                               int nparts, const float* __restrict__ pivot, float* __restrict__ mean,
                                                               float* __restrict__ rstd, int B, int T, int P,
                                                               int C, int L, float eps) {
  const int M = B * T;
  if ((int)blockIdx.x >= P * B) {
    const int m = ((int)blockIdx.x - P * B) * 256 + threadIdx.x;
    if (m >= M || (m % T) < P) return;
    float s1 = 0.f, s2 = 0.f;
    for (int g = 0; g < nparts; ++g) {                          // fixed order: deterministic
      const f32x2 v = *(const f32x2*)(part + ((size_t)g * M + m) * 2);
      s1 += v[0]; s2 += v[1];
    }
    // the partials are sums of d = x - pivot and d^2 with pivot ~ the row mean: E[d^2] - E[d]^2 has nothing to cancel (the unshifted
    // single-pass form loses the variance when |mean| >> std: 1/sqrt(eps) instead of rstd)
    const float md = s1 / (float)C;
    const float var = fmaxf(s2 / (float)C - md * md, 0.f);
    const float mu = (pivot != nullptr ? pivot[m] : 0.f) + md;
    mean[m] = mu;
    rstd[m] = 1.0f / sqrtf(var + eps);
    return;
  }
  __shared__ float dl[64];
  __shared__ float red[8];
  const int b = blockIdx.x / P, pi = blockIdx.x - b * P;
  const size_t row = (size_t)b * T + pi;
  // every load of the block goes out before the first use: the row's own columns and the weight rows (w [C][L], L floats contiguous per
  // column: float4 pieces when L % 4 == 0) do not depend on the latent difference
  float x[4] = {0.f, 0.f, 0.f, 0.f};                            // C <= 1024: up to four columns per thread
  f32x4 wv[4][5];                                               // L <= 20 in float4 pieces (the wider / odd widths take the scalar loop below)
  const bool vec = (L & 3) == 0 && L <= 20;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = threadIdx.x + 256 * u;
    if (c < C) {
      x[u] = out[row * C + c];
      if (vec) {
#pragma unroll
        for (int q = 0; q < 5; ++q) wv[u][q] = 4 * q < L ? *(const f32x4*)(w + (size_t)c * L + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  if ((int)threadIdx.x < L) dl[threadIdx.x] = enh[((size_t)b * P + pi) * L + threadIdx.x] - lat[row * L + threadIdx.x];
  __syncthreads();
  float s1 = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = threadIdx.x + 256 * u;
    if (c < C) {
      float a = 0.f;
      if (vec) {
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (4 * q + e < L) a = __builtin_fmaf(dl[4 * q + e], wv[u][q][e], a);      // same order as the scalar loop: l ascending
      } else {
        for (int l = 0; l < L; ++l) a = __builtin_fmaf(dl[l], w[(size_t)c * L + l], a);
      }
      x[u] += a;
      out[row * C + c] = x[u];
      out16[row * C + c] = (bf16)x[u];
      s1 += x[u];
    }
  }
  // two-pass statistics of the row (as the LayerNorm kernel takes them): block sum -> mean -> block sum of squared deviations
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  const float mu = block_sum(s1) / (float)C;
  float s2 = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if ((int)threadIdx.x + 256 * u < C) s2 = __builtin_fmaf(x[u] - mu, x[u] - mu, s2);
  const float var = block_sum(s2) / (float)C;
  if (threadIdx.x == 0) { mean[row] = mu; rstd[row] = 1.0f / sqrtf(var + eps); }
}
}  // namespace gvk

extern "C" int gvk_prompt_up_fix_stats(const float* enh, const float* lat, const float* w, float* out, void* out16, const float* part, int nparts,
                                       const float* pivot, float* mean, float* rstd, int B, int T, int P, int C, int L, float eps, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(enh && lat && w && out && out16 && part && mean && rstd, "gvk_prompt_up_fix_stats: null pointer");
  GVK_REQUIRE(B > 0 && P > 0 && P <= T && C > 0 && C <= 1024 && L > 0 && L <= 64 && nparts > 0, "gvk_prompt_up_fix_stats: bad arguments");
  const int nblk = P * B + (B * T + 255) / 256;
  GVK_LAUNCH(prompt_fix_stats_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, enh, lat, w, out, (bf16*)out16, part, nparts, pivot, mean, rstd, B, T, P,
             C, L, eps > 0.f ? eps : 1e-5f);
  return check_launch("prompt_up_fix_stats");
}

extern "C" int gvk_prompt_up_fix(const float* enh, const float* lat, const float* w, float* out, int B, int T, int P, int C, int L, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(enh && lat && w && out && B > 0 && P > 0 && P <= T && C > 0 && L > 0 && L <= 64, "gvk_prompt_up_fix: bad arguments");
  GVK_LAUNCH(prompt_up_fix_kernel, dim3(P, B), dim3(256), 0, (hipStream_t)stream, enh, lat, w, out, T, P, C, L);
  return check_launch("prompt_up_fix");
}

extern "C" int gvk_skinny_up(const gvk_skinny_up_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->lat && d->w && d->out, "gvk_skinny_up: null pointer");
  GVK_REQUIRE(d->M > 0 && d->C > 0 && d->C <= 1024 && d->C % 4 == 0, "gvk_skinny_up: C=%d must be a multiple of 4 and <= 1024", d->C);
  GVK_REQUIRE(d->ln_x == nullptr || (d->ln_mean && d->ln_rstd && d->ln_gamma && d->bias == nullptr && d->drop_p <= 0.f),
              "gvk_skinny_up: the LayerNorm-backward epilogue needs mean/rstd/gamma and takes no bias / dropout");
  GVK_REQUIRE(d->lat_override == nullptr || (d->T > 0 && d->P > 0 && d->P <= d->T), "gvk_skinny_up: override needs 0 < P <= T");
  UpArgs a{};
  a.lat = d->lat; a.w = d->w; a.bias = d->bias; a.res = d->res; a.out = d->out; a.lat_override = d->lat_override; a.T = d->T; a.P = d->P;
  a.M = d->M; a.C = d->C; a.w_layout = d->w_layout; a.accumulate = d->accumulate;
  a.ln_x = d->ln_x; a.ln_mean = d->ln_mean; a.ln_rstd = d->ln_rstd; a.ln_g = d->ln_gamma; a.out16 = (bf16*)d->out_bf16;
  a.seed = d->seed; a.seed_ptr = (const unsigned long long*)d->seed_ptr; a.drop_thresh = drop_threshold(d->drop_p); a.inv_keep = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
  hipStream_t s = (hipStream_t)stream;
  static const bool mfma_only = diag_env("GAVIKO_HIP_SKINNY_MFMA") != nullptr;
  a.alpha_ptr = d->alpha_ptr; a.gg_x = d->gg_x;
  GVK_REQUIRE(d->w2 == nullptr || ((d->z2 || d->y2) && d->L2 > 0), "gvk_skinny_up: the second projection needs z2 or y2 and L2");
  if (d->lat_b != nullptr) {                            // the layer-boundary form: 16-row-tile kernel only
    GVK_REQUIRE(d->w_b && d->ln_x && d->w2 && (d->w2_layout == 0 || d->w2_layout == 1), "gvk_skinny_up: lat_b needs w_b, the LayerNorm operands and w2");
    UpExtra ex{};
    ex.lat2 = d->lat_b; ex.w2up = d->w_b; ex.w2_layout = d->w2_layout;
    ex.seed2 = d->seed2; ex.seed_ptr = (const unsigned long long*)d->seed_ptr; ex.drop2_thresh = drop_threshold(d->drop2_p);
    ex.inv_keep2 = d->drop2_p > 0.f ? 1.f / (1.f - d->drop2_p) : 1.f;
    const int rc = launch_side_up(a, d->L, d->w2, d->bias2, d->z2, d->y2, d->L2, d->act2, s, nullptr, &ex);
    if (rc == 1) return set_error(-2, "gvk_skinny_up: lat_b is built for L = 20 and C in {192, 768, 1024} (got L=%d, C=%d)", d->L, d->C);
    return rc;
  }
  if (d->nx_w != nullptr) {                             // the next layer's MWSA entry on the produced rows: 16-row-tile kernel only
    GVK_REQUIRE(d->ln_x == nullptr && d->lat_b == nullptr, "gvk_skinny_up: nx_w takes the plain epilogue");
    UpNext nx{d->nx_w, d->nx_bias, d->nx_ln_gamma, d->nx_ln_beta, d->nx_mean, d->nx_rstd, d->nx_lat, d->nx_w2, d->nx_y2, d->nx_L2, d->nx_eps};
    const int rc = launch_side_up(a, d->L, d->w2, d->bias2, d->z2, d->y2, d->L2, d->act2, s, nullptr, nullptr, &nx);
    if (rc == 1) return set_error(-2, "gvk_skinny_up: nx_w is built for L = 20 and C in {192, 768, 1024} (got L=%d, C=%d)", d->L, d->C);
    return rc;
  }
  if (!mfma_only) {
    const int rc = launch_side_up(a, d->L, d->w2, d->bias2, d->z2, d->y2, d->L2, d->act2, s);
    if (rc != 1) return rc;
  }
  GVK_REQUIRE(d->w2 == nullptr, "gvk_skinny_up: the fused second projection needs the 16-row-tile kernel (C %% 32 == 0, L %% 4 == 0, L <= 32)");
  if (a.alpha_ptr != nullptr || a.gg_x != nullptr) {
    GVK_REQUIRE(d->ln_x == nullptr, "gvk_skinny_up: alpha_ptr / gg_x do not combine with the LayerNorm-backward epilogue");
    const int rc = launch_row_up(a, d->L, s);
    if (rc == 1) return set_error(-2, "gvk_skinny_up: alpha_ptr / gg_x need the row-per-wave kernel (L in {4,8,16,20}, C >= 128)");
    return rc;
  }
  if (!mfma_only) {
    const int rc = launch_row_up(a, d->L, s);
    if (rc != 1) return rc;
  }
  switch (d->L) {
    case 4: return launch_up<4>(a, s);
    case 8: return launch_up<8>(a, s);
    case 16: return launch_up<16>(a, s);
    case 20: return launch_up<20>(a, s);
    case 32: return launch_up<32>(a, s);
    default: return set_error(-2, "gvk_skinny_up: L=%d unsupported (4, 8, 16, 20, 32)", d->L);
  }
}

extern "C" int gvk_outer_reduce(const gvk_outer_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->narrow && d->wide && d->scratch && (d->out || d->colsum), "gvk_outer_reduce: null pointer");
  GVK_REQUIRE(d->M > 0 && d->C > 0 && d->C % 4 == 0, "gvk_outer_reduce: C must be a positive multiple of 4");
  GVK_REQUIRE(d->M <= kOuterSlabs * kOuterMaxRows, "gvk_outer_reduce: M=%d exceeds %d rows", d->M, kOuterSlabs * kOuterMaxRows);
  GVK_REQUIRE((d->mean == nullptr) == (d->rstd == nullptr), "gvk_outer_reduce: mean/rstd must come together");
  OuterArgs a{};
  a.narrow = d->narrow; a.wide = d->wide; a.lat_override = d->lat_override; a.T = d->T; a.P = d->P;
  a.mean = d->mean; a.rstd = d->rstd; a.ln_g = d->ln_gamma; a.ln_b = d->ln_beta; a.scratch = d->scratch; a.M = d->M; a.C = d->C;
  a.wide_act = d->wide_act;
  a.narrow2 = d->narrow2; a.wide2 = d->wide2; a.M1 = d->M;
  if (d->narrow2 != nullptr) {
    GVK_REQUIRE(d->wide2 != nullptr && d->M2 > 0 && d->mean == nullptr && d->lat_override == nullptr && d->drop_p <= 0.f && d->wide_act == 0,
                "gvk_outer_reduce: the second source takes plain rows only (no LN, override, dropout, activation)");
    a.M = d->M + d->M2;
    GVK_REQUIRE(a.M <= kOuterSlabs * kOuterMaxRows, "gvk_outer_reduce: M + M2 = %d exceeds %d rows", a.M, kOuterSlabs * kOuterMaxRows);
  }
  a.seed = d->seed; a.seed_ptr = (const unsigned long long*)d->seed_ptr; a.drop_thresh = drop_threshold(d->drop_p); a.inv_keep = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  switch (d->L) {
    case 4: rc = launch_outer<4>(a, s); break;
    case 8: rc = launch_outer<8>(a, s); break;
    case 16: rc = launch_outer<16>(a, s); break;
    case 20: rc = launch_outer<20>(a, s); break;
    case 24: rc = launch_outer<24>(a, s); break;
    case 32: rc = launch_outer<32>(a, s); break;
    case 64: rc = launch_outer<64>(a, s); break;
    default: return set_error(-2, "gvk_outer_reduce: L=%d unsupported (4, 8, 16, 20, 32, 64)", d->L);
  }
  if (rc) return rc;
  GVK_LAUNCH(outer_final_kernel, dim3((d->C + 63) / 64, d->L + 1), dim3(256), 0, s, d->scratch, d->out, d->colsum, d->L, d->C,
                     d->transposed, d->accumulate);
  return check_launch("outer_final");
}

extern "C" int gvk_small_wgrad(const float* a, const float* b, float* out, float* scratch, int M, int J, int L, int accumulate, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(a && b && out && scratch && M > 0 && J > 0 && L > 0, "gvk_small_wgrad: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  GVK_LAUNCH(small_wgrad_partial_kernel, dim3(kSlabs), dim3(256), 0, s, a, b, scratch, M, J, L);
  int rc = check_launch("small_wgrad_partial");
  if (rc) return rc;
  GVK_LAUNCH(slab_sum_kernel, dim3((J * L + 255) / 256), dim3(256), 0, s, scratch, out, J * L, kSlabs, accumulate);
  return check_launch("small_wgrad_final");
}

extern "C" int gvk_colsum(const float* x, float* out, float* scratch, int M, int C, int accumulate, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x && out && scratch && M > 0 && C > 0, "gvk_colsum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  GVK_LAUNCH(colsum_partial_kernel, dim3((C + 255) / 256, kSlabs), dim3(256), 0, s, x, scratch, M, C);
  int rc = check_launch("colsum_partial");
  if (rc) return rc;
  GVK_LAUNCH(slab_sum_kernel, dim3((C + 255) / 256), dim3(256), 0, s, scratch, out, C, kSlabs, accumulate);
  return check_launch("colsum_final");
}

extern "C" int gvk_reduce_batch(const gvk_reduce_job* jobs, int njobs, float* scratch, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(jobs && scratch && njobs > 0 && njobs <= kMaxJobs, "gvk_reduce_batch: 1..%d jobs per launch, scratch required", kMaxJobs);
  ReduceBatch bt{};
  bt.njobs = njobs;
  bt.scratch = scratch;
  int wg = 0, ob = 0;
  for (int k = 0; k < njobs; ++k) {
    const gvk_reduce_job& j = jobs[k];
    GVK_REQUIRE(j.a && j.out && j.M > 0 && j.J > 0 && (j.b == nullptr || j.L > 0), "gvk_reduce_batch: job %d malformed", k);
    GVK_REQUIRE(j.a2 == nullptr || (j.b == nullptr && j.M2 > 0), "gvk_reduce_batch: job %d: a second source (a2, M2) goes with a column sum only", k);
    const int nout = j.b ? j.J * j.L : j.J;
    const int rows = j.M + (j.a2 ? j.M2 : 0);
    const int nslab = std::min(kRedSlabs, std::max(1, (rows + 127) / 128));
    bt.job[k] = ReduceJob{j.a, j.b, j.out, rows, j.J, j.L, j.accumulate, wg, ob, j.a2, j.a2 ? j.M : 0x7fffffff, nslab};
    wg += ((nout + 63) / 64) * nslab;
    ob += nout;
  }
  bt.total_out = ob;
  hipStream_t s = (hipStream_t)stream;
  GVK_LAUNCH(reduce_batch_partial_kernel, dim3(wg), dim3(256), 0, s, bt);
  int rc = check_launch("reduce_batch/partial");
  if (rc) return rc;
  GVK_LAUNCH(reduce_batch_final_kernel, dim3((ob + 255) / 256), dim3(256), 0, s, bt);
  return check_launch("reduce_batch/final");
}

extern "C" int gvk_ln_lowrank_affine(const float* Q, const float* S, const float* W, const float* gamma, const float* beta, float* dW,
                                     float* dgamma, float* dbeta, float* dbias, int L, int C, int accumulate, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(Q && S && W && gamma && beta && dW && dgamma && dbeta && L > 0 && C >= L, "gvk_ln_lowrank_affine: bad arguments");
  GVK_LAUNCH(ln_lowrank_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, Q, S, W, gamma, beta, dW, dgamma,
                     dbeta, dbias, L, C, accumulate);
  return check_launch("ln_lowrank_affine");
}
