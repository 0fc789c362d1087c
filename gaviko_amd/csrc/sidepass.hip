// Rank-L projections of the GAViKO side paths on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate).
//
//   side_down:  y[m][0:L] = act( f(x[m][0:C]) . W^T + bias )        f = identity | LayerNorm | dropout mask | QuickGELU      (gaviko.py:155-156,231-232)
//   side_up:    out[m][0:C] = base + g( lat[m][0:L] . W^T + bias )   g = identity | dropout | LayerNorm-backward epilogue      (gaviko.py:187,242-243)
//               optionally followed by a second down-projection of the rows it has just produced (GPA's proj_down of the new local tokens)
//
// Both work on tiles of 16 token rows, one 512-thread workgroup per tile (M = 4132 -> 259 workgroups, one round over the 256 CUs), and
// split the C axis over the eight waves, so that
//   * every wide row is read once, straight into the registers of the MFMA operand it feeds (no LDS staging of x or W: the kernels keep
//     under 20 KiB of LDS and co-reside with the backbone's GEMM workgroups),
//   * the reduction over C happens inside the MFMA accumulators + ONE cross-wave pass through LDS per tile (the row-per-wave kernels in
//     rowwise.hip spend 20 wave reductions per row and re-stage the weight chunk by chunk behind barriers),
//   * all of a tile's loads (48 KiB of x per workgroup) are in flight together.
// k-order inside an MFMA is free (the sum runs over all of it), so an operand register is whatever 16-byte piece loads best:
//   down: lane (i = lane & 15, kq = lane >> 4) holds x[row i][c .. c+3], c = 32 g + 16 h + 4 kq -- A operand, element e feeds MFMA e;
//         the B operand is W[n = lane & 15 (+16)][same c .. c+3].
//   up:   D = W^T-fragment (A) x lat^T (B): lane (a = lane & 15, kq) holds W[c(a, j)][kq L/4 + s], the B operand lat[row lane & 15][kq L/4 + s];
//         the weight rows of column tiles 2p, 2p+1 are interleaved as in gemm_epilogue.hpp, so a lane ends up with EIGHT consecutive output
//         columns of one token row (32-byte pieces, 128 contiguous bytes per row and instruction pair).
#include "skinny_args.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

namespace {
constexpr int kSW = 8;                 // waves per workgroup
constexpr int kMaxG = 4;               // 32-column groups per wave: C <= 1024

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// sum over the four kq lanes that share lane & 15
__device__ __forceinline__ float kq_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// W fragment of the down projection: W[n][c .. c+3] for n < L, zero above.  layout 0: w [L][C]; 1: w [C][L]
__device__ __forceinline__ f32x4 load_w_down(const float* __restrict__ w, int layout, int L, int C, int n, int c) {
  if (n >= L) return f32x4{0.f, 0.f, 0.f, 0.f};
  if (layout == 0) return *(const f32x4*)(w + (size_t)n * C + c);
  return f32x4{w[(size_t)c * L + n], w[(size_t)(c + 1) * L + n], w[(size_t)(c + 2) * L + n], w[(size_t)(c + 3) * L + n]};
}
}  // namespace

// MODE 0: plain rows (optional LayerNorm of the row first, optional dropout mask on the row);  3: QuickGELU of the row first
template <int MODE>
__global__ __launch_bounds__(64 * kSW) void side_down_kernel(DownArgs p, int L) {
  __shared__ float red[kSW][16];
  __shared__ f32x4 part[kSW][2][64];
  __shared__ float yrow[16][33];
  __shared__ float w2s[64 * 33];
  const int lane = lane_id(), wave = wave_id();
  const int i = lane & 15, kq = lane >> 4;
  const int C = p.C, NG = C >> 5;                       // groups of 32 columns; wave w owns groups w, w + 8, ...
  const int row0 = blockIdx.x * 16;
  const int row = min(row0 + i, p.M - 1);
  const bool ln = MODE == 0 && p.ln_g != nullptr;
  // every load of the tile goes out first
  f32x4 x[kMaxG][2];
#pragma unroll
  for (int gi = 0; gi < kMaxG; ++gi) {
    const int g = wave + kSW * gi;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      x[gi][h] = g < NG ? *(const f32x4*)(p.x + (size_t)row * C + 32 * g + 16 * h + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (p.w2 != nullptr) {                                 // second-stage weight [L2][L] -> LDS, rows padded to 33 floats
    for (int t = threadIdx.x; t < p.L2 * L; t += 64 * kSW) {
      const int j = t / L, l = t - j * L;
      w2s[j * 33 + l] = p.w2[t];
    }
  }
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  if constexpr (MODE == 3) {
#pragma unroll
    for (int gi = 0; gi < kMaxG; ++gi)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) x[gi][h][e] = quick_gelu(x[gi][h][e]);
  }
  if (MODE == 0 && p.drop_thresh != 0u) {
#pragma unroll
    for (int gi = 0; gi < kMaxG; ++gi) {
      const int g = wave + kSW * gi;
      if (g < NG) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            x[gi][h][e] *= drop_scale(p.seed, (unsigned long long)row * C + 32 * g + 16 * h + 4 * kq + e, p.drop_thresh, p.inv_keep);
      }
    }
  }
  if (ln) {
    // two-pass statistics like torch's LayerNorm: mean, then the centred sum of squares (eps inside the square root)
    float s = 0.f;
#pragma unroll
    for (int gi = 0; gi < kMaxG; ++gi)
#pragma unroll
      for (int h = 0; h < 2; ++h) s += (x[gi][h][0] + x[gi][h][1]) + (x[gi][h][2] + x[gi][h][3]);     // groups beyond NG hold zeros
    s = kq_sum(s);
    if (kq == 0) red[wave][i] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < kSW; ++w) mean += red[w][i];
    mean /= (float)C;
    float q = 0.f;
#pragma unroll
    for (int gi = 0; gi < kMaxG; ++gi) {
      if (wave + kSW * gi < NG) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = x[gi][h][e] - mean; q += d * d; }
      }
    }
    q = kq_sum(q);
    __syncthreads();                                     // everyone has read the first-pass sums
    if (kq == 0) red[wave][i] = q;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < kSW; ++w) var += red[w][i];
    const float rstd = rsqrtf(var / (float)C + p.eps);
    if (wave == 0 && kq == 0 && row0 + i < p.M) {
      if (p.mean) p.mean[row] = mean;
      if (p.rstd) p.rstd[row] = rstd;
    }
#pragma unroll
    for (int gi = 0; gi < kMaxG; ++gi) {
      const int g = wave + kSW * gi;
      if (g < NG) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c = 32 * g + 16 * h + 4 * kq;
          const f32x4 g4 = *(const f32x4*)(p.ln_g + c), b4 = *(const f32x4*)(p.ln_b + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[gi][h][e] = (x[gi][h][e] - mean) * rstd * g4[e] + b4[e];
        }
      }
    }
  }
  // ---- projection: this wave's share of the sum over C
  const int ntl = (L + 15) >> 4;                         // latent tiles of 16
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int gi = 0; gi < kMaxG; ++gi) {
    const int g = wave + kSW * gi;
    if (g < NG) {                                        // wave-uniform
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = 32 * g + 16 * h + 4 * kq;
        const f32x4 w0 = load_w_down(p.w, p.w_layout, L, C, i, c);
        const f32x4 w1 = ntl > 1 ? load_w_down(p.w, p.w_layout, L, C, 16 + i, c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[0] = mfma4(x[gi][h][e], w0[e], acc[0]);
        if (ntl > 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[1] = mfma4(x[gi][h][e], w1[e], acc[1]);
        }
      }
    }
  }
  part[wave][0][lane] = acc[0];
  part[wave][1][lane] = acc[1];
  __syncthreads();
  // waves 0 (and 1 when L > 16) finish one latent tile each: D lane (n = lane & 15, token rows 4 kq + e)
  if (wave < ntl) {
    f32x4 t = part[0][wave][lane];
#pragma unroll
    for (int w = 1; w < kSW; ++w) t += part[w][wave][lane];
    const int n = 16 * wave + i;
    if (n < L) {
      const float b = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * kq + e, m = row0 + r;
        const float zz = t[e] + b;
        const float yy = p.act == 1 ? quick_gelu(zz) : zz;
        yrow[r][n] = yy;
        if (m < p.M) {
          if (p.z) p.z[(size_t)m * L + n] = zz;
          if (p.y) p.y[(size_t)m * L + n] = yy;
        }
      }
    }
  }
  if (p.w2 != nullptr) {                                 // y2[m][j] = sum_l y[m][l] W2[j][l]
    __syncthreads();
    for (int t = threadIdx.x; t < 16 * p.L2; t += 64 * kSW) {
      const int r = t / p.L2, j = t - r * p.L2, m = row0 + r;
      float a = 0.f;
      for (int l = 0; l < L; ++l) a = __builtin_fmaf(yrow[r][l], w2s[j * 33 + l], a);
      if (m < p.M) p.y2[(size_t)m * p.L2 + j] = a;
    }
  }
}

// ---- up projection ------------------------------------------------------------------------------------------------------------------
struct Up2Args {                                        // optional down-projection of the rows side_up has just written
  const float* w; const float* bias; float* z; float* y; int L, act;          // w [L][C]; z / y [M][L]
};

// LNB: LayerNorm-backward epilogue (out = base + LN'(v; ln_x, mean, rstd, gamma));  EXT: DVPT's scalar gate / input-GELU derivative
template <bool LNB, bool EXT>
__global__ __launch_bounds__(64 * kSW) void side_up_kernel(UpArgs p, int L, Up2Args q) {
  __shared__ float red[kSW][16][2];
  __shared__ f32x4 part[kSW][2][64];
  constexpr int kMaxP = 4;                              // 32-column pairs per wave: C <= 1024
  const int lane = lane_id(), wave = wave_id();
  const int a = lane & 15, kq = lane >> 4;              // A operand: weight row a of a column tile; B operand / D: token row a
  const int C = p.C, NP = C >> 5, K4 = L >> 2;          // K4 k-steps of 4 latents
  const int row0 = blockIdx.x * 16;
  const int row = min(row0 + a, p.M - 1);
  const bool rvalid = row0 + a < p.M;
  const float* __restrict__ base = p.accumulate ? p.out : p.res;
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  // the wide stream first: base (and the LayerNorm input for LNB) of this lane's 8-column pieces
  f32x4 bs[kMaxP][2], xs[LNB ? kMaxP : 1][2];
#pragma unroll
  for (int pi = 0; pi < kMaxP; ++pi) {
    const int pp = wave + kSW * pi;
    const int c = 32 * pp + 8 * kq;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bs[pi][j] = (pp < NP && base != nullptr) ? *(const f32x4*)(base + (size_t)row * C + c + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (LNB) xs[pi][j] = pp < NP ? *(const f32x4*)(p.ln_x + (size_t)row * C + c + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // B operand: lat[row][kq K4 + s]
  float lb[8];
  {
    const float* src = p.lat + (size_t)row * L;
    if (p.lat_override != nullptr) {
      const int sidx = row / p.T, t = row - sidx * p.T;
      if (t < p.P) src = p.lat_override + ((size_t)sidx * p.P + t) * L;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) lb[s] = s < K4 ? src[kq * K4 + s] : 0.f;
  }
  f32x4 acc[kMaxP][2];
#pragma unroll
  for (int pi = 0; pi < kMaxP; ++pi) {
    const int pp = wave + kSW * pi;
    acc[pi][0] = acc[pi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (pp < NP) {                                       // wave-uniform
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = 32 * pp + 8 * (a >> 2) + 4 * j + (a & 3);           // weight row fed to A-operand row a of column tile j
        float wa[8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
          wa[s] = s < K4 ? (p.w_layout == 0 ? p.w[(size_t)c * L + kq * K4 + s] : p.w[(size_t)(kq * K4 + s) * C + c]) : 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s)
          if (s < K4) acc[pi][j] = mfma4(wa[s], lb[s], acc[pi][j]);
      }
    }
  }
  // ---- epilogue: lane = token row a, columns 32 pp + 8 kq + (4 j + e)
  [[maybe_unused]] float s1 = 0.f, s2 = 0.f, mu = 0.f, rs = 0.f;
  if constexpr (LNB) { mu = p.ln_mean[row]; rs = p.ln_rstd[row]; }
#pragma unroll
  for (int pi = 0; pi < kMaxP; ++pi) {
    const int pp = wave + kSW * pi;
    if (pp < NP) {
      const int c = 32 * pp + 8 * kq;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 v = acc[pi][j];
        if constexpr (LNB) {
          const f32x4 g4 = *(const f32x4*)(p.ln_g + c + 4 * j);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dh = v[e] * g4[e], xh = (xs[pi][j][e] - mu) * rs;
            v[e] = dh;
            xs[pi][j][e] = xh;
            s1 += dh;
            s2 += dh * xh;
          }
        } else {
          if (p.bias != nullptr) v += *(const f32x4*)(p.bias + c + 4 * j);
          if constexpr (EXT) {
            if (p.alpha_ptr != nullptr) v *= p.alpha_ptr[0];
            if (p.gg_x != nullptr) {
              const f32x4 xg = *(const f32x4*)(p.gg_x + (size_t)row * C + c + 4 * j);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= quick_gelu_grad(xg[e]);
            }
          }
          if (p.drop_thresh != 0u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= drop_scale(p.seed, (unsigned long long)row * C + c + 4 * j + e, p.drop_thresh, p.inv_keep);
          }
          v += bs[pi][j];
          if (rvalid) *(f32x4*)(p.out + (size_t)row * C + c + 4 * j) = v;
        }
        acc[pi][j] = v;
      }
      if (!LNB && p.out16 != nullptr && rvalid) {
        bf16x8 h8;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h8[e] = (bf16)acc[pi][0][e]; h8[4 + e] = (bf16)acc[pi][1][e]; }
        *(bf16x8*)(p.out16 + (size_t)row * C + c) = h8;
      }
    }
  }
  if constexpr (LNB) {
    // row sums over all of C: the four kq lanes of a row, then the eight waves
    s1 = kq_sum(s1);
    s2 = kq_sum(s2);
    if (kq == 0) { red[wave][a][0] = s1; red[wave][a][1] = s2; }
    __syncthreads();
    s1 = s2 = 0.f;
#pragma unroll
    for (int w = 0; w < kSW; ++w) { s1 += red[w][a][0]; s2 += red[w][a][1]; }
    s1 /= (float)C;
    s2 /= (float)C;
#pragma unroll
    for (int pi = 0; pi < kMaxP; ++pi) {
      const int pp = wave + kSW * pi;
      if (pp < NP && rvalid) {
        const int c = 32 * pp + 8 * kq;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs * (acc[pi][j][e] - s1 - xs[pi][j][e] * s2) + bs[pi][j][e];
          *(f32x4*)(p.out + (size_t)row * C + c + 4 * j) = o;
        }
      }
    }
  }
  if constexpr (!LNB && !EXT) {
    if (q.w != nullptr) {
      // second projection of the rows just written: A operand = the output piece itself (token row a, k <-> column c + 4 j + e), B operand
      // W2[n][c + 4 j + e] with c = 32 pp + 8 kq
      const int ntl = (q.L + 15) >> 4;
      f32x4 d2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int pi = 0; pi < kMaxP; ++pi) {
        const int pp = wave + kSW * pi;
        if (pp < NP) {
          const int c = 32 * pp + 8 * kq;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f32x4 w0 = load_w_down(q.w, 0, q.L, C, a, c + 4 * j);
            const f32x4 w1 = ntl > 1 ? load_w_down(q.w, 0, q.L, C, 16 + a, c + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) d2[0] = mfma4(acc[pi][j][e], w0[e], d2[0]);
            if (ntl > 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) d2[1] = mfma4(acc[pi][j][e], w1[e], d2[1]);
            }
          }
        }
      }
      part[wave][0][lane] = d2[0];
      part[wave][1][lane] = d2[1];
      __syncthreads();
      if (wave < ntl) {
        f32x4 t = part[0][wave][lane];
#pragma unroll
        for (int w = 1; w < kSW; ++w) t += part[w][wave][lane];
        const int n = 16 * wave + a;
        if (n < q.L) {
          const float b = q.bias != nullptr ? q.bias[n] : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int m = row0 + 4 * kq + e;
            const float zz = t[e] + b;
            if (m < p.M) {
              if (q.z) q.z[(size_t)m * q.L + n] = zz;
              if (q.y) q.y[(size_t)m * q.L + n] = q.act == 1 ? quick_gelu(zz) : zz;
            }
          }
        }
      }
    }
  }
}

static bool side_shape_ok(int L, int C, int L2) {
  return C % 32 == 0 && C >= 64 && C <= 32 * kSW * kMaxG && L % 4 == 0 && L >= 4 && L <= 32 && L2 <= 64;
}

static bool side_enabled() {
  static const bool on = getenv("GAVIKO_HIP_SIDE") == nullptr || getenv("GAVIKO_HIP_SIDE")[0] != '0';   // A/B switch: 0 = the row-per-wave kernels
  return on;
}

// returns 1 when the call is not covered (the caller falls back to the row-per-wave / MFMA-tile kernels)
int launch_side_down(const DownArgs& a, int L, hipStream_t s) {
  if (!side_enabled() || !side_shape_ok(L, a.C, a.w2 ? a.L2 : 0) || (a.mode != 0 && a.mode != 3)) return 1;
  const dim3 grid((a.M + 15) / 16), block(64 * kSW);
  if (a.mode == 3) GVK_LAUNCH(side_down_kernel<3>, grid, block, 0, s, a, L);
  else GVK_LAUNCH(side_down_kernel<0>, grid, block, 0, s, a, L);
  return check_launch("side_down");
}

int launch_side_up(const UpArgs& a, int L, const float* w2, const float* bias2, float* z2, float* y2, int L2, int act2, hipStream_t s) {
  if (!side_enabled() || !side_shape_ok(L, a.C, 0)) return 1;
  const bool lnb = a.ln_x != nullptr, ext = a.alpha_ptr != nullptr || a.gg_x != nullptr;
  if (w2 != nullptr && (lnb || ext || L2 % 4 != 0 || L2 > 32)) return set_error(-2, "side_up: the fused second projection takes the plain epilogue and L2 in 4..32");
  Up2Args q{w2, bias2, z2, y2, L2, act2};
  const dim3 grid((a.M + 15) / 16), block(64 * kSW);
  if (lnb) GVK_LAUNCH((side_up_kernel<true, false>), grid, block, 0, s, a, L, q);
  else if (ext) GVK_LAUNCH((side_up_kernel<false, true>), grid, block, 0, s, a, L, q);
  else GVK_LAUNCH((side_up_kernel<false, false>), grid, block, 0, s, a, L, q);
  return check_launch("side_up");
}

}  // namespace gvk
