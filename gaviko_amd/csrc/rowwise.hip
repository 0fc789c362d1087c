// Row-per-wave forms of the rank-L "skinny" projections (the hot shapes: C = 768 / 1024, L = 20):
//   row_down   y[m][0:L] = act( LN?(drop?(x[m][:])) . W^T + b )   (+ y2 = y . W2^T)         gaviko.py:231-232, 155-156 and dgrads
//   row_up     out[m][:] = base[m][:] + drop?( lat[m][0:L] . W + b )   or   base + LN'(lat . W)   gaviko.py:242, 187 and dgrads
// Both are single passes over a [M][C] fp32 stream with 2*L flop per element -- HBM-bound, so they are laid out like the
// LayerNorm kernels (one 64-lane wave per token row, the row in registers as float4s: lane owns columns k*256 + 4*lane .. +3)
// instead of as small GEMMs.  At most one 512-thread workgroup per CU, rows split evenly over them.  The weight is staged
// through LDS one 256-column chunk at a time as Wc[l][0:256) (20 KB at L = 20, so the workgroup fits beside the backbone's
// GEMM / attention workgroups on a CU); a lane reads the float4 of its own columns for each l (conflict-free ds_read_b128),
// shared between the wave's rows.
//   down: 4*L FMAs per float4 as v_pk_fma_f32, then L cross-lane sums per row: four DPP adds give 16-lane row sums, a 4-entry
//         LDS line per l joins the four rows of the wave, L lanes finish (bias, activation, optional second projection).
//   up:   lat[m][l] is a wave-uniform LDS broadcast operand of the FMAs; no reduction at all; the LayerNorm-backward
//         epilogue's two row sums are plain wave reductions in this layout.
// fp32 VALU throughout (these feed trainable parameters); same arguments, masks and results as skinny.hip's MFMA kernels,
// which remain for the shapes this layout does not cover (C < 128, L = 32).
#include "common.hpp"
#include "skinny_args.hpp"

namespace gvk {

// In-kernel phase stamps for tools/probe/probe_rowwise.hip (compiled out of the library).
#ifdef GVK_STAMPS
__device__ long long g_stamps[4][16];
#define GVK_STAMP(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) g_stamps[blockIdx.x == 0 ? 0 : 1][i] = wall_clock64(); \
                          if (threadIdx.x == 64 * kNW - 1 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) g_stamps[blockIdx.x == 0 ? 2 : 3][i] = wall_clock64(); } while (0)
#else
#define GVK_STAMP(i) do { } while (0)
#endif

constexpr int kNW = 8;                 // waves per workgroup

constexpr int kKC = 4;                 // C <= 1024: up to four float4 per lane

// Wc[l][0:256) <- columns 256k .. 256k+255 of the weight.  src_lc: the source is [L][C] (straight copy), else [C][L]
// (transposed on the way in: consecutive lanes take consecutive c, so the LDS stores are conflict-free and the 16-byte global
// reads stay inside L2-resident lines).  Only ONE 256-column chunk of the weight is resident at a time (20 KB at L = 20):
// with the whole weight in LDS (60 KB) these workgroups could not share a CU with the backbone's GEMM / attention
// workgroups (2 x 64 KB) and waited for them to retire -- in the running step the kernels took twice their isolated time.
template <int L>
__device__ __forceinline__ void stage_chunk(float* Wc, const float* __restrict__ w, int C, int k, bool src_lc) {
  const int cols = min(256, C - 256 * k);
  if (src_lc) {
    for (int i = threadIdx.x; i < L * 64; i += 64 * kNW) {
      const int l = i >> 6, c = (i & 63) * 4;
      if (c < cols) *(f32x4*)(Wc + l * 256 + c) = *(const f32x4*)(w + (size_t)l * C + 256 * k + c);
    }
  } else {
    constexpr int L4 = L / 4;
    for (int i = threadIdx.x; i < 256 * L4; i += 64 * kNW) {
      const int q = i >> 8, c = i & 255;
      if (c < cols) {
        const f32x4 v = *(const f32x4*)(w + (size_t)(256 * k + c) * L + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) Wc[(4 * q + e) * 256 + c] = v[e];
      }
    }
  }
}

// ---- row distribution ---------------------------------------------------------------------------------------------------
// The grid is at most one workgroup per CU and the rows are split evenly over it (M = 4132 over 256 CUs: 16 or 17 rows each),
// because with fixed 16-row workgroups the 3 left-over workgroups of a 259-workgroup launch cost a whole second round.
// Inside a workgroup a pass gives every wave 2 rows; when 2*kNW < rows <= 3*kNW are left, the whole workgroup runs the 3-row
// form once (waves without a third row recompute their last one and discard it) instead of a second pass.
struct Pass { int R, first, n; };
__device__ __forceinline__ Pass next_pass(int& cursor, int r1, int wave) {
  const int rem = r1 - cursor;
  Pass ps;
  if (rem > 3 * kNW || rem <= 2 * kNW) {
    ps.R = 2; ps.first = cursor + 2 * wave; ps.n = max(0, min(2, rem - 2 * wave));
    cursor += min(rem, 2 * kNW);
  } else {
    const int k = rem - 2 * kNW;                         // waves < k own three rows
    ps.R = 3;
    if (wave < k) { ps.first = cursor + 3 * wave; ps.n = 3; }
    else { ps.first = cursor + 3 * k + 2 * (wave - k); ps.n = 2; }
    cursor = r1;
  }
  return ps;
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over each 16-lane row, left in all of its lanes: quad xor 1, quad xor 2, half-row mirror, row mirror -- four DPP adds,
// no LDS crossbar
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

template <int L, int R, int MODE, int KC>
__device__ __forceinline__ void row_down_pass(const DownArgs& p, float* Wc, float* red, const float* w2s, float bias_l,
                                              int first, int n, int lane) {
  const int C = p.C;
  const bool ln = MODE == 0 && p.ln_g != nullptr;
  int rows[R];
#pragma unroll
  for (int r = 0; r < R; ++r) rows[r] = min(first + min(r, max(n - 1, 0)), p.M - 1);
  f32x4 v[R][KC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const int c = k * 256 + lane * 4;
      v[r][k] = (c < C) ? *(const f32x4*)(p.x + (size_t)rows[r] * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  if constexpr (MODE == 3) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[r][k][e] = quick_gelu(v[r][k][e]);
  }
  if constexpr (MODE == 2) {
    // LayerNorm backward of the rows (as ln_bwd_kernel), leaving dx in v for the projection
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float mean = p.mean_in[rows[r]], rstd = p.rstd_in[rows[r]];
      f32x4 dh[KC];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int c = k * 256 + lane * 4;
        dh[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < C) {
          f32x4 dv;
          if (p.dy16 != nullptr) {
            const bf16x4 h = *(const bf16x4*)(p.dy16 + (size_t)rows[r] * C + c);
            dv = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
          } else {
            dv = *(const f32x4*)(p.dy + (size_t)rows[r] * C + c);
          }
          const f32x4 g = *(const f32x4*)(p.ln_g + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[r][k][e] = (v[r][k][e] - mean) * rstd;     // xhat
            dh[k][e] = dv[e] * g[e];
            s1 += dh[k][e];
            s2 += dh[k][e] * v[r][k][e];
          }
        }
      }
      const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int c = k * 256 + lane * 4;
        if (c < C) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rstd * (dh[k][e] - m1 - v[r][k][e] * m2);
          if (p.dres) o += *(const f32x4*)(p.dres + (size_t)rows[r] * C + c);
          v[r][k] = o;
          if (r < n) {
            *(f32x4*)(p.dx + (size_t)rows[r] * C + c) = o;
            if (p.dx16) {
              bf16x4 h = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
              *(bf16x4*)(p.dx16 + (size_t)rows[r] * C + c) = h;
            }
          }
        }
      }
    }
  }
  if constexpr (MODE == 1) {
    // LayerNorm forward of the rows (as ln_fwd_kernel: bf16 output, statistics saved); v stays raw for the projection
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < KC; ++k) s += v[r][k][0] + v[r][k][1] + v[r][k][2] + v[r][k][3];
      const float mean = wave_sum(s) / (float)C;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        if (k * 256 + lane * 4 < C) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = v[r][k][e] - mean;
            q += d * d;
          }
        }
      }
      const float rstd = rsqrtf(wave_sum(q) / (float)C + p.eps);
      if (r < n) {
        if (lane == 0) {
          if (p.mean) p.mean[rows[r]] = mean;
          if (p.rstd) p.rstd[rows[r]] = rstd;
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) {
          const int c = k * 256 + lane * 4;
          if (c < C) {
            const f32x4 g = *(const f32x4*)(p.ln_g + c);
            const f32x4 b = *(const f32x4*)(p.ln_b + c);
            bf16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (bf16)((v[r][k][e] - mean) * rstd * g[e] + b[e]);
            *(bf16x4*)(p.y16 + (size_t)rows[r] * C + c) = h;
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (MODE == 0 && p.drop_thresh != 0u) {
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int c = k * 256 + lane * 4;
        if (c < C) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[r][k][e] *= drop_scale(p.seed, (unsigned long long)rows[r] * C + c + e, p.drop_thresh, p.inv_keep);
        }
      }
    }
    if (ln) {                                            // two-pass statistics, as the LayerNorm kernels
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < KC; ++k) s += v[r][k][0] + v[r][k][1] + v[r][k][2] + v[r][k][3];
      const float mean = wave_sum(s) / (float)C;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int c = k * 256 + lane * 4;
        if (c < C) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = v[r][k][e] - mean;
            q += d * d;
          }
        }
      }
      const float rstd = rsqrtf(wave_sum(q) / (float)C + p.eps);
      if (lane == 0 && r < n) {
        if (p.mean) p.mean[rows[r]] = mean;
        if (p.rstd) p.rstd[rows[r]] = rstd;
      }
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int c = k * 256 + lane * 4;
        if (c < C) {
          const f32x4 g = *(const f32x4*)(p.ln_g + c);
          const f32x4 b = *(const f32x4*)(p.ln_b + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[r][k][e] = (v[r][k][e] - mean) * rstd * g[e] + b[e];
        }
      }
    }
  }
  // accumulators per (row, l); the FMAs are issued two l at a time as v_pk_fma_f32 (x broadcast into both halves)
  f32x2 a[R][L / 2];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int l = 0; l < L / 2; ++l) a[r][l] = f32x2{0.f, 0.f};
  const int nk = (C + 255) / 256;
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    if (k < nk) {                                        // workgroup-uniform
      if (k > 0) __syncthreads();                        // everyone is done with the previous chunk
      stage_chunk<L>(Wc, p.w, C, k, p.w_layout == 0);
      __syncthreads();
      if (k * 256 + lane * 4 < C) {
#pragma unroll
        for (int l = 0; l < L / 2; ++l) {
          const f32x4 w0 = *(const f32x4*)(Wc + (2 * l) * 256 + lane * 4);
          const f32x4 w1 = *(const f32x4*)(Wc + (2 * l + 1) * 256 + lane * 4);
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              a[r][l] = __builtin_elementwise_fma(f32x2{v[r][k][e], v[r][k][e]}, f32x2{w0[e], w1[e]}, a[r][l]);
        }
      }
    }
  }
  __syncthreads();                                       // the next pass (or nothing) may restage Wc
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float pr = row16_sum(a[r][l >> 1][l & 1]);
      if ((lane & 15) == 0) red[l * 4 + (lane >> 4)] = pr;
    }
    float yv = 0.f;
    if (lane < L) {                                      // a wave's own LDS traffic is in order: no barrier
      const f32x4 t = *(const f32x4*)(red + lane * 4);
      const float zz = (t[0] + t[1]) + (t[2] + t[3]) + bias_l;
      yv = p.act == 1 ? quick_gelu(zz) : zz;
      if (r < n) {
        if (p.z) p.z[(size_t)rows[r] * L + lane] = zz;
        if (p.y) p.y[(size_t)rows[r] * L + lane] = yv;
        if (p.ysplit != nullptr) {                       // elementwise.hip: pack_split_bf16_kernel, activation side
          bf16* d16 = p.ysplit + (size_t)rows[r] * p.ysplit_ld + p.ysplit_col;
          const bf16 hi = (bf16)yv;
          d16[lane] = hi; d16[L + lane] = (bf16)(yv - (float)hi); d16[2 * L + lane] = hi;
        }
      }
    }
    if (p.w2 != nullptr) {
      float acc = 0.f;
#pragma unroll
      for (int l = 0; l < L; ++l) acc = __builtin_fmaf(__shfl(yv, l, 64), w2s[lane * (L + 1) + l], acc);
      if (lane < p.L2 && r < n) p.y2[(size_t)rows[r] * p.L2 + lane] = acc;
    }
  }
}

template <int L, int MODE, int KC>
__global__ __launch_bounds__(64 * kNW, 3) void row_down_kernel(DownArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  const int lane = lane_id(), wave = wave_id();
  float* Wc = lsm;                                       // [L][256] weight chunk
  float* red = lsm + L * 256 + wave * L * 4;             // [L][4] row-of-16 partial sums per wave
  float* w2s = lsm + L * 256 + kNW * L * 4;              // [64][L+1] second-stage weight (rows >= L2 zero), padded rows
  if (p.w2 != nullptr) {
    for (int i = threadIdx.x; i < 64 * L; i += 64 * kNW) {
      const int t = i / L, l = i - t * L;
      w2s[t * (L + 1) + l] = t < p.L2 ? p.w2[t * L + l] : 0.f;
    }
  }                                                      // visible after the first staging barrier
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  const float bias_l = (p.bias != nullptr && lane < L) ? p.bias[lane] : 0.f;
  int cursor = (int)((long long)blockIdx.x * p.M / gridDim.x);
  const int r1 = (int)((long long)(blockIdx.x + 1) * p.M / gridDim.x);
  while (cursor < r1) {                                  // workgroup-uniform loop
    const Pass ps = next_pass(cursor, r1, wave);
    if (ps.R == 3) row_down_pass<L, 3, MODE, KC>(p, Wc, red, w2s, bias_l, ps.first, ps.n, lane);
    else row_down_pass<L, 2, MODE, KC>(p, Wc, red, w2s, bias_l, ps.first, ps.n, lane);
  }
}

template <int L, int R, bool LNB, bool EXT>
__device__ __forceinline__ void row_up_pass(const UpArgs& p, float* Wc, float* latrow, int first, int n, int lane) {
  const int C = p.C;
  const float* base = p.accumulate ? p.out : p.res;
  int rows[R];
#pragma unroll
  for (int r = 0; r < R; ++r) rows[r] = min(first + min(r, max(n - 1, 0)), p.M - 1);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (lane < L) {
      const float* src = p.lat + (size_t)rows[r] * L;
      if (p.lat_override != nullptr) {
        const int sidx = rows[r] / p.T, t = rows[r] - sidx * p.T;
        if (t < p.P) src = p.lat_override + ((size_t)sidx * p.P + t) * L;
      }
      latrow[r * L + lane] = src[lane];
    }
  }
  // LayerNorm-backward epilogue: the whole row of g*v and xhat is needed for the two row sums, so those stay in registers;
  // the plain epilogue streams: each 256-column chunk is finished (bias, dropout, + base, store) before the next is staged.
  f32x4 acc[LNB ? R : 1][LNB ? kKC : 1], xs[LNB ? R : 1][LNB ? kKC : 1];
  if constexpr (LNB) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < kKC; ++k) {
        const int c = k * 256 + lane * 4;
        xs[r][k] = (c < C) ? *(const f32x4*)(p.ln_x + (size_t)rows[r] * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
  const int nk = (C + 255) / 256;
#pragma unroll
  for (int k = 0; k < kKC; ++k) {
    if (k < nk) {                                        // workgroup-uniform
      const int c = k * 256 + lane * 4;
      f32x4 bs[R];
      if constexpr (!LNB) {                              // issue this chunk's stream loads before the staging barrier
#pragma unroll
        for (int r = 0; r < R; ++r)
          bs[r] = (c < C && base != nullptr) ? *(const f32x4*)(base + (size_t)rows[r] * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (k > 0) __syncthreads();                        // everyone is done with the previous chunk
      stage_chunk<L>(Wc, p.w, C, k, p.w_layout == 1);
      __syncthreads();
      f32x4 a[R];
#pragma unroll
      for (int r = 0; r < R; ++r) a[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < C) {
#pragma unroll 4
        for (int l = 0; l < L; ++l) {
          const f32x4 w = *(const f32x4*)(Wc + l * 256 + lane * 4);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float sc = latrow[r * L + l];          // wave-uniform LDS broadcast read (own writes: in order)
            const f32x2 s2 = {sc, sc};
            const f32x2 lo = __builtin_elementwise_fma(s2, f32x2{w[0], w[1]}, f32x2{a[r][0], a[r][1]});
            const f32x2 hi = __builtin_elementwise_fma(s2, f32x2{w[2], w[3]}, f32x2{a[r][2], a[r][3]});
            a[r] = f32x4{lo[0], lo[1], hi[0], hi[1]};
          }
        }
      }
      if constexpr (LNB) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][k] = a[r];
      } else if (c < C) {
        // ---- plain epilogue of this chunk: out = base + drop(v + bias)
        const f32x4 b4 = p.bias ? *(const f32x4*)(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r < n) {
            const int row = rows[r];
            f32x4 vv = a[r] + b4;
            if constexpr (EXT) {                         // DVPT: scalar gate and / or the input QuickGELU's derivative
              if (p.alpha_ptr != nullptr) vv *= p.alpha_ptr[0];
              if (p.gg_x != nullptr) {
                const f32x4 xg = *(const f32x4*)(p.gg_x + (size_t)row * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) vv[e] *= quick_gelu_grad(xg[e]);
              }
            }
            if (p.drop_thresh != 0u) {
#pragma unroll
              for (int e = 0; e < 4; ++e) vv[e] *= drop_scale(p.seed, (unsigned long long)row * C + c + e, p.drop_thresh, p.inv_keep);
            }
            vv += bs[r];
            *(f32x4*)(p.out + (size_t)row * C + c) = vv;
            if (p.out16 != nullptr) {
              bf16x4 h = {(bf16)vv[0], (bf16)vv[1], (bf16)vv[2], (bf16)vv[3]};
              *(bf16x4*)(p.out16 + (size_t)row * C + c) = h;
            }
          }
        }
      }
    }
  }
  __syncthreads();                                       // the next pass (or nothing) may restage Wc
  if constexpr (LNB) {
    // ---- LayerNorm-backward epilogue: dx = base + rstd * (g*v - mean(g*v) - xhat * mean(g*v*xhat))
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = rows[r];
      const float mu = p.ln_mean[row], rs = p.ln_rstd[row];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < kKC; ++k) {
        const int c = k * 256 + lane * 4;
        if (c < C) {
          const f32x4 g4 = *(const f32x4*)(p.ln_g + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dh = acc[r][k][e] * g4[e];
            const float xh = (xs[r][k][e] - mu) * rs;
            acc[r][k][e] = dh;
            xs[r][k][e] = xh;
            s1 += dh;
            s2 += dh * xh;
          }
        }
      }
      s1 = wave_sum(s1) / (float)C;
      s2 = wave_sum(s2) / (float)C;
      if (r < n) {
#pragma unroll
        for (int k = 0; k < kKC; ++k) {
          const int c = k * 256 + lane * 4;
          if (c < C) {
            const f32x4 b = base != nullptr ? *(const f32x4*)(base + (size_t)row * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (acc[r][k][e] - s1 - xs[r][k][e] * s2) + b[e];
            *(f32x4*)(p.out + (size_t)row * C + c) = o;
          }
        }
      }
    }
  }
}

template <int L, bool LNB, bool EXT>
__global__ __launch_bounds__(64 * kNW, LNB ? 3 : 4) void row_up_kernel(UpArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  __shared__ float latrow[kNW][3 * L];
  const int lane = lane_id(), wave = wave_id();
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  int cursor = (int)((long long)blockIdx.x * p.M / gridDim.x);
  const int r1 = (int)((long long)(blockIdx.x + 1) * p.M / gridDim.x);
  while (cursor < r1) {                                  // workgroup-uniform loop
    const Pass ps = next_pass(cursor, r1, wave);
    if (ps.R == 3) row_up_pass<L, 3, LNB, EXT>(p, lsm, latrow[wave], ps.first, ps.n, lane);
    else row_up_pass<L, 2, LNB, EXT>(p, lsm, latrow[wave], ps.first, ps.n, lane);
  }
}

static int row_grid(int M) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) cus = 256;
    else cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int want = (M + 2 * kNW - 1) / (2 * kNW);
  return want < cus ? want : cus;
}

template <typename K>
static int ensure_lds(K kernel, size_t bytes, size_t& granted, const char* who) {
  if (bytes <= granted) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(%s): %s", who, hipGetErrorString(e));
  granted = bytes;
  return 0;
}

template <int L>
static int launch_down_t(const DownArgs& a, hipStream_t s) {
  static size_t granted = 0;
  const size_t lds = (size_t)(L * 256 + kNW * L * 4 + 64 * (L + 1)) * sizeof(float);
  (void)granted;                                         // < 64 KB: no attribute needed
  const dim3 grid(row_grid(a.M)), block(64 * kNW);
  if (a.C <= 768) {                                       // three float4 chunks per lane: 25 % fewer row registers than the C <= 1024 form
    if (a.mode == 3) GVK_LAUNCH((row_down_kernel<L, 3, 3>), grid, block, (unsigned)lds, s, a);
    else if (a.mode == 1) GVK_LAUNCH((row_down_kernel<L, 1, 3>), grid, block, (unsigned)lds, s, a);
    else if (a.mode == 2) GVK_LAUNCH((row_down_kernel<L, 2, 3>), grid, block, (unsigned)lds, s, a);
    else GVK_LAUNCH((row_down_kernel<L, 0, 3>), grid, block, (unsigned)lds, s, a);
  } else {
    if (a.mode == 3) GVK_LAUNCH((row_down_kernel<L, 3, 4>), grid, block, (unsigned)lds, s, a);
    else if (a.mode == 1) GVK_LAUNCH((row_down_kernel<L, 1, 4>), grid, block, (unsigned)lds, s, a);
    else if (a.mode == 2) GVK_LAUNCH((row_down_kernel<L, 2, 4>), grid, block, (unsigned)lds, s, a);
    else GVK_LAUNCH((row_down_kernel<L, 0, 4>), grid, block, (unsigned)lds, s, a);
  }
  return check_launch("skinny_down(row)");
}

template <int L>
static int launch_up_t(const UpArgs& a, hipStream_t s) {
  const unsigned lds = (unsigned)(L * 256 * sizeof(float));       // < 64 KB: no attribute needed
  if (a.ln_x != nullptr) GVK_LAUNCH((row_up_kernel<L, true, false>), dim3(row_grid(a.M)), dim3(64 * kNW), lds, s, a);
  else if (a.alpha_ptr != nullptr || a.gg_x != nullptr) GVK_LAUNCH((row_up_kernel<L, false, true>), dim3(row_grid(a.M)), dim3(64 * kNW), lds, s, a);
  else GVK_LAUNCH((row_up_kernel<L, false, false>), dim3(row_grid(a.M)), dim3(64 * kNW), lds, s, a);
  return check_launch("skinny_up(row)");
}

static bool row_shape_ok(int L, int C, int L2) {
  return C >= 128 && C % 4 == 0 && C <= 256 * kKC && L % 4 == 0 && L2 <= 64 ;
}

int launch_row_down(const DownArgs& a, int L, hipStream_t s) {
  if (!row_shape_ok(L, a.C, a.w2 ? a.L2 : 0)) return 1;
  switch (L) {
    case 4: return launch_down_t<4>(a, s);
    case 8: return launch_down_t<8>(a, s);
    case 16: return launch_down_t<16>(a, s);
    case 20: return launch_down_t<20>(a, s);
    case 24: return launch_down_t<24>(a, s);                 // EVP at ViT-B: dim / 32
    default: return 1;                                   // L = 32 would spill (3 rows x 32 two-wide accumulators): MFMA-tile kernel
  }
}

int launch_row_up(const UpArgs& a, int L, hipStream_t s) {
  if (!row_shape_ok(L, a.C, 0)) return 1;
  switch (L) {
    case 4: return launch_up_t<4>(a, s);
    case 8: return launch_up_t<8>(a, s);
    case 16: return launch_up_t<16>(a, s);
    case 20: return launch_up_t<20>(a, s);
    case 24: return launch_up_t<24>(a, s);
    default: return 1;
  }
}

}  // namespace gvk
