// Row-per-wave forms of the rank-L "skinny" projections (the hot shapes: C = 768 / 1024, L = 20):
//   row_down   y[m][0:L] = act( LN?(drop?(x[m][:])) . W^T + b )   (+ y2 = y . W2^T)         gaviko.py:231-232, 155-156 and dgrads
//   row_up     out[m][:] = base[m][:] + drop?( lat[m][0:L] . W + b )   or   base + LN'(lat . W)   gaviko.py:242, 187 and dgrads
// Both are single passes over a [M][C] fp32 stream with 2*L flop per element -- HBM-bound, so they are laid out like the
// LayerNorm kernels (one 64-lane wave per token row, the row in registers as float4s: lane owns columns k*256 + 4*lane .. +3)
// instead of as small GEMMs: 16 rows per 512-thread workgroup (one workgroup per CU, two waves per SIMD), so the weight is
// staged once per CU.  The weight sits in LDS as Ws[l][c] (L*C*4 bytes, 60 KiB at L=20, C=768); a lane reads the
// float4 of its own columns for each l (conflict-free ds_read_b128), shared between the wave's two rows.
//   down: 4*L FMAs per float4, then L cross-lane sums per row: one DPP add, a 32-entry LDS line per l, L lanes finish.
//   up:   lat[m][l] is broadcast from lane l with v_readlane (an SGPR operand of the FMA); no reduction at all; the
//         LayerNorm-backward epilogue's two row sums are plain wave reductions in this layout.
// fp32 VALU throughout (these feed trainable parameters); same arguments, masks and results as skinny.hip's MFMA kernels,
// which remain for the shapes this layout does not cover (C < 128, L*C too large for LDS).
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

constexpr int kRS = 36;                // floats per reduction line (32 used; 16-byte aligned rows)
constexpr int kKC = 4;                 // C <= 1024: up to four float4 per lane

// Ws[l][c] <- weight; src_lc: source is [L][C] (straight copy), else [C][L] (transposed on the way in: consecutive lanes take
// consecutive c, so the LDS stores are conflict-free and the 16-byte global reads stay inside L2-resident lines).
template <int L>
__device__ __forceinline__ void stage_weight(float* Ws, const float* __restrict__ w, int C, bool src_lc) {
  if (src_lc) {
    const int n4 = L * C / 4;
    for (int i = threadIdx.x; i < n4; i += 64 * kNW) *(f32x4*)(Ws + 4 * i) = *(const f32x4*)(w + 4 * i);
  } else {
    constexpr int L4 = L / 4;
    for (int i = threadIdx.x; i < C * L4; i += 64 * kNW) {
      const int q = i / C, c = i - q * C;
      const f32x4 v = *(const f32x4*)(w + (size_t)c * L + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) Ws[(4 * q + e) * C + c] = v[e];
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
__device__ __forceinline__ float dpp_xor1(float v) {     // lane ^ 1 within a quad: one DPP move, no LDS crossbar
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}

template <int L, int R>
__device__ __forceinline__ void row_down_pass(const DownArgs& p, const float* Ws, float* red, const float (&w2r)[L], float bias_l,
                                              int first, int n, bool stage, int lane) {
  const int C = p.C;
  const bool ln = p.ln_g != nullptr;
  int rows[R];
#pragma unroll
  for (int r = 0; r < R; ++r) rows[r] = min(first + min(r, max(n - 1, 0)), p.M - 1);
  f32x4 v[R][kKC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int k = 0; k < kKC; ++k) {
      const int c = k * 256 + lane * 4;
      v[r][k] = (c < C) ? *(const f32x4*)(p.x + (size_t)rows[r] * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  if (stage) stage_weight<L>(const_cast<float*>(Ws), p.w, C, p.w_layout == 0);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (p.drop_thresh != 0u) {
#pragma unroll
      for (int k = 0; k < kKC; ++k) {
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
      for (int k = 0; k < kKC; ++k) s += v[r][k][0] + v[r][k][1] + v[r][k][2] + v[r][k][3];
      const float mean = wave_sum(s) / (float)C;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < kKC; ++k) {
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
      for (int k = 0; k < kKC; ++k) {
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
  if (stage) __syncthreads();                            // Ws complete (workgroup-uniform branch)

  // two-wide accumulators: the compiler packs them into v_pk_fma_f32 (half the VALU issue of scalar FMAs)
  f32x2 a[R][L];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int l = 0; l < L; ++l) a[r][l] = f32x2{0.f, 0.f};
#pragma unroll
  for (int k = 0; k < kKC; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const f32x4 w = *(const f32x4*)(Ws + l * C + c);
        const f32x2 wlo = {w[0], w[1]}, whi = {w[2], w[3]};
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const f32x2 xlo = {v[r][k][0], v[r][k][1]}, xhi = {v[r][k][2], v[r][k][3]};
          a[r][l] = __builtin_elementwise_fma(xlo, wlo, a[r][l]);          // explicit: the library builds with -ffp-contract=off
          a[r][l] = __builtin_elementwise_fma(xhi, whi, a[r][l]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int l = 0; l < L; ++l) {
      float pr = a[r][l][0] + a[r][l][1];
      pr += dpp_xor1(pr);
      if (!(lane & 1)) red[l * kRS + (lane >> 1)] = pr;
    }
    float yv = 0.f;
    if (lane < L) {                                      // a wave's own LDS traffic is in order: no barrier
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i) t += *(const f32x4*)(red + lane * kRS + 4 * i);
      const float zz = (t[0] + t[1]) + (t[2] + t[3]) + bias_l;
      yv = p.act == 1 ? quick_gelu(zz) : zz;
      if (r < n) {
        if (p.z) p.z[(size_t)rows[r] * L + lane] = zz;
        if (p.y) p.y[(size_t)rows[r] * L + lane] = yv;
      }
    }
    if (p.w2 != nullptr) {
      float acc = 0.f;
#pragma unroll
      for (int l = 0; l < L; ++l) acc = __builtin_fmaf(__shfl(yv, l, 64), w2r[l], acc);
      if (lane < p.L2 && r < n) p.y2[(size_t)rows[r] * p.L2 + lane] = acc;
    }
  }
}

template <int L>
__global__ __launch_bounds__(64 * kNW) void row_down_kernel(DownArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  const int lane = lane_id(), wave = wave_id();
  const float* Ws = lsm;
  float* red = lsm + L * p.C + wave * L * kRS;
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  const float bias_l = (p.bias != nullptr && lane < L) ? p.bias[lane] : 0.f;
  float w2r[L];                                          // second stage: lane t < L2 owns output t
#pragma unroll
  for (int l = 0; l < L; ++l) w2r[l] = (p.w2 != nullptr && lane < p.L2) ? p.w2[lane * L + l] : 0.f;
  int cursor = (int)((long long)blockIdx.x * p.M / gridDim.x);
  const int r1 = (int)((long long)(blockIdx.x + 1) * p.M / gridDim.x);
  bool stage = true;
  while (cursor < r1) {                                  // workgroup-uniform loop
    const Pass ps = next_pass(cursor, r1, wave);
    if (ps.R == 3) row_down_pass<L, 3>(p, Ws, red, w2r, bias_l, ps.first, ps.n, stage, lane);
    else row_down_pass<L, 2>(p, Ws, red, w2r, bias_l, ps.first, ps.n, stage, lane);
    stage = false;
  }
}

template <int L, int R>
__device__ __forceinline__ void row_up_pass(const UpArgs& p, const float* Ws, float* latrow, int first, int n, bool stage, int lane) {
  const int C = p.C;
  const float* base = p.accumulate ? p.out : p.res;
  const bool lnb = p.ln_x != nullptr;
  int rows[R];
#pragma unroll
  for (int r = 0; r < R; ++r) rows[r] = min(first + min(r, max(n - 1, 0)), p.M - 1);
  f32x4 bs[R][kKC], xs[R][kKC];
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int k = 0; k < kKC; ++k) {
      const int c = k * 256 + lane * 4;
      const bool ok = c < C;
      bs[r][k] = (ok && base != nullptr) ? *(const f32x4*)(base + (size_t)rows[r] * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      xs[r][k] = (ok && lnb) ? *(const f32x4*)(p.ln_x + (size_t)rows[r] * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (lane < L) {
      const float* src = p.lat + (size_t)rows[r] * L;
      if (p.lat_override != nullptr) {
        const int sidx = rows[r] / p.T, t = rows[r] - sidx * p.T;
        if (t < p.P) src = p.lat_override + ((size_t)sidx * p.P + t) * L;
      }
      latrow[r * L + lane] = src[lane];
    }
  }
  if (stage) {
    stage_weight<L>(const_cast<float*>(Ws), p.w, C, p.w_layout == 1);
    __syncthreads();
  }
  f32x4 acc[R][kKC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int k = 0; k < kKC; ++k) acc[r][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < L; ++l) {
    float sc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) sc[r] = latrow[r * L + l];             // wave-uniform LDS broadcast reads (own writes: in order)
#pragma unroll
    for (int k = 0; k < kKC; ++k) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
        const f32x4 w = *(const f32x4*)(Ws + l * C + c);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const f32x2 s2 = {sc[r], sc[r]};
          const f32x2 lo = __builtin_elementwise_fma(s2, f32x2{w[0], w[1]}, f32x2{acc[r][k][0], acc[r][k][1]});
          const f32x2 hi = __builtin_elementwise_fma(s2, f32x2{w[2], w[3]}, f32x2{acc[r][k][2], acc[r][k][3]});
          acc[r][k] = f32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = rows[r];
    if (!lnb) {
      // ---- plain epilogue: out = base + drop(v + bias)
      if (r < n) {
#pragma unroll
        for (int k = 0; k < kKC; ++k) {
          const int c = k * 256 + lane * 4;
          if (c < C) {
            f32x4 vv = acc[r][k];
            if (p.bias) vv += *(const f32x4*)(p.bias + c);
            if (p.drop_thresh != 0u) {
#pragma unroll
              for (int e = 0; e < 4; ++e) vv[e] *= drop_scale(p.seed, (unsigned long long)row * C + c + e, p.drop_thresh, p.inv_keep);
            }
            vv += bs[r][k];
            *(f32x4*)(p.out + (size_t)row * C + c) = vv;
            if (p.out16 != nullptr) {
              bf16x4 h = {(bf16)vv[0], (bf16)vv[1], (bf16)vv[2], (bf16)vv[3]};
              *(bf16x4*)(p.out16 + (size_t)row * C + c) = h;
            }
          }
        }
      }
    } else {
      // ---- LayerNorm-backward epilogue: dx = base + rstd * (g*v - mean(g*v) - xhat * mean(g*v*xhat))
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
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (acc[r][k][e] - s1 - xs[r][k][e] * s2) + bs[r][k][e];
            *(f32x4*)(p.out + (size_t)row * C + c) = o;
          }
        }
      }
    }
  }
}

template <int L>
__global__ __launch_bounds__(64 * kNW) void row_up_kernel(UpArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  __shared__ float latrow[kNW][3 * L];
  const int lane = lane_id(), wave = wave_id();
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  int cursor = (int)((long long)blockIdx.x * p.M / gridDim.x);
  const int r1 = (int)((long long)(blockIdx.x + 1) * p.M / gridDim.x);
  bool stage = true;
  while (cursor < r1) {                                  // workgroup-uniform loop
    const Pass ps = next_pass(cursor, r1, wave);
    if (ps.R == 3) row_up_pass<L, 3>(p, lsm, latrow[wave], ps.first, ps.n, stage, lane);
    else row_up_pass<L, 2>(p, lsm, latrow[wave], ps.first, ps.n, stage, lane);
    stage = false;
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
  const size_t lds = (size_t)(L * a.C + kNW * L * kRS) * sizeof(float);
  if (int rc = ensure_lds(&row_down_kernel<L>, lds, granted, "row_down")) return rc;
  GVK_LAUNCH((row_down_kernel<L>), dim3(row_grid(a.M)), dim3(64 * kNW), (unsigned)lds, s, a);
  return check_launch("skinny_down(row)");
}

template <int L>
static int launch_up_t(const UpArgs& a, hipStream_t s) {
  static size_t granted = 0;
  const size_t lds = (size_t)(L * a.C) * sizeof(float);
  if (int rc = ensure_lds(&row_up_kernel<L>, lds, granted, "row_up")) return rc;
  GVK_LAUNCH((row_up_kernel<L>), dim3(row_grid(a.M)), dim3(64 * kNW), (unsigned)lds, s, a);
  return check_launch("skinny_up(row)");
}

static bool row_shape_ok(int L, int C, int L2) {
  return C >= 128 && C % 4 == 0 && C <= 256 * kKC && L % 4 == 0 && L2 <= 64 && (size_t)(L * C + kNW * L * kRS) * sizeof(float) <= 150 * 1024;
}

int launch_row_down(const DownArgs& a, int L, hipStream_t s) {
  if (!row_shape_ok(L, a.C, a.w2 ? a.L2 : 0)) return 1;
  switch (L) {
    case 4: return launch_down_t<4>(a, s);
    case 8: return launch_down_t<8>(a, s);
    case 16: return launch_down_t<16>(a, s);
    case 20: return launch_down_t<20>(a, s);
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
    default: return 1;
  }
}

}  // namespace gvk
