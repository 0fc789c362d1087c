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
//   * all of a tile's loads (48 KiB of x per workgroup, the weight fragments, the side operands) are in flight together: the kernels are
//     templated on everything that shapes a loop (latent width L = 20, 32-column groups per wave, weight layout), loads are clamped rather
//     than branched around, so the compiler emits one block of loads followed by one block of MFMAs.
// k-order inside an MFMA is free (the sum runs over all of it), so an operand register is whatever 16-byte piece loads best:
//   down: lane (i = lane & 15, kq = lane >> 4) holds x[row i][c .. c+3], c = 32 g + 16 h + 4 kq -- A operand, element e feeds MFMA e;
//         the B operand is W[n = lane & 15 (+16)][same c .. c+3].
//   up:   D = W^T-fragment (A) x lat^T (B): lane (a = lane & 15, kq) holds W[c(a, j)][kq L/4 + s], the B operand lat[row lane & 15][kq L/4 + s];
//         the weight rows of column tiles 2p, 2p+1 are interleaved as in gemm_epilogue.hpp, so a lane ends up with EIGHT consecutive output
//         columns of one token row (32-byte pieces, 128 contiguous bytes per row and instruction pair).
#include "skinny_args.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

// Streaming accesses of the [M][C] tensors (read or written once per launch): plain loads and stores.  Marking them non-temporal, to keep
// the 12 MB a pass moves from evicting the backbone kernels' operands from the per-XCD L2, was measured worse (660 vs 681 volumes/s, round
// 2: the next side kernel re-reads the same rows from L2 / the MALL a few tens of microseconds later); that variant is not in the source.
__device__ __forceinline__ f32x4 ld_stream(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ void st_stream(float* p, const f32x4 v) { *(f32x4*)p = v; }


namespace {
constexpr int kSW = 8;                 // waves per workgroup
constexpr int kSL = 20;                // latent width these kernels are built for (configs/gaviko.yaml prompt_latent_dim / local_dim)
constexpr int kK4 = kSL / 4;           // MFMA k-steps of the up projection

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// sum over the four kq lanes that share lane & 15
__device__ __forceinline__ float kq_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// W fragment of a down projection: W[n][c .. c+3], zero for n >= kSL (the load itself is clamped, never skipped).  WL 0: w [L][C]; 1: w [C][L]
template <int WL>
__device__ __forceinline__ f32x4 load_w_down(const float* __restrict__ w, int C, int n, int c) {
  const int nn = n < kSL ? n : 0;
  f32x4 v;
  if constexpr (WL == 0) v = *(const f32x4*)(w + (size_t)nn * C + c);
  else v = f32x4{w[(size_t)c * kSL + nn], w[(size_t)(c + 1) * kSL + nn], w[(size_t)(c + 2) * kSL + nn], w[(size_t)(c + 3) * kSL + nn]};
  return n < kSL ? v : zero4();
}
}  // namespace

// NGW = 32-column groups per wave (C = 256 NGW, or fewer: waves past the end work on zeros), WL = weight layout.
// MODE 0: y = act(f(x) . W^T + b), f = identity | LayerNorm(ln_g, ln_b) | dropout mask                                  (gvk_skinny_down)
// MODE 1: y16 = bf16 LayerNorm(x) with mean / rstd saved, and the projection of the RAW row                            (gvk_layernorm_fwd_proj)
// MODE 2: dx = dres + LayerNorm'(dy; x, mean_in, rstd_in, ln_g) (+ bf16 copy dx16), and the projection of dx             (gvk_layernorm_bwd_proj)
template <int NGW, int WL, int MODE>
__global__ __launch_bounds__(64 * kSW, NGW <= 3 ? 4 : 2) void side_down_kernel(DownArgs p) {     // <= 128 VGPRs: two workgroups per CU (259 tiles on 256 CUs)
  __shared__ float red[kSW][16];
  __shared__ float red2[kSW][16];
  __shared__ f32x4 part[kSW][2][64];
  __shared__ float yrow[16][33];
  __shared__ float w2s[64 * 33];
  const int lane = lane_id(), wave = wave_id();
  const int i = lane & 15, kq = lane >> 4;
  const int C = p.C, NG = C >> 5;                       // groups of 32 columns; wave w owns groups w, w + 8, ...
  const int row0 = (blockIdx.x + p.tile_off) * 16;
  const int row = min(row0 + i, p.M - 1);
  const bool ln = MODE == 0 && p.ln_g != nullptr;
  // ---- every load of the tile goes out first: the rows, the weight fragments, the LayerNorm affine
  f32x4 x[NGW][2], wf[NGW][2][2];
  [[maybe_unused]] f32x4 dyv[MODE == 2 ? NGW : 1][2], drs[MODE == 2 ? NGW : 1][2];
  int col[NGW][2];
  bool ok[NGW];
#pragma unroll
  for (int gi = 0; gi < NGW; ++gi) {
    const int g = wave + kSW * gi;
    ok[gi] = g < NG;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      col[gi][h] = 32 * (ok[gi] ? g : 0) + 16 * h + 4 * kq;
      x[gi][h] = ld_stream(p.x + (size_t)row * C + col[gi][h]);
      if constexpr (MODE == 2) {
        dyv[gi][h] = ld_stream(p.dy + (size_t)row * C + col[gi][h]);
        drs[gi][h] = p.dres != nullptr ? ld_stream(p.dres + (size_t)row * C + col[gi][h]) : zero4();
      }
    }
  }
  if constexpr (MODE == 0) {                             // the LayerNorm modes hold two or three wide streams: their weight fragments are fetched
#pragma unroll                                           // where the projection consumes them (128 VGPRs = two workgroups per CU, or the 259 tiles
    for (int gi = 0; gi < NGW; ++gi)                     //  of M = 4132 need a second round)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        wf[gi][h][0] = load_w_down<WL>(p.w, C, i, col[gi][h]);
        wf[gi][h][1] = load_w_down<WL>(p.w, C, 16 + i, col[gi][h]);
      }
  }
  if (p.w2 != nullptr) {                                 // second-stage weight [L2][L] -> LDS, rows padded to 33 floats
    for (int t = threadIdx.x; t < p.L2 * kSL; t += 64 * kSW) {
      const int j = t / kSL, l = t - j * kSL;
      w2s[j * 33 + l] = p.w2[t];
    }
  }
  if (p.drop_thresh != 0u) {
    if (p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          x[gi][h][e] *= drop_scale(p.seed, (unsigned long long)row * C + col[gi][h] + e, p.drop_thresh, p.inv_keep);
  }
  if (ln) {
    // two-pass statistics like torch's LayerNorm: mean, then the centred sum of squares (eps inside the square root)
    float s = 0.f;
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
      for (int h = 0; h < 2; ++h) s += ok[gi] ? (x[gi][h][0] + x[gi][h][1]) + (x[gi][h][2] + x[gi][h][3]) : 0.f;
    s = kq_sum(s);
    if (kq == 0) red[wave][i] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < kSW; ++w) mean += red[w][i];
    mean /= (float)C;
    float q = 0.f;
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = x[gi][h][e] - mean; q += ok[gi] ? d * d : 0.f; }
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
    for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
      for (int h = 0; h < 2; ++h) {                       // the affine vectors are cache-resident: fetched here, not held across the statistics
        const f32x4 g4 = *(const f32x4*)(p.ln_g + col[gi][h]), b4 = *(const f32x4*)(p.ln_b + col[gi][h]);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[gi][h][e] = (x[gi][h][e] - mean) * rstd * g4[e] + b4[e];
      }
  }
  if constexpr (MODE == 1) {
    // LayerNorm forward of the rows (two-pass statistics), bf16 output; x stays raw for the projection
    float s = 0.f;
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
      for (int h = 0; h < 2; ++h) s += ok[gi] ? (x[gi][h][0] + x[gi][h][1]) + (x[gi][h][2] + x[gi][h][3]) : 0.f;
    s = kq_sum(s);
    if (kq == 0) red[wave][i] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < kSW; ++w) mean += red[w][i];
    mean /= (float)C;
    float q = 0.f;
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = x[gi][h][e] - mean; q += ok[gi] ? d * d : 0.f; }
    q = kq_sum(q);
    __syncthreads();
    if (kq == 0) red[wave][i] = q;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < kSW; ++w) var += red[w][i];
    const float rstd = rsqrtf(var / (float)C + p.eps);
    const bool rv = row0 + i < p.M;
    if (wave == 0 && kq == 0 && rv) {
      if (p.mean) p.mean[row] = mean;
      if (p.rstd) p.rstd[row] = rstd;
    }
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 g4 = *(const f32x4*)(p.ln_g + col[gi][h]), b4 = *(const f32x4*)(p.ln_b + col[gi][h]);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)((x[gi][h][e] - mean) * rstd * g4[e] + b4[e]);
        if (rv && ok[gi]) *(bf16x4*)(p.y16 + (size_t)row * C + col[gi][h]) = o;
      }
  }
  if constexpr (MODE == 2) {
    // LayerNorm backward of the rows (as ln_bwd_kernel); dx replaces x and feeds the projection
    const float mean = p.mean_in[row], rstd = p.rstd_in[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 g4 = *(const f32x4*)(p.ln_g + col[gi][h]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (x[gi][h][e] - mean) * rstd, dh = dyv[gi][h][e] * g4[e];
          x[gi][h][e] = xh;
          dyv[gi][h][e] = dh;
          s1 += ok[gi] ? dh : 0.f;
          s2 += ok[gi] ? dh * xh : 0.f;
        }
      }
    s1 = kq_sum(s1);
    s2 = kq_sum(s2);
    if (kq == 0) { red[wave][i] = s1; red2[wave][i] = s2; }
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int w = 0; w < kSW; ++w) { m1 += red[w][i]; m2 += red2[w][i]; }
    m1 /= (float)C;
    m2 /= (float)C;
    const bool rv = row0 + i < p.M;
#pragma unroll
    for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (dyv[gi][h][e] - m1 - x[gi][h][e] * m2) + drs[gi][h][e];
        x[gi][h] = o;
        if (rv && ok[gi]) {
          st_stream(p.dx + (size_t)row * C + col[gi][h], o);
          if (p.dx16 != nullptr) {
            const bf16x4 hh = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
            *(bf16x4*)(p.dx16 + (size_t)row * C + col[gi][h]) = hh;
          }
        }
      }
  }
  // ---- projection: this wave's share of the sum over C (groups past the end contribute zeros)
  f32x4 acc[2] = {zero4(), zero4()};
#pragma unroll
  for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 xv = ok[gi] ? x[gi][h] : zero4();
      if constexpr (MODE != 0) {
        wf[gi][h][0] = load_w_down<WL>(p.w, C, i, col[gi][h]);
        wf[gi][h][1] = load_w_down<WL>(p.w, C, 16 + i, col[gi][h]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[0] = mfma4(xv[e], wf[gi][h][0][e], acc[0]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[1] = mfma4(xv[e], wf[gi][h][1][e], acc[1]);
    }
  part[wave][0][lane] = acc[0];
  part[wave][1][lane] = acc[1];
  __syncthreads();
  // waves 0 and 1 finish one latent tile each: D lane (n = lane & 15, token rows 4 kq + e)
  if (wave < 2) {
    f32x4 t = part[0][wave][lane];
#pragma unroll
    for (int w = 1; w < kSW; ++w) t += part[w][wave][lane];
    const int n = 16 * wave + i;
    if (n < kSL) {
      const float b = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * kq + e, m = row0 + r;
        const float zz = t[e] + b;
        const float yy = p.act == 1 ? quick_gelu(zz) : zz;
        yrow[r][n] = yy;
        if (m < p.M) {
          if (p.z) p.z[(size_t)m * kSL + n] = zz;
          if (p.y) p.y[(size_t)m * kSL + n] = yy;
          if (p.ysplit != nullptr) {                     // split-bf16 copy [hi | lo | hi] into spare K columns of a GEMM operand (elementwise.hip)
            bf16* d16 = p.ysplit + (size_t)m * p.ysplit_ld + p.ysplit_col;
            const bf16 hi = (bf16)yy;
            d16[n] = hi; d16[kSL + n] = (bf16)(yy - (float)hi); d16[2 * kSL + n] = hi;
          }
        }
      }
    }
  }
  if (p.w2 != nullptr) {                                 // y2[m][j] = sum_l y[m][l] W2[j][l]
    __syncthreads();
    for (int t = threadIdx.x; t < 16 * p.L2; t += 64 * kSW) {
      const int r = t / p.L2, j = t - r * p.L2, m = row0 + r;
      float a = 0.f;
#pragma unroll
      for (int l = 0; l < kSL; ++l) a = __builtin_fmaf(yrow[r][l], w2s[j * 33 + l], a);
      if (m < p.M) p.y2[(size_t)m * p.L2 + j] = a;
    }
  }
}

// ---- up projection ------------------------------------------------------------------------------------------------------------------
struct Up2Args {                                        // optional down-projection of the rows side_up has just written
  const float* w; const float* bias; float* z; float* y; int act;            // w [kSL][C] (wl 0) or [C][kSL] (wl 1); z / y [M][kSL]
  const float* dy;                                                           // LNM 2: the LayerNorm output gradient [M][C]
  // LNM 3: a second rank-L term added behind the LayerNorm backward, out = base + LN'(lat . W^T) + lat2 . W2up^T  (w2up [kSL][C]), and the
  // down-projection above reads the rows through a dropout mask (the MWSA backward across a layer boundary, engine._mwsa_chain_bwd)
  const float* lat2; const float* w2up;
  int wl;
  unsigned long long seed; const unsigned long long* seed_ptr; unsigned int drop_thresh; float inv_keep;
  // LNM 0, optional: the NEXT layer's MWSA entry on the rows just written (gaviko.py:231-232 of layer i+1 behind :242 of layer i):
  // lat = LayerNorm(out; g3, b3) . W3^T + bias3 (W3 [kSL][C]; mean3 / rstd3 saved), y4 = lat . W4^T (W4 [L4][kSL], L4 <= 64)
  const float* w3; const float* bias3; const float* g3; const float* b3; float* mean3; float* rstd3; float* y3;
  const float* w4; float* y4; int L4; float eps3;
};

// NPW = 32-column pairs per wave, WL = weight layout (0: w [C][L], 1: w [L][C]).  LNM selects the epilogue on v = lat . W^T:
//   0: out = base + dropout(v + bias)            1: out = base + LN'(v; ln_x, mean, rstd, gamma)            (gvk_skinny_up)
//   2: out = base + LN'(dy; ln_x, mean, rstd, gamma) + v, + bf16 copy     (gvk_layernorm_bwd_up: the MLP block's LayerNorm backward and GPA's
//      dG1 += dzx . W_d in one pass over the row, engine._backward_segment)
//   3: out = base + LN'(v; ...) + lat2 . W2up^T, then y2 = (out o dropout mask) . W3^T: three launches of the MWSA backward in one pass --
//      layer i+1's dL_in = dL_out + LN'(dlat . Wd) (gaviko.py:231), layer i's GPA share dL += dzl . Wd_gpa (:156) and layer i's
//      dctx = proj_drop'(dL) . Wup (:242-243) -- the local-stream gradient is read once and written once instead of 3 + 2 times
template <int NPW, int WL, int LNM>
__global__ __launch_bounds__(64 * kSW, NPW <= 3 ? 4 : 2) void side_up_kernel(UpArgs p, Up2Args q) {
  constexpr bool LNB = LNM != 0;
  __shared__ float red[kSW][16][2];
  __shared__ f32x4 part[kSW][2][64];
  __shared__ float yrow3[LNM == 0 ? 16 : 1][33];        // LNM 0: the next layer's latent rows of this tile, and its second-stage weight
  __shared__ float w4s[LNM == 0 ? 64 * 33 : 1];
  const int lane = lane_id(), wave = wave_id();
  const int a = lane & 15, kq = lane >> 4;              // A operand: weight row a of a column tile; B operand / D: token row a
  const int C = p.C, NP = C >> 5;
  const int row0 = (blockIdx.x + p.tile_off) * 16;
  const int row = min(row0 + a, p.M - 1);
  const bool rvalid = row0 + a < p.M;
  const float* __restrict__ base = p.accumulate ? p.out : p.res;
  // ---- loads: base (+ LayerNorm input) pieces of 8 columns, the latent row, the weight fragments, bias / gamma
  f32x4 bs[NPW][2], xs[LNB ? NPW : 1][2], vec[LNB ? 1 : NPW][2];
  [[maybe_unused]] f32x4 ys[LNM == 2 ? NPW : 1][2];
  float wa[NPW][2][kK4], lb[kK4];
  int cc[NPW];
  bool ok[NPW];
#pragma unroll
  for (int pi = 0; pi < NPW; ++pi) {
    const int pp = wave + kSW * pi;
    ok[pi] = pp < NP;
    cc[pi] = 32 * (ok[pi] ? pp : 0) + 8 * kq;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bs[pi][j] = base != nullptr ? ld_stream(base + (size_t)row * C + cc[pi] + 4 * j) : zero4();
      if constexpr (LNB) xs[pi][j] = ld_stream(p.ln_x + (size_t)row * C + cc[pi] + 4 * j);
      if constexpr (LNM == 2) ys[pi][j] = ld_stream(q.dy + (size_t)row * C + cc[pi] + 4 * j);
    }
  }
  {
    const float* src = p.lat + (size_t)row * kSL;
    if (p.lat_override != nullptr) {
      const int sidx = row / p.T, t = row - sidx * p.T;
      if (t < p.P) src = p.lat_override + ((size_t)sidx * p.P + t) * kSL;
    }
#pragma unroll
    for (int s = 0; s < kK4; ++s) lb[s] = src[kq * kK4 + s];
  }
#pragma unroll
  for (int pi = 0; pi < NPW; ++pi)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = cc[pi] - 8 * kq + 8 * (a >> 2) + 4 * j + (a & 3);      // weight row fed to A-operand row a of column tile j
      if constexpr (LNM != 2) {                          // LNM 2 forms its rank-L term late, where it is added (three wide streams are live until then)
#pragma unroll
        for (int s = 0; s < kK4; ++s)
          wa[pi][j][s] = WL == 0 ? p.w[(size_t)c * kSL + kq * kK4 + s] : p.w[(size_t)(kq * kK4 + s) * C + c];
      }
      if constexpr (!LNB) vec[pi][j] = p.bias != nullptr ? *(const f32x4*)(p.bias + cc[pi] + 4 * j) : zero4();
    }
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  [[maybe_unused]] float mu = 0.f, rs = 0.f;
  if constexpr (LNB) { mu = p.ln_mean[row]; rs = p.ln_rstd[row]; }
  [[maybe_unused]] float lb2[kK4];                       // LNM 3: the second latent row (v2 = lat2 . W2up^T is formed late, where it is added)
  if constexpr (LNM == 3) {
#pragma unroll
    for (int s = 0; s < kK4; ++s) lb2[s] = q.lat2[(size_t)row * kSL + kq * kK4 + s];
  }
  // ---- D[column a of tile j][token row] = sum over the latent index
  f32x4 acc[NPW][2];
#pragma unroll
  for (int pi = 0; pi < NPW; ++pi)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc[pi][j] = zero4();
      if constexpr (LNM != 2) {
#pragma unroll
        for (int s = 0; s < kK4; ++s) acc[pi][j] = mfma4(wa[pi][j][s], lb[s], acc[pi][j]);
      }
    }
  // ---- epilogue: lane = token row a, columns cc + (4 j + e)
  [[maybe_unused]] float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int pi = 0; pi < NPW; ++pi) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 v = acc[pi][j];
      if constexpr (LNB) {
        const f32x4 gam = *(const f32x4*)(p.ln_g + cc[pi] + 4 * j);       // cache-resident: fetched at its use
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dh = (LNM == 2 ? ys[pi][j][e] : v[e]) * gam[e], xh = (xs[pi][j][e] - mu) * rs;
          v[e] = dh;
          xs[pi][j][e] = xh;
          s1 += ok[pi] ? dh : 0.f;
          s2 += ok[pi] ? dh * xh : 0.f;
        }
      } else {
        v += vec[pi][j];
        if (p.drop_thresh != 0u) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= drop_scale(p.seed, (unsigned long long)row * C + cc[pi] + 4 * j + e, p.drop_thresh, p.inv_keep);
        }
        v += bs[pi][j];
        if (rvalid && ok[pi]) st_stream(p.out + (size_t)row * C + cc[pi] + 4 * j, v);
      }
      acc[pi][j] = v;
    }
    if (!LNB && p.out16 != nullptr && rvalid && ok[pi]) {
      bf16x8 h8;
#pragma unroll
      for (int e = 0; e < 4; ++e) { h8[e] = (bf16)acc[pi][0][e]; h8[4 + e] = (bf16)acc[pi][1][e]; }
      *(bf16x8*)(p.out16 + (size_t)row * C + cc[pi]) = h8;
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
    for (int pi = 0; pi < NPW; ++pi) {
      [[maybe_unused]] f32x4 v2[2];
      if constexpr (LNM == 3) {                          // lat2 . W2up^T (w2up [kSL][C]) for these columns; every lane takes part in the MFMAs
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = cc[pi] - 8 * kq + 8 * (a >> 2) + 4 * j + (a & 3);
          v2[j] = zero4();
#pragma unroll
          for (int s = 0; s < kK4; ++s) v2[j] = mfma4(q.w2up[(size_t)(kq * kK4 + s) * C + c], lb2[s], v2[j]);
        }
      }
      if constexpr (LNM == 2) {                          // the rank-L term lat . W^T, formed here (weights fetched at use)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = cc[pi] - 8 * kq + 8 * (a >> 2) + 4 * j + (a & 3);
          v2[j] = zero4();
#pragma unroll
          for (int s = 0; s < kK4; ++s)
            v2[j] = mfma4(WL == 0 ? p.w[(size_t)c * kSL + kq * kK4 + s] : p.w[(size_t)(kq * kK4 + s) * C + c], lb[s], v2[j]);
        }
      }
      if (rvalid && ok[pi]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs * (acc[pi][j][e] - s1 - xs[pi][j][e] * s2) + bs[pi][j][e];
          if constexpr (LNM == 2 || LNM == 3) o += v2[j];
          st_stream(p.out + (size_t)row * C + cc[pi] + 4 * j, o);
          acc[pi][j] = o;
        }
        if (LNM == 2 && p.out16 != nullptr) {
          bf16x8 h8;
#pragma unroll
          for (int e = 0; e < 4; ++e) { h8[e] = (bf16)acc[pi][0][e]; h8[4 + e] = (bf16)acc[pi][1][e]; }
          *(bf16x8*)(p.out16 + (size_t)row * C + cc[pi]) = h8;
        }
      }
    }
  }
  if constexpr (LNM == 0 || LNM == 3) {
    if (q.w != nullptr) {
      // second projection of the rows just written: A operand = the output piece itself (token row a, k <-> column cc + 4 j + e), B operand
      // W2[n][cc + 4 j + e]; rows / column groups past the end contribute zeros
      if (q.drop_thresh != 0u && q.seed_ptr != nullptr) q.seed += *q.seed_ptr;
      f32x4 d2[2] = {zero4(), zero4()};
#pragma unroll
      for (int pi = 0; pi < NPW; ++pi) {
        f32x4 w2f[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (q.wl == 0) {
            w2f[j][0] = load_w_down<0>(q.w, C, a, cc[pi] + 4 * j);
            w2f[j][1] = load_w_down<0>(q.w, C, 16 + a, cc[pi] + 4 * j);
          } else {
            w2f[j][0] = load_w_down<1>(q.w, C, a, cc[pi] + 4 * j);
            w2f[j][1] = load_w_down<1>(q.w, C, 16 + a, cc[pi] + 4 * j);
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 ov = (ok[pi] && rvalid) ? acc[pi][j] : zero4();
          if (q.drop_thresh != 0u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] *= drop_scale(q.seed, (unsigned long long)row * C + cc[pi] + 4 * j + e, q.drop_thresh, q.inv_keep);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) d2[0] = mfma4(ov[e], w2f[j][0][e], d2[0]);
#pragma unroll
          for (int e = 0; e < 4; ++e) d2[1] = mfma4(ov[e], w2f[j][1][e], d2[1]);
        }
      }
      part[wave][0][lane] = d2[0];
      part[wave][1][lane] = d2[1];
      __syncthreads();
      if (wave < 2) {
        f32x4 t = part[0][wave][lane];
#pragma unroll
        for (int w = 1; w < kSW; ++w) t += part[w][wave][lane];
        const int n = 16 * wave + a;
        if (n < kSL) {
          const float b = q.bias != nullptr ? q.bias[n] : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int m = row0 + 4 * kq + e;
            const float zz = t[e] + b;
            if (m < p.M) {
              if (q.z) q.z[(size_t)m * kSL + n] = zz;
              if (q.y) q.y[(size_t)m * kSL + n] = q.act == 1 ? quick_gelu(zz) : zz;
            }
          }
        }
      }
    }
  }
  if constexpr (LNM == 0) {
    if (q.w3 != nullptr) {
      // ---- the next layer's LayerNorm + proj_down (+ qkv) of the rows in acc: two-pass statistics like torch's LayerNorm
      for (int t = threadIdx.x; t < q.L4 * kSL; t += 64 * kSW) {
        const int j = t / kSL, l = t - j * kSL;
        w4s[j * 33 + l] = q.w4[t];
      }
      float s = 0.f;
#pragma unroll
      for (int pi = 0; pi < NPW; ++pi)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += ok[pi] ? (acc[pi][j][0] + acc[pi][j][1]) + (acc[pi][j][2] + acc[pi][j][3]) : 0.f;
      s = kq_sum(s);
      __syncthreads();                                   // the second projection is done with `part` / nobody still reads `red`
      if (kq == 0) red[wave][a][0] = s;
      __syncthreads();
      float mean = 0.f;
#pragma unroll
      for (int w = 0; w < kSW; ++w) mean += red[w][a][0];
      mean /= (float)C;
      float qs = 0.f;
#pragma unroll
      for (int pi = 0; pi < NPW; ++pi)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = acc[pi][j][e] - mean; qs += ok[pi] ? d * d : 0.f; }
      qs = kq_sum(qs);
      if (kq == 0) red[wave][a][1] = qs;
      __syncthreads();
      float var = 0.f;
#pragma unroll
      for (int w = 0; w < kSW; ++w) var += red[w][a][1];
      const float rstd = rsqrtf(var / (float)C + q.eps3);
      if (wave == 0 && kq == 0 && rvalid) {
        if (q.mean3) q.mean3[row] = mean;
        if (q.rstd3) q.rstd3[row] = rstd;
      }
      f32x4 d3[2] = {zero4(), zero4()};
#pragma unroll
      for (int pi = 0; pi < NPW; ++pi) {
        f32x4 w3f[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          w3f[j][0] = load_w_down<0>(q.w3, C, a, cc[pi] + 4 * j);
          w3f[j][1] = load_w_down<0>(q.w3, C, 16 + a, cc[pi] + 4 * j);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 g4 = *(const f32x4*)(q.g3 + cc[pi] + 4 * j), b4 = *(const f32x4*)(q.b3 + cc[pi] + 4 * j);
          f32x4 xn;
#pragma unroll
          for (int e = 0; e < 4; ++e) xn[e] = ok[pi] ? (acc[pi][j][e] - mean) * rstd * g4[e] + b4[e] : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) d3[0] = mfma4(xn[e], w3f[j][0][e], d3[0]);
#pragma unroll
          for (int e = 0; e < 4; ++e) d3[1] = mfma4(xn[e], w3f[j][1][e], d3[1]);
        }
      }
      part[wave][0][lane] = d3[0];
      part[wave][1][lane] = d3[1];
      __syncthreads();
      if (wave < 2) {
        f32x4 t = part[0][wave][lane];
#pragma unroll
        for (int w = 1; w < kSW; ++w) t += part[w][wave][lane];
        const int n = 16 * wave + a;
        if (n < kSL) {
          const float b = q.bias3 != nullptr ? q.bias3[n] : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * kq + e, m = row0 + r;
            const float yy = t[e] + b;
            yrow3[r][n] = yy;
            if (m < p.M && q.y3) q.y3[(size_t)m * kSL + n] = yy;
          }
        }
      }
      if (q.w4 != nullptr) {                             // y4[m][j] = sum_l lat[m][l] W4[j][l]
        __syncthreads();
        for (int t = threadIdx.x; t < 16 * q.L4; t += 64 * kSW) {
          const int r = t / q.L4, j = t - r * q.L4, m = row0 + r;
          float v = 0.f;
#pragma unroll
          for (int l = 0; l < kSL; ++l) v = __builtin_fmaf(yrow3[r][l], w4s[j * 33 + l], v);
          if (m < p.M) q.y4[(size_t)m * q.L4 + j] = v;
        }
      }
    }
  }
}

// diagnostics (GAVIKO_HIP_SIDE_CHUNKS = k): a launch is issued as k back-to-back launches over row ranges -- the same work at 1/k of the
// concurrency, to see whether the backbone kernels suffer from the side kernels' bandwidth BURST or from their total work
static int side_chunks() {
  static const int k = diag_env("GAVIKO_HIP_SIDE_CHUNKS") ? max(1, atoi(diag_env("GAVIKO_HIP_SIDE_CHUNKS"))) : 1;
  return k;
}

static bool side_enabled() {
  static const bool on = diag_env("GAVIKO_HIP_SIDE") == nullptr || diag_env("GAVIKO_HIP_SIDE")[0] != '0';   // A/B switch: 0 = the row-per-wave kernels
  return on;
}

// 32-column groups per wave for the widths of the three backbones (vit-t16 192, vit-b16 768, vit-l16 1024); 0 = not covered
static int groups_per_wave(int C) { return C == 192 ? 1 : C == 768 ? 3 : C == 1024 ? 4 : 0; }

// returns 1 when the call is not covered (the caller falls back to the row-per-wave / MFMA-tile kernels)
int launch_side_down(const DownArgs& a, int L, hipStream_t s) {
  // mode 3 (DVPT's QuickGELU input) stays on the row-per-wave kernel: its scalar prompt_gate gradient is one signed sum over every latent
  // of the batch and is pinned at 1e-4 on the fp32 path with that kernel's summation order
  const int ngw = groups_per_wave(a.C);
  if (!side_enabled() || L != kSL || ngw == 0 || a.mode < 0 || a.mode > 2 || (a.w2 != nullptr && a.L2 > 64)) return 1;
  if (a.mode != 0 && (a.w2 != nullptr || a.drop_thresh != 0u)) return 1;
  if (a.dy16 != nullptr) return 1;                                  // bf16 gradient input: the row-per-wave kernel
  // A/B switch, see DESIGN.md section 7b: '1' = both LayerNorm modes on the tile kernels, 'f' = the forward one (MODE 1), 'b' = the backward one
  static const char ln_sel = diag_env("GAVIKO_HIP_SIDE_LN") != nullptr ? diag_env("GAVIKO_HIP_SIDE_LN")[0] : 'f';
  if (a.mode == 1 && !(ln_sel == '1' || ln_sel == 'f')) return 1;
  if (a.mode == 2 && !(ln_sel == '1' || ln_sel == 'b')) return 1;
  const int tiles = (a.M + 15) / 16, chunks = side_chunks(), per = (tiles + chunks - 1) / chunks;
  const dim3 block(64 * kSW);
  DownArgs ac = a;
  for (int t0 = 0; t0 < tiles; t0 += per) {
  ac.tile_off = t0;
  const dim3 grid(min(per, tiles - t0));
#define GVK_SD(N_, W_, M_) GVK_LAUNCH((side_down_kernel<N_, W_, M_>), grid, block, 0, s, ac)
#define GVK_SD_N(W_, M_) { if (ngw == 1) GVK_SD(1, W_, M_); else if (ngw == 3) GVK_SD(3, W_, M_); else GVK_SD(4, W_, M_); }
  if (a.mode == 0) { if (a.w_layout == 0) GVK_SD_N(0, 0) else GVK_SD_N(1, 0) }
  else if (a.mode == 1) { if (a.w_layout == 0) GVK_SD_N(0, 1) else GVK_SD_N(1, 1) }
  else { if (a.w_layout == 0) GVK_SD_N(0, 2) else GVK_SD_N(1, 2) }
#undef GVK_SD_N
#undef GVK_SD
  }
  return check_launch("side_down");
}

int launch_side_up(const UpArgs& a, int L, const float* w2, const float* bias2, float* z2, float* y2, int L2, int act2, hipStream_t s, const float* ln_dy,
                   const UpExtra* ex, const UpNext* nx) {
  const int npw = groups_per_wave(a.C);
  const bool lnb = a.ln_x != nullptr, ext = a.alpha_ptr != nullptr || a.gg_x != nullptr;
  if (!side_enabled() || L != kSL || npw == 0 || ext) return 1;       // DVPT's gate / GELU' epilogue: row-per-wave kernel (see launch_side_down)
  if (w2 != nullptr && ((lnb && ex == nullptr) || L2 != kSL))
    return set_error(-2, "side_up: the fused second projection takes the plain epilogue (or the layer-boundary form) and L2 = %d", kSL);
  if (ln_dy != nullptr && !lnb) return set_error(-2, "side_up: ln_dy needs the LayerNorm operands (ln_x, mean, rstd, gamma)");
  if (ex != nullptr && (!lnb || ln_dy != nullptr || w2 == nullptr || ex->lat2 == nullptr || ex->w2up == nullptr || a.w_layout != 1))
    return set_error(-2, "side_up: the layer-boundary form needs the LayerNorm operands, lat2 / w2up, a second projection and w_layout 1");
  Up2Args q{};
  q.w = w2; q.bias = bias2; q.z = z2; q.y = y2; q.act = act2; q.dy = ln_dy;
  if (ex != nullptr) {
    q.lat2 = ex->lat2; q.w2up = ex->w2up; q.wl = ex->w2_layout;
    q.seed = ex->seed2; q.seed_ptr = ex->seed_ptr; q.drop_thresh = ex->drop2_thresh; q.inv_keep = ex->inv_keep2;
  }
  const int lnm = ex != nullptr ? 3 : ln_dy != nullptr ? 2 : (lnb ? 1 : 0);
  if (nx != nullptr) {
    if (lnm != 0 || !nx->w || !nx->g || !nx->b || !nx->lat || (nx->w2 != nullptr && (!nx->y2 || nx->L2 <= 0 || nx->L2 > 64)))
      return set_error(-2, "side_up: the next-layer stage takes the plain epilogue, w / gamma / beta / lat and L2 <= 64");
    q.w3 = nx->w; q.bias3 = nx->bias; q.g3 = nx->g; q.b3 = nx->b; q.mean3 = nx->mean; q.rstd3 = nx->rstd; q.y3 = nx->lat;
    q.w4 = nx->w2; q.y4 = nx->y2; q.L4 = nx->L2; q.eps3 = nx->eps > 0.f ? nx->eps : 1e-5f;
  }
  const int tiles = (a.M + 15) / 16, chunks = side_chunks(), per = (tiles + chunks - 1) / chunks;
  const dim3 block(64 * kSW);
  UpArgs ac = a;
  for (int t0 = 0; t0 < tiles; t0 += per) {
  ac.tile_off = t0;
  const dim3 grid(min(per, tiles - t0));
#define GVK_SU(N_, W_, B_) GVK_LAUNCH((side_up_kernel<N_, W_, B_>), grid, block, 0, s, ac, q)
#define GVK_SU_N(W_, B_) { if (npw == 1) GVK_SU(1, W_, B_); else if (npw == 3) GVK_SU(3, W_, B_); else GVK_SU(4, W_, B_); }
  if (lnm == 3) GVK_SU_N(1, 3)
  else if (a.w_layout == 0) { if (lnm == 2) GVK_SU_N(0, 2) else if (lnm == 1) GVK_SU_N(0, 1) else GVK_SU_N(0, 0) }
  else { if (lnm == 2) GVK_SU_N(1, 2) else if (lnm == 1) GVK_SU_N(1, 1) else GVK_SU_N(1, 0) }
#undef GVK_SU_N
#undef GVK_SU
  }
  return check_launch("side_up");
}

}  // namespace gvk
