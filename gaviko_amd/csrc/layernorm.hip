// LayerNorm forward / backward, one 64-lane wave per token row (HBM-bound; rows stay in registers).
// fp32 statistics with the two-pass variance torch's CPU kernel uses (mean, then mean((x-mean)^2)).
#include "common.hpp"
#include "skinny_args.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

constexpr int kMaxChunks = 4;   // C <= 1024: up to four float4 per lane

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16* __restrict__ y16, float* __restrict__ y32,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o, int M, int C, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = lane_id();
  const float* xr = x + (size_t)row * C;
  f32x4 v[kMaxChunks];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxChunks; ++k) {
    const int c = k * 256 + lane * 4;
    v[k] = (c < C) ? *(const f32x4*)(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxChunks; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[k][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    if (mean_o) mean_o[row] = mean;
    if (rstd_o) rstd_o[row] = rstd;
  }
#pragma unroll
  for (int k = 0; k < kMaxChunks; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
      const f32x4 g = *(const f32x4*)(gamma + c);
      const f32x4 b = *(const f32x4*)(beta + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[k][e] - mean) * rstd * g[e] + b[e];
      if (y16) {
        bf16x4 h = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
        *(bf16x4*)(y16 + (size_t)row * C + c) = h;
      }
      if (y32) *(f32x4*)(y32 + (size_t)row * C + c) = o;
    }
  }
}

// The same with the GPA prompt fix of the PREVIOUS layer applied on the way in: rows with (row % T) < P first receive
// x[row] += (enh[b][p] - lat[row]) . Wup^T (Wup [C][L]; gaviko.py:183-187 -- the plain latents already rode the fc2 GEMM, elementwise.hip)
// and are written back, then every row is normalised as usual.  Saves the separate 128-row fix launch on the backbone stream.
__global__ __launch_bounds__(256) void ln_fwd_fix_kernel(float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         bf16* __restrict__ y16, float* __restrict__ mean_o, float* __restrict__ rstd_o, int M, int C,
                                                         float eps, const float* __restrict__ enh, const float* __restrict__ lat,
                                                         const float* __restrict__ wup, int T, int P, int L) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = lane_id();
  float* xr = x + (size_t)row * C;
  f32x4 v[kMaxChunks];
#pragma unroll
  for (int k = 0; k < kMaxChunks; ++k) {
    const int c = k * 256 + lane * 4;
    v[k] = (c < C) ? *(const f32x4*)(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int b = row / T, t = row - b * T;
  if (t < P) {                                           // wave-uniform: a prompt row
    // the lane's four columns are four CONSECUTIVE rows of Wup [C][L]: 4 L contiguous floats, read as float4s (L % 4 == 0) -- element j of
    // that run belongs to column c + j / L, latent j % L
    const float dl = lane < L ? enh[((size_t)b * P + t) * L + lane] - lat[(size_t)row * L + lane] : 0.f;
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
        const f32x4* wr = (const f32x4*)(wup + (size_t)c * L);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < 4; ++e) {
          float acc = 0.f;
          for (int l4 = 0; l4 < L / 4; ++l4) {
            const f32x4 w4 = wr[e * (L / 4) + l4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_fmaf(__shfl(dl, 4 * l4 + u, 64), w4[u], acc);
          }
          a[e] = acc;
        }
        v[k] += a;
        *(f32x4*)(xr + c) = v[k];
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxChunks; ++k) s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxChunks; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[k][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    if (mean_o) mean_o[row] = mean;
    if (rstd_o) rstd_o[row] = rstd;
  }
#pragma unroll
  for (int k = 0; k < kMaxChunks; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
      const f32x4 g = *(const f32x4*)(gamma + c);
      const f32x4 bb = *(const f32x4*)(beta + c);
      bf16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (bf16)((v[k][e] - mean) * rstd * g[e] + bb[e]);
      *(bf16x4*)(y16 + (size_t)row * C + c) = h;
    }
  }
}

// dx = dres + rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat))
template <bool DY16>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy_, const float* __restrict__ x,
                                                     const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     float* __restrict__ dx, bf16* __restrict__ dx16, int M, int C, int rpg, int gstride) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  if (rpg > 0) row = (row / rpg) * gstride + row % rpg;       // a row subset: the first rpg rows of every group of gstride rows (M = groups * rpg)
  const int lane = lane_id();
  const float mean = mean_i[row], rstd = rstd_i[row];
  f32x4 xh[kMaxChunks], dh[kMaxChunks];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxChunks; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
      const f32x4 xv = *(const f32x4*)(x + (size_t)row * C + c);
      f32x4 dv;
      if constexpr (DY16) {                                  // the gradient as a dgrad GEMM stored it (bf16)
        const bf16x4 h = *(const bf16x4*)((const bf16*)dy_ + (size_t)row * C + c);
        dv = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
      } else {
        dv = *(const f32x4*)((const float*)dy_ + (size_t)row * C + c);
      }
      const f32x4 g = *(const f32x4*)(gamma + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[k][e] = (xv[e] - mean) * rstd;
        dh[k][e] = dv[e] * g[e];
        s1 += dh[k][e];
        s2 += dh[k][e] * xh[k][e];
      }
    } else {
      xh[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      dh[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
  for (int k = 0; k < kMaxChunks; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rstd * (dh[k][e] - m1 - xh[k][e] * m2);
      if (dres) {
        const f32x4 r = *(const f32x4*)(dres + (size_t)row * C + c);
        o += r;
      }
      *(f32x4*)(dx + (size_t)row * C + c) = o;
      if (dx16) {
        bf16x4 h = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
        *(bf16x4*)(dx16 + (size_t)row * C + c) = h;
      }
    }
  }
}

// Affine grads, deterministic two-stage: stage 1 = 64 row-slabs x column chunks, stage 2 = sum the 64 partials.
__global__ __launch_bounds__(256) void ln_affine_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                                float* __restrict__ scratch, int M, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int slab = blockIdx.y;                  // 0..63
  const int rows_per = (M + 63) / 64;
  const int r0 = slab * rows_per, r1 = min(M, r0 + rows_per);
  float dg = 0.f, db = 0.f;
  if (c < C) {
    for (int r = r0; r < r1; ++r) {
      const float d = dy[(size_t)r * C + c];
      dg += d * (x[(size_t)r * C + c] - mean_i[r]) * rstd_i[r];
      db += d;
    }
    scratch[(size_t)slab * C + c] = dg;
    scratch[(size_t)(64 + slab) * C + c] = db;
  }
}
__global__ __launch_bounds__(256) void ln_affine_final_kernel(const float* __restrict__ scratch, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int C, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float dg = 0.f, db = 0.f;
  for (int s = 0; s < 64; ++s) {
    dg += scratch[(size_t)s * C + c];
    db += scratch[(size_t)(64 + s) * C + c];
  }
  if (accumulate) {
    dgamma[c] += dg;
    dbeta[c] += db;
  } else {
    dgamma[c] = dg;
    dbeta[c] = db;
  }
}

}  // namespace gvk

extern "C" int gvk_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32, float* mean,
                                 float* rstd, int M, int C, float eps, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x && gamma && beta && (y_bf16 || y_f32), "gvk_layernorm_fwd: null pointer");
  GVK_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 256 * kMaxChunks, "gvk_layernorm_fwd: C=%d must be a multiple of 4 and <= 1024", C);
  GVK_LAUNCH(ln_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (bf16*)y_bf16, y_f32, mean,
                     rstd, M, C, eps);
  return check_launch("layernorm_fwd");
}

extern "C" int gvk_layernorm_fwd_fix(float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd, int M, int C, float eps,
                                     const float* enh, const float* lat, const float* wup, int T, int P, int L, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x && gamma && beta && y_bf16 && enh && lat && wup, "gvk_layernorm_fwd_fix: null pointer");
  GVK_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 256 * kMaxChunks, "gvk_layernorm_fwd_fix: C=%d must be a multiple of 4 and <= 1024", C);
  GVK_REQUIRE(T > 0 && P > 0 && P <= T && L > 0 && L <= 64 && L % 4 == 0 && M % T == 0, "gvk_layernorm_fwd_fix: need 0 < P <= T, L <= 64 and a multiple of 4, M a multiple of T");
  GVK_LAUNCH(ln_fwd_fix_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (bf16*)y_bf16, mean, rstd, M, C,
             eps > 0.f ? eps : 1e-5f, enh, lat, wup, T, P, L);
  return check_launch("layernorm_fwd_fix");
}

extern "C" int gvk_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                 const float* dres, float* dx, void* dx_bf16, int M, int C, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(dy && x && mean && rstd && gamma && dx, "gvk_layernorm_bwd: null pointer");
  GVK_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 256 * kMaxChunks, "gvk_layernorm_bwd: C=%d must be a multiple of 4 and <= 1024", C);
  GVK_LAUNCH(ln_bwd_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const void*)dy, x, mean, rstd, gamma, dres, dx,
                     (bf16*)dx_bf16, M, C, 0, 0);
  return check_launch("layernorm_bwd");
}

extern "C" int gvk_layernorm_bwd_rows(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                      const float* dres, float* dx, void* dx_bf16, int groups, int rows_per_group, int group_stride, int C,
                                      void* stream) {
  using namespace gvk;
  GVK_REQUIRE(dy && x && mean && rstd && gamma && dx, "gvk_layernorm_bwd_rows: null pointer");
  GVK_REQUIRE(groups > 0 && rows_per_group > 0 && group_stride >= rows_per_group, "gvk_layernorm_bwd_rows: groups=%d rows_per_group=%d group_stride=%d",
              groups, rows_per_group, group_stride);
  GVK_REQUIRE(C > 0 && C % 4 == 0 && C <= 256 * kMaxChunks, "gvk_layernorm_bwd_rows: C=%d must be a multiple of 4 and <= 1024", C);
  const int M = groups * rows_per_group;
  GVK_LAUNCH(ln_bwd_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const void*)dy, x, mean, rstd, gamma, dres, dx,
                     (bf16*)dx_bf16, M, C, rows_per_group, group_stride);
  return check_launch("layernorm_bwd_rows");
}

// ---- LayerNorm with a fused rank-L projection of the rows it holds: the row-per-wave projection kernel of rowwise.hip run
// with a LayerNorm prologue (mode 1: forward, projects the raw rows; mode 2: backward, projects dx).
namespace gvk {
static int check_proj(const gvk_rowproj_desc* pj, int C, const char* who) {
  GVK_REQUIRE(pj && pj->w && pj->y, "%s: null projection operand", who);
  GVK_REQUIRE(C % 4 == 0 && C <= 256 * kMaxChunks, "%s: C=%d must be a multiple of 4 and <= 1024", who, C);
  GVK_REQUIRE(pj->w_layout == 0 || pj->w_layout == 1, "%s: w_layout must be 0 or 1", who);
  GVK_REQUIRE(pj->act == 0 || pj->act == 1, "%s: act must be 0 (none) or 1 (QuickGELU)", who);
  return 0;
}
static int launch_proj(DownArgs& a, const gvk_rowproj_desc* pj, hipStream_t s, const char* who) {
  a.w = pj->w; a.bias = pj->bias; a.y = pj->y; a.z = pj->z; a.act = pj->act; a.w_layout = pj->w_layout;
  a.ysplit = (bf16*)pj->y_split; a.ysplit_ld = pj->ld_split; a.ysplit_col = pj->col_split;
  GVK_REQUIRE(pj->y_split == nullptr || (pj->col_split >= 0 && pj->col_split + 3 * pj->L <= pj->ld_split), "%s: y_split slot out of range", who);
  int rc = launch_side_down(a, pj->L, s);                 // 16-row tiles on the fp32 matrix cores (sidepass.hip) where they apply
  if (rc == 1) rc = launch_row_down(a, pj->L, s);
  if (rc == 1) return set_error(-2, "%s: the fused projection covers L in {4, 8, 16, 20} and C >= 128 (got L=%d, C=%d): use gvk_skinny_down", who, pj->L, a.C);
  return rc;
}
}  // namespace gvk

extern "C" int gvk_layernorm_fwd_proj(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd,
                                      int M, int C, float eps, const gvk_rowproj_desc* proj, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x && gamma && beta && y_bf16, "gvk_layernorm_fwd_proj: null pointer");
  GVK_REQUIRE(M > 0 && C > 0, "gvk_layernorm_fwd_proj: empty shape");
  if (int rc = check_proj(proj, C, "gvk_layernorm_fwd_proj")) return rc;
  DownArgs a{};
  a.mode = 1; a.x = x; a.ln_g = gamma; a.ln_b = beta; a.y16 = (bf16*)y_bf16; a.mean = mean; a.rstd = rstd; a.M = M; a.C = C;
  a.eps = eps > 0.f ? eps : 1e-5f; a.inv_keep = 1.f;
  return launch_proj(a, proj, (hipStream_t)stream, "gvk_layernorm_fwd_proj");
}

extern "C" int gvk_layernorm_bwd_proj(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                      const float* dres, float* dx, void* dx_bf16, int M, int C, const gvk_rowproj_desc* proj,
                                      void* stream) {
  using namespace gvk;
  GVK_REQUIRE(dy && x && mean && rstd && gamma && dx, "gvk_layernorm_bwd_proj: null pointer");
  GVK_REQUIRE(M > 0 && C > 0, "gvk_layernorm_bwd_proj: empty shape");
  if (int rc = check_proj(proj, C, "gvk_layernorm_bwd_proj")) return rc;
  DownArgs a{};
  a.mode = 2; a.x = x; a.dy = dy; a.mean_in = mean; a.rstd_in = rstd; a.ln_g = gamma; a.dres = dres; a.dx = dx; a.dx16 = (bf16*)dx_bf16;
  a.M = M; a.C = C; a.eps = 1e-5f; a.inv_keep = 1.f;
  return launch_proj(a, proj, (hipStream_t)stream, "gvk_layernorm_bwd_proj");
}

// The three backward forms above with the output gradient in bf16 (what the dgrad GEMM in front stores: half the bytes on both sides)
extern "C" int gvk_layernorm_bwd_dy16(const gvk_ln_bwd_dy16_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->dy_bf16 && d->x && d->mean && d->rstd && d->gamma && d->dx, "gvk_layernorm_bwd_dy16: null pointer");
  const int C = d->C;
  GVK_REQUIRE(d->M > 0 && C > 0 && C % 4 == 0 && C <= 256 * kMaxChunks, "gvk_layernorm_bwd_dy16: C=%d must be a multiple of 4 and <= 1024", C);
  if (d->proj != nullptr) {
    GVK_REQUIRE(d->rows_per_group == 0, "gvk_layernorm_bwd_dy16: the projection form covers all rows");
    if (int rc = check_proj(d->proj, C, "gvk_layernorm_bwd_dy16")) return rc;
    DownArgs a{};
    a.mode = 2; a.x = d->x; a.dy16 = (const bf16*)d->dy_bf16; a.mean_in = d->mean; a.rstd_in = d->rstd; a.ln_g = d->gamma; a.dres = d->dres; a.dx = d->dx;
    a.dx16 = (bf16*)d->dx_bf16; a.M = d->M; a.C = C; a.eps = 1e-5f; a.inv_keep = 1.f;
    return launch_proj(a, d->proj, (hipStream_t)stream, "gvk_layernorm_bwd_dy16");
  }
  int M = d->M, rpg = 0, gs = 0;
  if (d->rows_per_group > 0) {
    GVK_REQUIRE(d->groups > 0 && d->group_stride >= d->rows_per_group && (long)(d->groups - 1) * d->group_stride + d->rows_per_group <= d->M,
                "gvk_layernorm_bwd_dy16: groups=%d rows_per_group=%d group_stride=%d do not fit M=%d", d->groups, d->rows_per_group, d->group_stride, d->M);
    M = d->groups * d->rows_per_group; rpg = d->rows_per_group; gs = d->group_stride;
  }
  GVK_LAUNCH(ln_bwd_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, d->dy_bf16, d->x, d->mean, d->rstd, d->gamma, d->dres, d->dx,
             (bf16*)d->dx_bf16, M, C, rpg, gs);
  return check_launch("layernorm_bwd_dy16");
}

// LayerNorm backward of the MLP block fused with GPA's rank-L scatter (gaviko.py:155: dG1 += dzx . W_d): one pass over the row
extern "C" int gvk_layernorm_bwd_up(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                                    float* dx, void* dx_bf16, const float* lat, const float* w, int w_layout, int M, int C, int L, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(dy && x && mean && rstd && gamma && dx && lat && w, "gvk_layernorm_bwd_up: null pointer");
  GVK_REQUIRE(M > 0 && C > 0 && (w_layout == 0 || w_layout == 1), "gvk_layernorm_bwd_up: bad shape / layout");
  UpArgs a{};
  a.lat = lat; a.w = w; a.res = dres; a.out = dx; a.out16 = (bf16*)dx_bf16; a.ln_x = x; a.ln_mean = mean; a.ln_rstd = rstd; a.ln_g = gamma;
  a.M = M; a.C = C; a.w_layout = w_layout; a.accumulate = 0; a.inv_keep = 1.f;
  const int rc = launch_side_up(a, L, nullptr, nullptr, nullptr, nullptr, 0, 0, (hipStream_t)stream, dy);
  if (rc == 1) return set_error(-2, "gvk_layernorm_bwd_up: covers L = 20 and C in {192, 768, 1024} (got L=%d, C=%d): use gvk_layernorm_bwd + gvk_skinny_up", L, C);
  return rc;
}

extern "C" int gvk_layernorm_bwd_affine(const float* dy, const float* x, const float* mean, const float* rstd, float* dgamma,
                                        float* dbeta, float* scratch, int M, int C, int accumulate, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(dy && x && mean && rstd && dgamma && dbeta && scratch, "gvk_layernorm_bwd_affine: null pointer");
  GVK_REQUIRE(M > 0 && C > 0, "gvk_layernorm_bwd_affine: empty shape");
  GVK_LAUNCH(ln_affine_partial_kernel, dim3((C + 255) / 256, 64), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd,
                     scratch, M, C);
  int rc = check_launch("layernorm_bwd_affine/partial");
  if (rc) return rc;
  GVK_LAUNCH(ln_affine_final_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, dgamma, dbeta, C,
                     accumulate);
  return check_launch("layernorm_bwd_affine/final");
}
