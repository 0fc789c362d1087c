// SSF (Scaling & Shifting Features, model/ssf.py): every ssf_ada(x, s, t) = x*s + t of the reference follows a LayerNorm or a
// Linear directly, so the forward is the plain ViT with EFFECTIVE parameters, recomputed every step from the trainable (s, t):
//   LayerNorm + ssf : gamma' = gamma*s, beta' = beta*s + t                       (ssf.py:65-66, 105-106, 138)
//   Linear + ssf    : W'[n][:] = s[n]*W[n][:],  b' = b*s + t                     (ssf.py:67-68, 72-73, 107, 120-121, 232)
// and the backward is the plain ViT backward through the effective weights plus, per site, the two column sums
//   dt[n] = sum_m dy[m][n]      ds[n] = sum_m dy[m][n] * z[m][n],   z = pre-ssf value = (y - t)/s
// taken from tensors the backward holds anyway (dy = the dgrad operand of that Linear; y = its saved output, or the
// difference of two residual-stream checkpoints for the out-proj / fc2 / patch-embed sites).
#include "common.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

// out[n][k] = W[n][k]*s[n] and out_t[k][n] = the same, 64x64 tiles through LDS (+1 pad): coalesced on both sides.
template <typename OUT>
__global__ __launch_bounds__(256) void ssf_fold_weight_kernel(const float* __restrict__ w, const float* __restrict__ s, OUT* __restrict__ out,
                                                              OUT* __restrict__ out_t, int N, int K) {
  __shared__ float tile[64][65];
  const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int n = n0 + i, k = k0 + tx;
    float v = 0.f;
    if (n < N && k < K) {
      v = w[(size_t)n * K + k] * s[n];
      out[(size_t)n * K + k] = (OUT)v;
    }
    tile[i][tx] = v;
  }
  if (out_t == nullptr) return;
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int k = k0 + i, n = n0 + tx;
    if (k < K && n < N) out_t[(size_t)k * N + n] = (OUT)tile[tx][i];
  }
}

__global__ __launch_bounds__(256) void ssf_fold_vec_kernel(const float* __restrict__ a, const float* __restrict__ s, const float* __restrict__ t,
                                                           float* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = (a != nullptr ? a[i] : 0.f) * s[i];
  if (t != nullptr) v += t[i];
  out[i] = v;
}

struct ColGradArgs {
  const void* dy; const void* y0; const float* y1; const float* pos;
  const float* s; const float* t; float* ds; float* dt; float* scratch;
  int M, N, ld_dy, ld_y, dy_f32, y0_f32;
  int rows_in, rows_out, row_off;          // rows_in > 0: logical row m -> buffer row (m / rows_in) * rows_out + row_off + m % rows_in
  int y0_cols; float y0_mul;               // y0 columns n < y0_cols are stored multiplied by 1 / y0_mul (the pre-scaled q block of a saved qkv)
  float y_mul;                             // y = y_mul * (y0 - y1) - pos (a site stored behind nn.Dropout: kept / (1 - p) -> kept)
};
constexpr int kCgSlabs = 64;

__device__ __forceinline__ float ld_mixed(const void* p, size_t i, int is_f32) {
  return is_f32 ? ((const float*)p)[i] : (float)((const bf16*)p)[i];
}

// stage 1: slab x column -> partial (sum dy*y, sum dy); consecutive threads take consecutive columns
__global__ __launch_bounds__(256) void ssf_colgrad_partial_kernel(ColGradArgs p) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int slab = blockIdx.y;
  const int per = (p.M + kCgSlabs - 1) / kCgSlabs;
  const int m0 = slab * per, m1 = min(p.M, m0 + per);
  if (n >= p.N) return;
  float a = 0.f, d = 0.f;
  for (int m = m0; m < m1; ++m) {
    size_t row = (size_t)m;
    int prow = 0;
    if (p.rows_in > 0) {
      const int smp = m / p.rows_in;
      prow = m - smp * p.rows_in;
      row = (size_t)smp * p.rows_out + p.row_off + prow;
    }
    const float dy = ld_mixed(p.dy, row * p.ld_dy + n, p.dy_f32);
    float y = ld_mixed(p.y0, row * p.ld_y + n, p.y0_f32);
    if (n < p.y0_cols) y *= p.y0_mul;
    if (p.y1 != nullptr) y -= p.y1[row * p.ld_y + n];
    y *= p.y_mul;
    if (p.pos != nullptr) y -= p.pos[(size_t)prow * p.N + n];
    a = __builtin_fmaf(dy, y, a);
    d += dy;
  }
  p.scratch[(size_t)slab * 2 * p.N + n] = a;
  p.scratch[(size_t)slab * 2 * p.N + p.N + n] = d;
}
// stage 2: ds = (sum dy*y - t * sum dy) / s,  dt = sum dy
__global__ __launch_bounds__(256) void ssf_colgrad_final_kernel(ColGradArgs p) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= p.N) return;
  float a = 0.f, d = 0.f;
  for (int sl = 0; sl < kCgSlabs; ++sl) {
    a += p.scratch[(size_t)sl * 2 * p.N + n];
    d += p.scratch[(size_t)sl * 2 * p.N + p.N + n];
  }
  p.dt[n] = d;
  p.ds[n] = (a - p.t[n] * d) / p.s[n];
}

// LayerNorm + ssf site: given the effective-affine gradients dgamma' = sum dy*xhat, dbeta' = sum dy:
//   ds = gamma*dgamma' + beta*dbeta',  dt = dbeta'
__global__ __launch_bounds__(256) void ssf_ln_grad_kernel(const float* __restrict__ dgp, const float* __restrict__ dbp, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ ds, float* __restrict__ dt, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  ds[i] = gamma[i] * dgp[i] + beta[i] * dbp[i];
  dt[i] = dbp[i];
}

// Final norm + ssf in front of the head (ssf.py:138, 240-246): only the pooled rows carry gradient.
//   dpn[b][c] = sum_k dlogits[b][k] Wh[k][c] / R ;  dgamma'[c] = sum_{b, r in pool} dpn[b][c] xhat[b,r][c] ;  dbeta'[c] = R * sum_b dpn[b][c]
// then folded as above.  One thread per column; mean/rstd of the rows come from a LayerNorm statistics pass.
__global__ __launch_bounds__(256) void ssf_head_grad_kernel(const float* __restrict__ g, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ wh, const float* __restrict__ dlogits,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ ds, float* __restrict__ dt, int B, int T, int C, int K, int r0, int R) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float dgp = 0.f, dbp = 0.f;
  for (int b = 0; b < B; ++b) {
    float dpn = 0.f;
    for (int k = 0; k < K; ++k) dpn += dlogits[b * K + k] * wh[(size_t)k * C + c];
    dpn /= (float)R;
    for (int r = r0; r < r0 + R; ++r) {
      const size_t row = (size_t)b * T + r;
      dgp += dpn * (g[row * C + c] - mean[row]) * rstd[row];
      dbp += dpn;
    }
  }
  ds[c] = gamma[c] * dgp + beta[c] * dbp;
  dt[c] = dbp;
}

}  // namespace gvk

extern "C" int gvk_ssf_fold_weight(const float* w, const float* s, void* out, void* out_t, int N, int K, int out_f32, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(w && s && out && N > 0 && K > 0, "gvk_ssf_fold_weight: bad arguments");
  const dim3 grid((K + 63) / 64, (N + 63) / 64);
  if (out_f32) GVK_LAUNCH(ssf_fold_weight_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, w, s, (float*)out, (float*)out_t, N, K);
  else GVK_LAUNCH(ssf_fold_weight_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, w, s, (bf16*)out, (bf16*)out_t, N, K);
  return check_launch("ssf_fold_weight");
}

extern "C" int gvk_ssf_fold_vec(const float* a, const float* s, const float* t, float* out, int n, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(s && out && n > 0, "gvk_ssf_fold_vec: bad arguments");
  GVK_LAUNCH(ssf_fold_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, s, t, out, n);
  return check_launch("ssf_fold_vec");
}

extern "C" int gvk_ssf_colgrad(const gvk_ssf_colgrad_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->dy && d->y0 && d->s && d->t && d->ds && d->dt && d->scratch, "gvk_ssf_colgrad: null pointer");
  GVK_REQUIRE(d->M > 0 && d->N > 0 && d->ld_dy >= d->N && d->ld_y >= d->N, "gvk_ssf_colgrad: bad shape");
  GVK_REQUIRE(d->rows_in == 0 || (d->rows_in > 0 && d->rows_out >= d->rows_in + d->row_off), "gvk_ssf_colgrad: bad row mapping");
  GVK_REQUIRE(d->pos == nullptr || d->rows_in > 0, "gvk_ssf_colgrad: pos needs the row mapping");
  ColGradArgs a{d->dy, d->y0, d->y1, d->pos, d->s, d->t, d->ds, d->dt, d->scratch, d->M, d->N, d->ld_dy, d->ld_y, d->dy_f32, d->y0_f32,
                d->rows_in, d->rows_out, d->row_off, d->y0_cols, d->y0_mul, d->y_mul == 0.f ? 1.f : d->y_mul};
  GVK_LAUNCH(ssf_colgrad_partial_kernel, dim3((d->N + 255) / 256, kCgSlabs), dim3(256), 0, (hipStream_t)stream, a);
  int rc = check_launch("ssf_colgrad/partial");
  if (rc) return rc;
  GVK_LAUNCH(ssf_colgrad_final_kernel, dim3((d->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("ssf_colgrad/final");
}

extern "C" int gvk_ssf_ln_grad(const float* dgamma_eff, const float* dbeta_eff, const float* gamma, const float* beta, float* ds, float* dt, int n,
                               void* stream) {
  using namespace gvk;
  GVK_REQUIRE(dgamma_eff && dbeta_eff && gamma && beta && ds && dt && n > 0, "gvk_ssf_ln_grad: bad arguments");
  GVK_LAUNCH(ssf_ln_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dgamma_eff, dbeta_eff, gamma, beta, ds, dt, n);
  return check_launch("ssf_ln_grad");
}

extern "C" int gvk_ssf_head_grad(const float* g, const float* mean, const float* rstd, const float* wh, const float* dlogits, const float* gamma,
                                 const float* beta, float* ds, float* dt, int B, int T, int C, int K, int r0, int R, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(g && mean && rstd && wh && dlogits && gamma && beta && ds && dt, "gvk_ssf_head_grad: null pointer");
  GVK_REQUIRE(B > 0 && T > 0 && C > 0 && K > 0 && r0 >= 0 && R > 0 && r0 + R <= T, "gvk_ssf_head_grad: bad shape");
  GVK_LAUNCH(ssf_head_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, mean, rstd, wh, dlogits, gamma, beta, ds, dt, B, T, C,
             K, r0, R);
  return check_launch("ssf_head_grad");
}
