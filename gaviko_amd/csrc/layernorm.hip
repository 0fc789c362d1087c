// LayerNorm forward / backward, one 64-lane wave per token row (HBM-bound; rows stay in registers).
// fp32 statistics with the two-pass variance torch's CPU kernel uses (mean, then mean((x-mean)^2)).
#include "common.hpp"
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

// dx = dres + rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat))
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     float* __restrict__ dx, bf16* __restrict__ dx16, int M, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = lane_id();
  const float mean = mean_i[row], rstd = rstd_i[row];
  f32x4 xh[kMaxChunks], dh[kMaxChunks];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxChunks; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
      const f32x4 xv = *(const f32x4*)(x + (size_t)row * C + c);
      const f32x4 dv = *(const f32x4*)(dy + (size_t)row * C + c);
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

// ---- LayerNorm kernels with a fused rank-20 row projection -----------------------------------------------------------------
// GAViKO's GPA branch projects the very rows these kernels already hold in registers (gaviko.py:155-156: proj_down of the
// post-attention stream; its autograd: dcomb = dG . W_up).  As separate launches those projections re-read 12.7 MB and sit at
// the head of the latency-critical side chains; fused, they cost one LDS-staged weight (61 KB) and 240 FMAs per lane and row.
// One workgroup = 4 waves x 4 rows.  The weight sits in LDS as Ws[l][c] so a lane reads the float4 of its own 4 columns.
constexpr int kPL = 20;            // latent width handled by the fused projection
constexpr int kProjRows = 16;      // rows per workgroup
constexpr int kRedStride = 33;

struct RowProj {
  const float* w; const float* bias; float* y; float* z;
  int w_layout;                    // 0: w [L][C] (nn.Linear(C->L).weight)   1: w [C][L] (an up-projection's weight, transposed use)
  int act;                         // 0 none, 1 QuickGELU (z = pre-activation, may be NULL)
};

__device__ __forceinline__ void stage_proj_weight(float* Ws, const RowProj& pj, int C) {
  const int n4 = kPL * C / 4;
  if (pj.w_layout == 0) {
    for (int i = threadIdx.x; i < n4; i += blockDim.x) *(f32x4*)(Ws + 4 * i) = *(const f32x4*)(pj.w + 4 * i);
  } else {
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
      const int c = (4 * i) / kPL, l = (4 * i) % kPL;
      const f32x4 v = *(const f32x4*)(pj.w + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) Ws[(l + e) * C + c] = v[e];
    }
  }
}

// y[row][0..L) = act(v . W^T + b) for the two rows v0, v1 (lane holds columns k*256 + 4*lane .. +3 of each)
__device__ __forceinline__ void project_two_rows(const f32x4 (&v0)[kMaxChunks], const f32x4 (&v1)[kMaxChunks], const float* Ws, float* red,
                                                 const RowProj& pj, int C, int lane, int row0, int row1, int M) {
  float a0[kPL], a1[kPL];
#pragma unroll
  for (int l = 0; l < kPL; ++l) { a0[l] = 0.f; a1[l] = 0.f; }
#pragma unroll
  for (int k = 0; k < kMaxChunks; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
#pragma unroll
      for (int l = 0; l < kPL; ++l) {
        const f32x4 w = *(const f32x4*)(Ws + l * C + c);
        a0[l] += v0[k][0] * w[0] + v0[k][1] * w[1] + v0[k][2] * w[2] + v0[k][3] * w[3];
        a1[l] += v1[k][0] * w[0] + v1[k][1] * w[1] + v1[k][2] * w[2] + v1[k][3] * w[3];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = r ? row1 : row0;
#pragma unroll
    for (int l = 0; l < kPL; ++l) {
      float pr = r ? a1[l] : a0[l];
      pr += __shfl_xor(pr, 1, 64);
      if (!(lane & 1)) red[l * kRedStride + (lane >> 1)] = pr;
    }
    if (lane < kPL) {                                   // same-wave LDS traffic is in order: no barrier needed
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) sum += red[lane * kRedStride + i];
      if (pj.bias) sum += pj.bias[lane];
      if (row < M) {
        if (pj.z) pj.z[(size_t)row * kPL + lane] = sum;
        pj.y[(size_t)row * kPL + lane] = pj.act == 1 ? quick_gelu(sum) : sum;
      }
    }
  }
}

__global__ __launch_bounds__(256) void ln_fwd_proj_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, bf16* __restrict__ y16, float* __restrict__ mean_o,
                                                          float* __restrict__ rstd_o, int M, int C, float eps, RowProj pj) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  float* Ws = lsm;
  const int lane = lane_id(), wave = wave_id();
  float* red = lsm + kPL * C + wave * kPL * kRedStride;
  const int m0 = blockIdx.x * kProjRows + wave * 4;
  f32x4 v[4][kMaxChunks];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(m0 + r, M - 1);
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k) {
      const int c = k * 256 + lane * 4;
      v[r][k] = (c < C) ? *(const f32x4*)(x + (size_t)row * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  stage_proj_weight(Ws, pj, C);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + r;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k) s += v[r][k][0] + v[r][k][1] + v[r][k][2] + v[r][k][3];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[r][k][e] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (row < M) {
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
          bf16x4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = (bf16)((v[r][k][e] - mean) * rstd * g[e] + b[e]);
          *(bf16x4*)(y16 + (size_t)row * C + c) = h;
        }
      }
    }
  }
  __syncthreads();                                     // Ws staged by all four waves
  project_two_rows(v[0], v[1], Ws, red, pj, C, lane, m0, m0 + 1, M);
  project_two_rows(v[2], v[3], Ws, red, pj, C, lane, m0 + 2, m0 + 3, M);
}

__global__ __launch_bounds__(256) void ln_bwd_proj_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                          const float* __restrict__ gamma, const float* __restrict__ dres,
                                                          float* __restrict__ dx, bf16* __restrict__ dx16, int M, int C, RowProj pj) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  float* Ws = lsm;
  const int lane = lane_id(), wave = wave_id();
  float* red = lsm + kPL * C + wave * kPL * kRedStride;
  const int m0 = blockIdx.x * kProjRows + wave * 4;
  f32x4 xh[4][kMaxChunks], dh[4][kMaxChunks];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(m0 + r, M - 1);
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k) {
      const int c = k * 256 + lane * 4;
      xh[r][k] = (c < C) ? *(const f32x4*)(x + (size_t)row * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      dh[r][k] = (c < C) ? *(const f32x4*)(dy + (size_t)row * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  stage_proj_weight(Ws, pj, C);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(m0 + r, M - 1);
    const float mean = mean_i[row], rstd = rstd_i[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
        const f32x4 g = *(const f32x4*)(gamma + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[r][k][e] = (xh[r][k][e] - mean) * rstd;
          dh[r][k][e] = dh[r][k][e] * g[e];
          s1 += dh[r][k][e];
          s2 += dh[r][k][e] * xh[r][k][e];
        }
      }
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (dh[r][k][e] - m1 - xh[r][k][e] * m2);
        if (dres) {
          const f32x4 rr = *(const f32x4*)(dres + (size_t)row * C + c);
          o += rr;
        }
        dh[r][k] = o;                                  // keep dx for the projection
        if (m0 + r < M) {
          *(f32x4*)(dx + (size_t)row * C + c) = o;
          if (dx16) {
            bf16x4 h = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
            *(bf16x4*)(dx16 + (size_t)row * C + c) = h;
          }
        }
      }
    }
  }
  __syncthreads();
  project_two_rows(dh[0], dh[1], Ws, red, pj, C, lane, m0, m0 + 1, M);
  project_two_rows(dh[2], dh[3], Ws, red, pj, C, lane, m0 + 2, m0 + 3, M);
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

extern "C" int gvk_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                 const float* dres, float* dx, void* dx_bf16, int M, int C, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(dy && x && mean && rstd && gamma && dx, "gvk_layernorm_bwd: null pointer");
  GVK_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 256 * kMaxChunks, "gvk_layernorm_bwd: C=%d must be a multiple of 4 and <= 1024", C);
  GVK_LAUNCH(ln_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, dres, dx,
                     (bf16*)dx_bf16, M, C);
  return check_launch("layernorm_bwd");
}

namespace gvk {
static int check_proj(const gvk_rowproj_desc* pj, int C, const char* who) {
  GVK_REQUIRE(pj && pj->w && pj->y, "%s: null projection operand", who);
  GVK_REQUIRE(pj->L == kPL, "%s: the fused projection is built for L=%d only (got %d): use gvk_skinny_down", who, kPL, pj->L);
  GVK_REQUIRE(C % 4 == 0 && (kPL * C) % 4 == 0 && C <= 256 * kMaxChunks, "%s: C=%d must be a multiple of 4 and <= 1024", who, C);
  GVK_REQUIRE(pj->w_layout == 0 || pj->w_layout == 1, "%s: w_layout must be 0 or 1", who);
  GVK_REQUIRE(pj->act == 0 || pj->act == 1, "%s: act must be 0 (none) or 1 (QuickGELU)", who);
  return 0;
}
static size_t proj_lds_bytes(int C) { return (size_t)(kPL * C + 4 * kPL * kRedStride) * sizeof(float); }
template <typename K> static int set_lds(K kernel, size_t bytes, const char* who) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(%s): %s", who, hipGetErrorString(e));
  return 0;
}
}  // namespace gvk

extern "C" int gvk_layernorm_fwd_proj(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd,
                                      int M, int C, float eps, const gvk_rowproj_desc* proj, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x && gamma && beta && y_bf16, "gvk_layernorm_fwd_proj: null pointer");
  GVK_REQUIRE(M > 0 && C > 0, "gvk_layernorm_fwd_proj: empty shape");
  if (int rc = check_proj(proj, C, "gvk_layernorm_fwd_proj")) return rc;
  const size_t lds = proj_lds_bytes(C);
  static size_t attr = 0;
  if (lds > attr) {
    if (int rc = set_lds(&ln_fwd_proj_kernel, lds, "ln_fwd_proj")) return rc;
    attr = lds;
  }
  RowProj pj{proj->w, proj->bias, proj->y, proj->z, proj->w_layout, proj->act};
  GVK_LAUNCH(ln_fwd_proj_kernel, dim3((M + kProjRows - 1) / kProjRows), dim3(256), (unsigned)lds, (hipStream_t)stream, x, gamma, beta,
             (bf16*)y_bf16, mean, rstd, M, C, eps, pj);
  return check_launch("layernorm_fwd_proj");
}

extern "C" int gvk_layernorm_bwd_proj(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                      const float* dres, float* dx, void* dx_bf16, int M, int C, const gvk_rowproj_desc* proj,
                                      void* stream) {
  using namespace gvk;
  GVK_REQUIRE(dy && x && mean && rstd && gamma && dx, "gvk_layernorm_bwd_proj: null pointer");
  GVK_REQUIRE(M > 0 && C > 0, "gvk_layernorm_bwd_proj: empty shape");
  if (int rc = check_proj(proj, C, "gvk_layernorm_bwd_proj")) return rc;
  const size_t lds = proj_lds_bytes(C);
  static size_t attr = 0;
  if (lds > attr) {
    if (int rc = set_lds(&ln_bwd_proj_kernel, lds, "ln_bwd_proj")) return rc;
    attr = lds;
  }
  RowProj pj{proj->w, proj->bias, proj->y, proj->z, proj->w_layout, proj->act};
  GVK_LAUNCH(ln_bwd_proj_kernel, dim3((M + kProjRows - 1) / kProjRows), dim3(256), (unsigned)lds, (hipStream_t)stream, dy, x, mean, rstd,
             gamma, dres, dx, (bf16*)dx_bf16, M, C, pj);
  return check_launch("layernorm_bwd_proj");
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
