// Token assembly and the pooled classification head.
//   rows_broadcast : G[b][row_off + r][:] = src[r][:] + add[r][:]      (cls_token + pos[0], prompts + prompt_pos;
//                    vision_transformer.py:154-156, gaviko.py:536-543, vpt.py:127-131,147-153)
//   rows_batch_sum : out[r][:] = sum_b dG[b][row_off + r][:]            (gradient of a batch-broadcast parameter)
//   head_fwd       : logits = Linear(mean_r LN(G[b][r0 .. r0+R)))       (final LN restricted to the pooled rows:
//                    gaviko.py:306,316 pools rows 0..P; vision_transformer.py:89,161 pools row 0 or all rows)
//   head_bwd       : dWh, dbh and the gradient into the pooled rows of the final residual stream (other rows: zero)
#include "common.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

__global__ __launch_bounds__(256) void rows_broadcast_kernel(float* __restrict__ out, const float* __restrict__ src, const float* __restrict__ add,
                                                             int B, int T, int row_off, int R, int C) {
  const int64_t total = (int64_t)B * R * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = i % C;
    const int64_t t = i / C;
    const int r = t % R, b = t / R;
    out[((int64_t)b * T + row_off + r) * C + c] = src[(int64_t)r * C + c] + (add ? add[(int64_t)r * C + c] : 0.f);
  }
}

__global__ __launch_bounds__(256) void rows_batch_sum_kernel(const float* __restrict__ dg, float* __restrict__ out, float* __restrict__ out2,
                                                             int B, int T, int row_off, int R, int C, int accumulate) {
  const int64_t total = (int64_t)R * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = i % C, r = i / C;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dg[((int64_t)b * T + row_off + r) * C + c];
    out[i] = accumulate ? out[i] + a : a;
    if (out2) out2[i] = accumulate ? out2[i] + a : a;
  }
}

struct HeadArgs {
  const float* g;        // final residual stream [B*T][C]
  const float* ln_g; const float* ln_b;
  const float* wh; const float* bh;   // [K][C], [K]
  float* logits;         // [B][K]
  float* pooled;         // [B][C] (saved)
  const float* dlogits;  // [B][K]
  float* dg;             // [B*T][C]: rows r0..r0+R of each sample are written
  float* dwh; float* dbh;
  int B, T, C, K, r0, R, accumulate;
};

// One workgroup per sample, SIXTEEN waves: the final LayerNorm of the R pooled rows (gaviko.py:306,316: prompts + CLS), their mean, the
// head.  A wave's rows are a chain of dependent round trips (load, two wave reductions); with four waves the 33 rows of GAViKO were nine
// such links, 22 us at the end of every forward -- sixteen waves make it three.
constexpr int kHeadWaves = 16;
__global__ __launch_bounds__(64 * kHeadWaves) void head_fwd_kernel(HeadArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* pool_s = (float*)smem;   // [kHeadWaves][C] then reduced into [0][C]
  const int b = blockIdx.x, lane = lane_id(), wave = wave_id(), C = p.C;
  f32x4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int r = wave; r < p.R; r += kHeadWaves) {
    const float* xr = p.g + ((size_t)b * p.T + p.r0 + r) * C;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = k * 256 + lane * 4;
      v[k] = (c < C) ? *(const f32x4*)(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + 1e-5f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
        const f32x4 g = *(const f32x4*)(p.ln_g + c);
        const f32x4 bb = *(const f32x4*)(p.ln_b + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k][e] += (v[k][e] - mean) * rstd * g[e] + bb[e];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = k * 256 + lane * 4;
    if (c < C) *(f32x4*)(pool_s + wave * C + c) = acc[k];
  }
  __syncthreads();
  const float invR = 1.f / (float)p.R;
  float tsum = 0.f;
  const int c0 = threadIdx.x;                              // C <= 1024 = the workgroup's threads: one column each
  if (c0 < C) {
#pragma unroll
    for (int w = 0; w < kHeadWaves; ++w) tsum += pool_s[w * C + c0];          // wave order: deterministic
    tsum *= invR;
  }
  __syncthreads();
  if (c0 < C) {
    pool_s[c0] = tsum;
    if (p.pooled) p.pooled[(size_t)b * C + c0] = tsum;
  }
  __syncthreads();
  for (int k = wave; k < p.K; k += kHeadWaves) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += pool_s[c] * p.wh[(size_t)k * C + c];
    a = wave_sum(a);
    if (lane == 0) p.logits[b * p.K + k] = a + p.bh[k];
  }
}

// grid (B, ceil(R / 4)), one pooled row per wave: dpooled = dlogits . Wh (recomputed per workgroup: K * C products); every pooled row gets
// dy = dpooled / R through the LN backward.  (One workgroup per sample walked its 33 rows nine deep at the very start of the backward.)
__global__ __launch_bounds__(256) void head_bwd_rows_kernel(HeadArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dp_s = (float*)smem;   // [C]
  const int b = blockIdx.x, lane = lane_id(), wave = wave_id(), C = p.C;
  const float invR = 1.f / (float)p.R;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int k = 0; k < p.K; ++k) a += p.dlogits[b * p.K + k] * p.wh[(size_t)k * C + c];
    dp_s[c] = a * invR;
  }
  __syncthreads();
  for (int r = blockIdx.y * 4 + wave; r < p.R; r += 4 * gridDim.y) {
    const size_t off = ((size_t)b * p.T + p.r0 + r) * C;
    f32x4 xh[4], dh[4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = k * 256 + lane * 4;
      xh[k] = (c < C) ? *(const f32x4*)(p.g + off + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += xh[k][0] + xh[k][1] + xh[k][2] + xh[k][3];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xh[k][e] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + 1e-5f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
        const f32x4 g = *(const f32x4*)(p.ln_g + c);
        const f32x4 dy = *(const f32x4*)(dp_s + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[k][e] = (xh[k][e] - mean) * rstd;
          dh[k][e] = dy[e] * g[e];
          s1 += dh[k][e];
          s2 += dh[k][e] * xh[k][e];
        }
      } else {
        dh[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (dh[k][e] - m1 - xh[k][e] * m2);
        *(f32x4*)(p.dg + off + c) = o;
      }
    }
  }
}

__global__ __launch_bounds__(256) void head_bwd_w_kernel(HeadArgs p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < p.K * p.C) {
    const int k = i / p.C, c = i - k * p.C;
    float a = 0.f;
    for (int b = 0; b < p.B; ++b) a += p.dlogits[b * p.K + k] * p.pooled[(size_t)b * p.C + c];
    p.dwh[i] = p.accumulate ? p.dwh[i] + a : a;
  }
  if (i < p.K) {
    float a = 0.f;
    for (int b = 0; b < p.B; ++b) a += p.dlogits[b * p.K + i];
    p.dbh[i] = p.accumulate ? p.dbh[i] + a : a;
  }
}

}  // namespace gvk

extern "C" int gvk_rows_broadcast(float* out, const float* src, const float* add, int B, int T, int row_off, int R, int C, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(out && src && B > 0 && R > 0 && C > 0 && row_off >= 0 && row_off + R <= T, "gvk_rows_broadcast: bad arguments");
  int64_t blocks = ((int64_t)B * R * C + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  GVK_LAUNCH(rows_broadcast_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, src, add, B, T, row_off, R, C);
  return check_launch("rows_broadcast");
}

extern "C" int gvk_rows_batch_sum(const float* dg, float* out, float* out2, int B, int T, int row_off, int R, int C, int accumulate, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(dg && out && B > 0 && R > 0 && C > 0 && row_off >= 0 && row_off + R <= T, "gvk_rows_batch_sum: bad arguments");
  int64_t blocks = ((int64_t)R * C + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  GVK_LAUNCH(rows_batch_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dg, out, out2, B, T, row_off, R, C, accumulate);
  return check_launch("rows_batch_sum");
}

static int head_fill(gvk::HeadArgs& a, const gvk_head_desc* d) {
  using namespace gvk;
  GVK_REQUIRE(d && d->g && d->ln_gamma && d->ln_beta && d->wh && d->bh, "gvk_head: null pointer");
  GVK_REQUIRE(d->B > 0 && d->C > 0 && d->C % 4 == 0 && d->C <= 1024 && d->K > 0 && d->R > 0 && d->r0 >= 0 && d->r0 + d->R <= d->T,
              "gvk_head: bad shape (C=%d must be a multiple of 4 and <= 1024; rows %d..%d of T=%d)", d->C, d->r0, d->r0 + d->R, d->T);
  a.g = d->g; a.ln_g = d->ln_gamma; a.ln_b = d->ln_beta; a.wh = d->wh; a.bh = d->bh; a.logits = d->logits; a.pooled = d->pooled;
  a.dlogits = d->dlogits; a.dg = d->dg; a.dwh = d->dwh; a.dbh = d->dbh;
  a.B = d->B; a.T = d->T; a.C = d->C; a.K = d->K; a.r0 = d->r0; a.R = d->R; a.accumulate = d->accumulate;
  return 0;
}

extern "C" int gvk_head_fwd(const gvk_head_desc* d, void* stream) {
  using namespace gvk;
  HeadArgs a{};
  int rc = head_fill(a, d);
  if (rc) return rc;
  GVK_REQUIRE(d->logits, "gvk_head_fwd: logits null");
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kHeadWaves * 1024 * 4);
    if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(head_fwd): %s", hipGetErrorString(e));
    attr = true;
  }
  GVK_LAUNCH(head_fwd_kernel, dim3(d->B), dim3(64 * kHeadWaves), kHeadWaves * d->C * 4, (hipStream_t)stream, a);
  return check_launch("head_fwd");
}

extern "C" int gvk_head_bwd(const gvk_head_desc* d, void* stream) {
  using namespace gvk;
  HeadArgs a{};
  int rc = head_fill(a, d);
  if (rc) return rc;
  GVK_REQUIRE(d->dlogits && d->pooled && d->dwh && d->dbh, "gvk_head_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (d->dg != nullptr) {
    GVK_LAUNCH(head_bwd_rows_kernel, dim3(d->B, (d->R + 3) / 4), dim3(256), d->C * 4, s, a);
    rc = check_launch("head_bwd_rows");
    if (rc) return rc;
  }
  GVK_LAUNCH(head_bwd_w_kernel, dim3((d->K * d->C + 255) / 256), dim3(256), 0, s, a);
  return check_launch("head_bwd_w");
}

// ---------------------------------------------------------------------------------------------------------------------
// VPT (model/vpt.py): prompt projection Linear(prompt_dim, C) on a handful of rows, and the per-layer token re-pack of
// deep VPT.  vpt.py:147-153 rebuilds the sequence before every deep layer as [cls | P new prompts | x[:, 1+skip:]] with
// skip = deep_prompt_embeddings[i].shape[1] = prompt_dim (NOT num_prompts -- reference quirk 16), so T shrinks by skip - P.
namespace gvk {

__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                               float* __restrict__ out, int R, int K, int C) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * C) return;
  const int r = idx / C, c = idx - r * C;
  float a = b ? b[c] : 0.f;
  for (int k = 0; k < K; ++k) a += x[r * K + k] * w[(size_t)c * K + k];
  out[idx] = a;
}
// dw[c][k] (+)= sum_r dout[r][c] x[r][k];  db[c] (+)= sum_r dout[r][c];  dx[r][k] = sum_c dout[r][c] w[c][k]
__global__ __launch_bounds__(256) void small_linear_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dout, float* __restrict__ dw,
                                                                 float* __restrict__ db, int R, int K, int C, int accumulate) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < C * K) {
    const int c = idx / K, k = idx - c * K;
    float a = 0.f;
    for (int r = 0; r < R; ++r) a += dout[(size_t)r * C + c] * x[r * K + k];
    dw[idx] = accumulate ? dw[idx] + a : a;
  }
  if (idx < C && db != nullptr) {
    float a = 0.f;
    for (int r = 0; r < R; ++r) a += dout[(size_t)r * C + idx];
    db[idx] = accumulate ? db[idx] + a : a;
  }
}
__global__ __launch_bounds__(64) void small_linear_bwd_x_kernel(const float* __restrict__ w, const float* __restrict__ dout, float* __restrict__ dx,
                                                                int R, int K, int C, int accumulate) {
  const int r = blockIdx.x / K, k = blockIdx.x - r * K;       // one wave per dx element
  float a = 0.f;
  for (int c = threadIdx.x; c < C; c += 64) a += dout[(size_t)r * C + c] * w[(size_t)c * K + k];
  a = wave_sum(a);
  if (threadIdx.x == 0) dx[blockIdx.x] = accumulate ? dx[blockIdx.x] + a : a;
}

// out[b][0] = in[b][0]; out[b][1+p] = prompt[p]; out[b][1+P+j] = in[b][1+skip+j]
__global__ __launch_bounds__(256) void vpt_repack_fwd_kernel(const float* __restrict__ in, const float* __restrict__ prompt, float* __restrict__ out,
                                                             int B, int Tin, int Tout, int P, int skip, int C) {
  const int c4 = C / 4;
  const int64_t total = (int64_t)B * Tout * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (i % c4) * 4;
    const int64_t t_ = i / c4;
    const int t = t_ % Tout, b = t_ / Tout;
    f32x4 v;
    if (t == 0) v = *(const f32x4*)(in + ((int64_t)b * Tin) * C + c);
    else if (t <= P) v = *(const f32x4*)(prompt + (int64_t)(t - 1) * C + c);
    else v = *(const f32x4*)(in + ((int64_t)b * Tin + (t - P + skip)) * C + c);
    *(f32x4*)(out + ((int64_t)b * Tout + t) * C + c) = v;
  }
}
// din[b][0] = dout[b][0]; din[b][1..skip] = 0; din[b][1+skip+j] = dout[b][1+P+j]
__global__ __launch_bounds__(256) void vpt_repack_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int Tin, int Tout, int P,
                                                             int skip, int C) {
  const int c4 = C / 4;
  const int64_t total = (int64_t)B * Tin * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (i % c4) * 4;
    const int64_t t_ = i / c4;
    const int t = t_ % Tin, b = t_ / Tin;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t == 0) v = *(const f32x4*)(dout + ((int64_t)b * Tout) * C + c);
    else if (t > skip) v = *(const f32x4*)(dout + ((int64_t)b * Tout + (t - skip + P)) * C + c);
    *(f32x4*)(din + ((int64_t)b * Tin + t) * C + c) = v;
  }
}

template <typename IN>
__global__ __launch_bounds__(256) void cast_bf16_f32_strided_kernel(const IN* __restrict__ in, float* __restrict__ out, int M, int C, int ld_in) {
  const int c4 = C / 4;
  const int64_t total = (int64_t)M * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (i % c4) * 4;
    const int64_t m = i / c4;
    if constexpr (sizeof(IN) == 2) {
      const bf16x4 v = *(const bf16x4*)(in + m * ld_in + c);
      *(f32x4*)(out + m * C + c) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    } else {
      *(f32x4*)(out + m * C + c) = *(const f32x4*)(in + m * ld_in + c);
    }
  }
}

}  // namespace gvk

extern "C" int gvk_small_linear_fwd(const float* x, const float* w, const float* b, float* out, int R, int K, int C, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x && w && out && R > 0 && K > 0 && C > 0, "gvk_small_linear_fwd: bad arguments");
  GVK_LAUNCH(small_linear_fwd_kernel, dim3((R * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, w, b, out, R, K, C);
  return check_launch("small_linear_fwd");
}
extern "C" int gvk_small_linear_bwd(const float* x, const float* w, const float* dout, float* dw, float* db, float* dx, int R, int K, int C,
                                    int accumulate, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x && w && dout && R > 0 && K > 0 && C > 0, "gvk_small_linear_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dw != nullptr) {
    const int n = C * K > C ? C * K : C;
    GVK_LAUNCH(small_linear_bwd_w_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, dout, dw, db, R, K, C, accumulate);
    int rc = check_launch("small_linear_bwd_w");
    if (rc) return rc;
  }
  if (dx != nullptr) {
    GVK_LAUNCH(small_linear_bwd_x_kernel, dim3(R * K), dim3(64), 0, s, w, dout, dx, R, K, C, accumulate);
    return check_launch("small_linear_bwd_x");
  }
  return 0;
}
extern "C" int gvk_vpt_repack_fwd(const float* in, const float* prompt, float* out, int B, int Tin, int Tout, int P, int skip, int C, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(in && prompt && out && B > 0 && C % 4 == 0, "gvk_vpt_repack_fwd: bad arguments");
  GVK_REQUIRE(Tout == Tin - skip + P && Tout > 1 + P && skip >= 0, "gvk_vpt_repack_fwd: Tout=%d must equal Tin-skip+P (%d-%d+%d)", Tout, Tin, skip, P);
  int64_t blocks = ((int64_t)B * Tout * (C / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  GVK_LAUNCH(vpt_repack_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, prompt, out, B, Tin, Tout, P, skip, C);
  return check_launch("vpt_repack_fwd");
}
extern "C" int gvk_vpt_repack_bwd(const float* dout, float* din, int B, int Tin, int Tout, int P, int skip, int C, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(dout && din && B > 0 && C % 4 == 0, "gvk_vpt_repack_bwd: bad arguments");
  GVK_REQUIRE(Tout == Tin - skip + P && skip >= 0, "gvk_vpt_repack_bwd: Tout=%d must equal Tin-skip+P", Tout);
  int64_t blocks = ((int64_t)B * Tin * (C / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  GVK_LAUNCH(vpt_repack_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, din, B, Tin, Tout, P, skip, C);
  return check_launch("vpt_repack_bwd");
}
extern "C" int gvk_cast_bf16_f32_strided(const void* in, float* out, int M, int C, int ld_in, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(in && out && M > 0 && C > 0 && C % 4 == 0 && ld_in % 4 == 0 && ld_in >= C, "gvk_cast_bf16_f32_strided: bad arguments");
  int64_t blocks = ((int64_t)M * (C / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  GVK_LAUNCH(cast_bf16_f32_strided_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, out, M, C, ld_in);
  return check_launch("cast_bf16_f32_strided");
}
extern "C" int gvk_copy_f32_strided(const float* in, float* out, int M, int C, int ld_in, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(in && out && M > 0 && C > 0 && C % 4 == 0 && ld_in % 4 == 0 && ld_in >= C, "gvk_copy_f32_strided: bad arguments");
  int64_t blocks = ((int64_t)M * (C / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  GVK_LAUNCH(cast_bf16_f32_strided_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, M, C, ld_in);
  return check_launch("copy_f32_strided");
}
