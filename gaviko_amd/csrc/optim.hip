// Fused optimisation step around the hot path: gradient-norm clipping + Adam over every trainable
// tensor in ONE launch pair, replacing train.py:315-319 (clip_grad_norm_ = ~300 tiny norm kernels + a host sync,
// torch.optim.Adam.step = a foreach chain over 304 tensors).  The gradients already live in the engine's flat fp32 buffer
// (p.grad are views of it); exp_avg / exp_avg_sq use the same flat layout; the parameters stay ordinary separate tensors and are
// reached through a pointer table, so nothing about the model's storage changes.
//   gvk_sumsq      two-stage deterministic sum of squares of the flat gradient -> device scalar (no host sync)
//   gvk_adam_step  clip = min(1, max_norm / (sqrt(sumsq) + 1e-6)) (torch.nn.utils.clip_grad_norm_), g *= clip (written back, as
//                  torch does), m = b1 m + (1-b1) g, v = b2 v + (1-b2) g^2, p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// lr and beta1 of the step come from the host-side OneCycleLR mirror (gaviko_amd/optim.py); both are plain kernel arguments.
#include "common.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

constexpr int kSumsqBlocks = 256;
constexpr int kAdamBlockElems = 1024;     // 256 threads x float4

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ scratch) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 3 < n) {
      const f32x4 v = *(const f32x4*)(x + i);
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    } else {
      for (long j = i; j < n; ++j) s += x[j] * x[j];
    }
  }
  s = wave_sum(s);
  if (lane_id() == 0) red[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) scratch[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ scratch, float* __restrict__ out) {
  __shared__ float red[4];
  float s = wave_sum(scratch[threadIdx.x]);
  if (lane_id() == 0) red[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamArgs {
  const unsigned long long* ptr_tab;   // [ntensors] parameter data pointers
  const int* blk_tab;                  // [nblocks][4]: tensor id, offset inside the tensor, offset into the flat buffers, count
  float* grad; float* m; float* v;
  const float* norm_sq;
  float lr, beta1, beta2, eps, bias_c1, bias_c2, max_norm;
};

__global__ __launch_bounds__(256) void adam_step_kernel(AdamArgs a) {
  const int* bt = a.blk_tab + 4 * blockIdx.x;
  const int tid = bt[0], toff = bt[1], foff = bt[2], cnt = bt[3];
  float* p = (float*)a.ptr_tab[tid] + toff;
  float clip = 1.f;
  if (a.norm_sq != nullptr) clip = fminf(1.f, a.max_norm / (sqrtf(a.norm_sq[0]) + 1e-6f));
  const float step = a.lr / a.bias_c1, rs2 = 1.f / sqrtf(a.bias_c2);
  for (int i = threadIdx.x; i < cnt; i += 256) {
    const float g = a.grad[foff + i] * clip;
    const float m = a.beta1 * a.m[foff + i] + (1.f - a.beta1) * g;
    const float v = a.beta2 * a.v[foff + i] + (1.f - a.beta2) * g * g;
    a.grad[foff + i] = g;
    a.m[foff + i] = m;
    a.v[foff + i] = v;
    p[i] -= step * (m / (sqrtf(v) * rs2 + a.eps));
  }
}

}  // namespace gvk

extern "C" int gvk_sumsq(const float* x, int64_t n, float* scratch, float* out, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x && scratch && out && n > 0, "gvk_sumsq: bad arguments (scratch: f32 [256])");
  GVK_REQUIRE(((uintptr_t)x & 15) == 0, "gvk_sumsq: x must be 16-byte aligned");
  GVK_LAUNCH(sumsq_partial_kernel, dim3(kSumsqBlocks), dim3(256), 0, (hipStream_t)stream, x, (long)n, scratch);
  int rc = check_launch("sumsq/partial");
  if (rc) return rc;
  GVK_LAUNCH(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, out);
  return check_launch("sumsq/final");
}

extern "C" int gvk_adam_step(const gvk_adam_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->ptr_tab && d->blk_tab && d->grad && d->m && d->v, "gvk_adam_step: null pointer");
  GVK_REQUIRE(d->nblocks > 0, "gvk_adam_step: nothing to update");
  GVK_REQUIRE(d->bias_c1 > 0.f && d->bias_c2 > 0.f && d->eps > 0.f, "gvk_adam_step: bias corrections and eps must be positive");
  AdamArgs a{(const unsigned long long*)d->ptr_tab, (const int*)d->blk_tab, d->grad, d->m, d->v, d->norm_sq,
             d->lr, d->beta1, d->beta2, d->eps, d->bias_c1, d->bias_c2, d->max_norm};
  GVK_LAUNCH(adam_step_kernel, dim3(d->nblocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("adam_step");
}
