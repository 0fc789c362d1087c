// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <stdint.h>

namespace gvk {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define GVK_LDS __attribute__((address_space(3)))
#define GVK_GLOBAL __attribute__((address_space(1)))

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Attention-style grids: `nblk` row blocks of each of `ngrp` independent (batch, head) groups all stream the SAME operand of their group.
// Consecutive workgroups go to consecutive XCDs (8 private L2s), so with the plain (block, group) order the 9 blocks of a group land
// on 8 different L2s and the group's operand is fetched 8 times (measured: 115 MB per attention forward against 25 MB algorithmic).
// This maps a 1-D grid so that every block of a group runs on ONE XCD (groups dealt round-robin to XCDs).  Bijective for any shape.
__device__ __forceinline__ void xcd_group_block(int bid, int nblk, int ngrp, int& grp, int& blk) {
  const int per = ngrp >> 3, main = per * 8 * nblk;          // groups that fill whole rounds of 8 XCDs
  if (bid < main) {
    const int xcd = bid & 7, j = bid >> 3;
    grp = (j / nblk) * 8 + xcd;
    blk = j - (j / nblk) * nblk;
  } else {                                                     // the ngrp % 8 left-over groups: plain order
    const int r = bid - main;
    grp = per * 8 + r / nblk;
    blk = r - (r / nblk) * nblk;
  }
}

// async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const GVK_GLOBAL void*)gsrc, (GVK_LDS void*)lds_wave_base, 16, 0, 0);
}

// 4x16 transposed LDS read (ds_read_b64_tr_b16): within each 16-lane group, lane 4q+p supplies the address of
// row q, columns 4p..4p+3; lane i receives column i of the 4 rows (row q in element q).  EXEC must be all ones.
__device__ __forceinline__ bf16x4 lds_read_tr16(const void* p) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GVK_LDS s16x4*)p);
  return __builtin_bit_cast(bf16x4, v);
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): one v_rcp, one v_exp, 6 FMAs -- libdevice erff costs ~10x that and
// dominated the fused GELU epilogues.  The fp32 side paths that need the libm-grade value call erff directly.
// Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7), written with explicit FMAs (the library builds with -ffp-contract=off) and the
// exponential straight on v_exp_f32: exp(-a^2) = exp2(-a^2 * log2 e).
__device__ __forceinline__ float as_poly(float t) {
  float p = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  p = __builtin_fmaf(t, p, 1.421413741f);
  p = __builtin_fmaf(t, p, -0.284496736f);
  p = __builtin_fmaf(t, p, 0.254829592f);
  return t * p;
}
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  const float e = __builtin_amdgcn_exp2f(ax * ax * -1.44269504088896340736f);
  const float r = __builtin_fmaf(-as_poly(t), e, 1.0f);
  return copysignf(r, x);
}
// exact-erf GELU of nn.GELU() (vision_transformer.py:32); `fast` variants feed bf16 outputs only
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float gelu_fast(float x) {
  const float h = 0.5f * x;
  return __builtin_fmaf(h, erf_fast(x * 0.70710678118654752440f), h);
}
__device__ __forceinline__ float gelu_fast_grad(float x) {
  // Phi(x) + x phi(x); e = exp(-x^2/2) is shared by the erf tail and the density
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);
  const float erfv = copysignf(__builtin_fmaf(-as_poly(t), e, 1.0f), x);
  return __builtin_fmaf(x * 0.39894228040143267794f, e, __builtin_fmaf(0.5f, erfv, 0.5f));
}
// GELU(x) and GELU'(x) together: one reciprocal, one exponential, one polynomial (the forward GEMM epilogue that also leaves the derivative)
__device__ __forceinline__ void gelu_fast_both(float x, float& y, float& dy) {
  // erf exactly as gelu_fast / erf_fast take it (the activation must not depend on whether the derivative is wanted); its exponential
  // e = exp(-x^2 / 2) is the density's
  const float z = x * 0.70710678118654752440f, ax = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  const float e = __builtin_amdgcn_exp2f(ax * ax * -1.44269504088896340736f);
  const float erfv = copysignf(__builtin_fmaf(-as_poly(t), e, 1.0f), z);
  const float h = 0.5f * x;
  y = __builtin_fmaf(h, erfv, h);
  dy = __builtin_fmaf(x * 0.39894228040143267794f, e, __builtin_fmaf(0.5f, erfv, 0.5f));
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float quick_gelu(float x) { return x * sigmoidf_(1.702f * x); }
__device__ __forceinline__ float quick_gelu_grad(float x) {
  float s = sigmoidf_(1.702f * x);
  return s + 1.702f * x * s * (1.0f - s);
}

}  // namespace gvk

// A/B switches of the measurement builds.  The product library is compiled WITHOUT GVK_DIAG: every such switch folds to its measured-best
// constant there and the environment is never consulted.  `python -m gaviko_amd.build --diag` builds libgaviko_hip_diag.so with GVK_DIAG
// (plus the diagnostics of include/gaviko_hip_diag.h) for tools/.
#include <stdlib.h>
namespace gvk {
#ifdef GVK_DIAG
inline const char* diag_env(const char* name) { return getenv(name); }
#else
inline const char* diag_env(const char*) { return nullptr; }
#endif
}  // namespace gvk

// host-side error plumbing shared by every C-ABI entry point
extern "C" const char* gvk_last_error(void);
namespace gvk {
int set_error(int code, const char* fmt, ...);

// ---- launch plans (runtime.hip) -----------------------------------------------------------------------------------------
// Every kernel launch of the library goes through gvk::launch.  While a plan is being recorded on the calling thread the
// launch is also stored as a closure (kernel, grid, block, LDS bytes, stream, by-value arguments) so gvk_plan_replay can
// re-issue the whole step from one C loop: no Python, no validation, and -- unlike a captured hipGraph, whose executor
// serialises three forked branches on this runtime (tools/probe/probe_streams.hip) -- exact stream/event semantics.
bool plan_recording();
void plan_push(std::function<void()>&& node);
void plan_push_launch(hipStream_t stream, std::function<void()>&& node);   // same (the diag library can swap a stream's kernels for empty ones)

template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned lds, hipStream_t stream, Args... args) {
  if (plan_recording())
    plan_push_launch(stream, [=]() { hipLaunchKernelGGL(kernel, grid, block, lds, stream, static_cast<KArgs>(args)...); });
  hipLaunchKernelGGL(kernel, grid, block, lds, stream, static_cast<KArgs>(args)...);
}
#define GVK_LAUNCH(kernel, grid, block, lds, stream, ...) ::gvk::launch(kernel, grid, block, lds, stream, __VA_ARGS__)

int check_launch(const char* what);
}  // namespace gvk

#define GVK_REQUIRE(cond, ...)                                  \
  do {                                                          \
    if (!(cond)) return gvk::set_error(-2, __VA_ARGS__);        \
  } while (0)
