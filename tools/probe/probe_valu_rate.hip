// GPU probe: issue cost (cycles per wave instruction on one SIMD) of v_exp_f32, v_add_f32, v_cvt_pk_bf16_f32 and of an MFMA beside them,
// with one and with two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 probe_valu_rate.hip -o probe_valu_rate && ./probe_valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float seed) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + i * 0.01f + threadIdx.x * 1e-4f;
  f32x16 acc = {};
  bf16x8 fa, fb;
#pragma unroll
  for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(seed + j); fb[j] = (__bf16)(seed - j); }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = a[i] + seed;
    } else if constexpr (MODE == 2) {          // 4 MFMAs + 16 exp (independent)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
    } else if constexpr (MODE == 3) {          // 4 MFMAs alone
#pragma unroll
      for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    } else if constexpr (MODE == 4) {          // 4 MFMAs + 16 adds
#pragma unroll
      for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = a[i] + seed;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int per_iter, int lds_bytes) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 8 * 1024);
  const int iters = 2000;
  for (int wpc : {1, 2}) {
    // wpc workgroups of 4 waves per CU (one or two waves per SIMD): dynamic LDS keeps the count exact
    const int grid = 256 * wpc;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), wpc == 1 ? 100 * 1024 : 60 * 1024, 0, out, cyc, iters, 0.5f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), wpc == 1 ? 100 * 1024 : 60 * 1024, 0, out, cyc, iters, 0.5f);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, cyc, 8 * grid, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
    // s_memtime counts at 100 MHz on this part: report raw ticks per iteration and leave the conversion to the reader via MODE 3
    printf("%-28s %d wave(s)/SIMD: %8.2f ticks / iteration (%d instr)\n", name, wpc, s / grid / iters, per_iter);
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  run<3>("4 mfma 32x32x16", 4, 0);
  run<0>("16 v_exp_f32", 16, 0);
  run<1>("16 v_add_f32", 16, 0);
  run<2>("4 mfma + 16 v_exp_f32", 20, 0);
  run<4>("4 mfma + 16 v_add_f32", 20, 0);
  return 0;
}
