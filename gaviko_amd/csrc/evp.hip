// EVP (model/evp.py, `--method evp`): the pieces that are not already ViT / rank-L kernels.
//   evp_highpass   PromptGenerator.fft (evp.py:126-147) AS IT EXECUTES on a [B,1,D,H,W] volume: fft2/ifft2 run over (H, W) but the
//                  mask is indexed on (D, H) and every axis is fftshift-ed, so the filter is: for a fixed subset of depth slices, zero a
//                  band of H-frequencies for every W-frequency, elsewhere nothing -- i.e. out[b,d] = | Hp . X[b,d] | with one real H x H
//                  matrix Hp = I - Re(F^-1 diag(band) F) on the filtered slices and |X[b,d]| on the others.  No FFT is needed: one
//                  160x160x160 real product per filtered slice (the host builds Hp and the slice mask in float64).
//   pad2d / add2d / gelu_fwd / gelu_bwd / rows_patch / rows_gather: small glue for the rank-(dim/32) prompt latents, which are zero-padded to
//                  a width the row kernels are built for (6 -> 8, 24, 32).
#include "common.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

// out[s][i][j] = | sum_k hp[i][k] * x[s][k][j] |  (filtered slice)  or  | x[s][i][j] |  -- 32x32 output tile per workgroup, 2x2 per thread
__global__ __launch_bounds__(256) void evp_highpass_kernel(const float* __restrict__ x, const float* __restrict__ hp, const int* __restrict__ dmask,
                                                           float* __restrict__ out, int D, int H, int W) {
  __shared__ float sA[32][33], sX[32][33];
  const int s = blockIdx.z, d = s % D;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const float* xs = x + (size_t)s * H * W;
  float* os = out + (size_t)s * H * W;
  if (!dmask[d]) {                                          // slice-uniform
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int i = i0 + ty * 2 + a, j = j0 + tx * 2 + b;
        if (i < H && j < W) os[(size_t)i * W + j] = fabsf(xs[(size_t)i * W + j]);
      }
    return;
  }
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  for (int k0 = 0; k0 < H; k0 += 32) {
    for (int t = threadIdx.x; t < 32 * 32; t += 256) {
      const int r = t >> 5, c = t & 31;
      sA[r][c] = (i0 + r < H && k0 + c < H) ? hp[(size_t)(i0 + r) * H + k0 + c] : 0.f;
      sX[r][c] = (k0 + r < H && j0 + c < W) ? xs[(size_t)(k0 + r) * W + j0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const float a0 = sA[ty * 2][k], a1 = sA[ty * 2 + 1][k], b0 = sX[k][tx * 2], b1 = sX[k][tx * 2 + 1];
      acc[0][0] = __builtin_fmaf(a0, b0, acc[0][0]); acc[0][1] = __builtin_fmaf(a0, b1, acc[0][1]);
      acc[1][0] = __builtin_fmaf(a1, b0, acc[1][0]); acc[1][1] = __builtin_fmaf(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int i = i0 + ty * 2 + a, j = j0 + tx * 2 + b;
      if (i < H && j < W) os[(size_t)i * W + j] = fabsf(acc[a][b]);
    }
}

// dst[i][j] (drows x dcols, ld_dst) = src[i][j] (or src[j][i] when transpose) inside rows x cols, else 0
__global__ __launch_bounds__(256) void pad2d_kernel(const float* __restrict__ src, int ld_src, int rows, int cols, int transpose,
                                                    float* __restrict__ dst, int ld_dst, int drows, int dcols) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)drows * dcols) return;
  const int i = (int)(idx / dcols), j = (int)(idx - (long)i * dcols);
  float v = 0.f;
  if (!transpose) { if (i < rows && j < cols) v = src[(size_t)i * ld_src + j]; }
  else { if (j < rows && i < cols) v = src[(size_t)j * ld_src + i]; }
  dst[(size_t)i * ld_dst + j] = v;
}
__global__ __launch_bounds__(256) void add2d_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, float* __restrict__ out,
                                                    int ldo, int rows, int cols) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)rows * cols) return;
  const int i = (int)(idx / cols), j = (int)(idx - (long)i * cols);
  out[(size_t)i * ldo + j] = a[(size_t)i * lda + j] + b[(size_t)i * ldb + j];
}
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = gelu_erf(x[i]);
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx[i] = dy[i] * gelu_erf_grad(x[i]);
}
// tok[b][row_off + n][:] (=  or +=) src[b*N + n][:] (+ pos[n][:])
__global__ __launch_bounds__(256) void rows_patch_kernel(float* __restrict__ tok, const float* __restrict__ src, const float* __restrict__ pos, int B,
                                                         int T, int N, int C, int row_off, int accumulate) {
  const int c4 = C / 4;
  const long total = (long)B * N * c4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4) * 4;
    const long m = i / c4;
    const int b = (int)(m / N), n = (int)(m - (long)b * N);
    f32x4 v = *(const f32x4*)(src + m * C + c);
    if (pos != nullptr) v += *(const f32x4*)(pos + (size_t)n * C + c);
    float* d = tok + ((size_t)b * T + row_off + n) * C + c;
    if (accumulate) v += *(const f32x4*)d;
    *(f32x4*)d = v;
  }
}
__global__ __launch_bounds__(256) void rows_gather_kernel(const float* __restrict__ tok, float* __restrict__ dst, int B, int T, int N, int C, int row_off) {
  const int c4 = C / 4;
  const long total = (long)B * N * c4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4) * 4;
    const long m = i / c4;
    const int b = (int)(m / N), n = (int)(m - (long)b * N);
    *(f32x4*)(dst + m * C + c) = *(const f32x4*)(tok + ((size_t)b * T + row_off + n) * C + c);
  }
}

static unsigned blocks_for(long n, long cap = 4096) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace gvk

extern "C" int gvk_evp_highpass(const float* img, const float* hp, const int32_t* depth_mask, float* out, int B, int D, int H, int W, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(img && hp && depth_mask && out && B > 0 && D > 0 && H > 0 && W > 0, "gvk_evp_highpass: bad arguments");
  GVK_LAUNCH(evp_highpass_kernel, dim3((W + 31) / 32, (H + 31) / 32, B * D), dim3(256), 0, (hipStream_t)stream, img, hp, (const int*)depth_mask, out, D, H, W);
  return check_launch("evp_highpass");
}
extern "C" int gvk_pad2d_f32(const float* src, int ld_src, int rows, int cols, int transpose, float* dst, int ld_dst, int drows, int dcols, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(src && dst && rows > 0 && cols > 0 && drows > 0 && dcols > 0 && ld_src >= cols && ld_dst >= dcols, "gvk_pad2d_f32: bad arguments");
  GVK_REQUIRE(transpose ? (drows >= cols && dcols >= rows) : (drows >= rows && dcols >= cols), "gvk_pad2d_f32: destination smaller than the source");
  GVK_LAUNCH(pad2d_kernel, dim3(blocks_for((long)drows * dcols, 1L << 30)), dim3(256), 0, (hipStream_t)stream, src, ld_src, rows, cols, transpose, dst, ld_dst, drows,
             dcols);
  return check_launch("pad2d_f32");
}
extern "C" int gvk_add2d_f32(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int rows, int cols, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(a && b && out && rows > 0 && cols > 0 && lda >= cols && ldb >= cols && ldo >= cols, "gvk_add2d_f32: bad arguments");
  GVK_LAUNCH(add2d_kernel, dim3(blocks_for((long)rows * cols, 1L << 30)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out, ldo, rows, cols);
  return check_launch("add2d_f32");
}
extern "C" int gvk_gelu_fwd_f32(const float* x, float* y, int64_t n, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x && y && n > 0, "gvk_gelu_fwd_f32: bad arguments");
  GVK_LAUNCH(gelu_fwd_kernel, dim3(blocks_for(n, 1L << 30)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n);
  return check_launch("gelu_fwd_f32");
}
extern "C" int gvk_gelu_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(dy && x && dx && n > 0, "gvk_gelu_bwd_f32: bad arguments");
  GVK_LAUNCH(gelu_bwd_kernel, dim3(blocks_for(n, 1L << 30)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, (long)n);
  return check_launch("gelu_bwd_f32");
}
extern "C" int gvk_rows_patch(float* tok, const float* src, const float* pos, int B, int T, int N, int C, int row_off, int accumulate, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(tok && src && B > 0 && N > 0 && C % 4 == 0 && row_off >= 0 && row_off + N <= T, "gvk_rows_patch: bad arguments");
  GVK_LAUNCH(rows_patch_kernel, dim3(blocks_for((long)B * N * (C / 4))), dim3(256), 0, (hipStream_t)stream, tok, src, pos, B, T, N, C, row_off, accumulate);
  return check_launch("rows_patch");
}
extern "C" int gvk_rows_gather(const float* tok, float* dst, int B, int T, int N, int C, int row_off, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(tok && dst && B > 0 && N > 0 && C % 4 == 0 && row_off >= 0 && row_off + N <= T, "gvk_rows_gather: bad arguments");
  GVK_LAUNCH(rows_gather_kernel, dim3(blocks_for((long)B * N * (C / 4))), dim3(256), 0, (hipStream_t)stream, tok, dst, B, T, N, C, row_off);
  return check_launch("rows_gather");
}
