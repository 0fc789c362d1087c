// Casts / layout kernels (HBM-bound): fp32 -> bf16, transposing cast for dgrad weights, patch im2col.
#include "common.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out, int64_t n) {
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (; i + 3 < n; i += stride) {
    const f32x4 v = *(const f32x4*)(in + i);
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *(bf16x4*)(out + i) = o;
  }
  if (i < n && i + 3 >= n)
    for (int64_t j = i; j < n; ++j) out[j] = (bf16)in[j];
}

// Split-bf16 packing of a narrow fp32 operand into spare K columns of a bf16 GEMM operand, so that a rank-L fp32 product rides a bf16
// MFMA GEMM at (nearly) fp32 accuracy: x = hi + lo with hi = bf16(x), lo = bf16(x - hi), and a.w ~ a_hi.w_hi + a_lo.w_hi + a_hi.w_lo
// (the dropped lo.lo term is 2^-16 relative).  Activation side (weight_side = 0): columns [a_hi | a_lo | a_hi]; weight side:
// [w_hi | w_hi | w_lo | b_hi | b_lo], the bias riding two constant-1 columns of the activation operand.  3 ca + 2 columns in all.
__global__ __launch_bounds__(256) void pack_split_bf16_kernel(const float* __restrict__ a, int ca, const float* __restrict__ b,
                                                              bf16* __restrict__ dst, int ldd, int col0, int rows, int weight_side) {
  const int w = ca + (weight_side ? 1 : 0);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * w) return;
  const int r = i / w, c = i - r * w;
  bf16* d = dst + (size_t)r * ldd + col0;
  if (c < ca) {
    const float v = a[(size_t)r * ca + c];
    const bf16 hi = (bf16)v, lo = (bf16)(v - (float)hi);
    d[c] = hi;
    d[ca + c] = weight_side ? hi : lo;
    d[2 * ca + c] = weight_side ? lo : hi;
  } else {
    const float v = b != nullptr ? b[r] : 0.f;
    const bf16 hi = (bf16)v;
    d[3 * ca] = hi;
    d[3 * ca + 1] = (bf16)(v - (float)hi);
  }
}

// out[c][r] = in[r][c]; 64x64 tile through LDS (+1 pad), coalesced both sides.
template <typename OUT, typename IN = float>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const IN* __restrict__ in, OUT* __restrict__ out, int rows, int cols) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? (float)in[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) out[(size_t)c * rows + r] = (OUT)tile[tx][i];
  }
}

// One thread = 4 consecutive kw of one (patch, kd, kh): a float4 read, an 8-byte bf16x4 write.
// out row = patch index (b, d, h, w), column = (kd*ph + kh)*pw + kw.
template <typename OUT>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, OUT* __restrict__ out, int B, int D, int H,
                                                       int W, int pd, int ph, int pw) {
  const int nd = D / pd, nh = H / ph, nw = W / pw;
  const int K = pd * ph * pw, kq = K / 4;
  const int64_t total = (int64_t)B * nd * nh * nw * kq;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    // order the work so that consecutive threads walk consecutive image addresses: (b, z, y, x4) over the volume
    const int wq = W / 4;
    int64_t t = idx;
    const int x4 = t % wq; t /= wq;
    const int y = t % H; t /= H;
    const int z = t % D; const int b = t / D;
    const int x = x4 * 4;
    const f32x4 v = *(const f32x4*)(img + (((int64_t)b * D + z) * H + y) * W + x);
    const int d = z / pd, kd = z - d * pd, h = y / ph, kh = y - h * ph, w = x / pw, kw = x - w * pw;
    const int64_t row = (((int64_t)b * nd + d) * nh + h) * nw + w;
    OUT* dst = out + row * K + (kd * ph + kh) * pw + kw;
    if constexpr (sizeof(OUT) == 2) {
      bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
      *(bf16x4*)dst = o;
    } else {
      *(f32x4*)dst = v;
    }
  }
}

// out[n][k] = w[n][k] + s * sum_j B[n'][j] A[j][k] on the q rows (n < C) and the v rows (n >= 2C)
__global__ __launch_bounds__(256) void lora_merge_kernel(const float* __restrict__ w, const float* __restrict__ a_q, const float* __restrict__ b_q,
                                                         const float* __restrict__ a_v, const float* __restrict__ b_v, float* __restrict__ out,
                                                         int C, int r, float s) {
  const int64_t total = (int64_t)3 * C * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = i / C, k = i - (int64_t)n * C;
    float v = w[i];
    if (n < C || n >= 2 * C) {
      const float* A = n < C ? a_q : a_v;
      const float* Bm = n < C ? b_q + (size_t)n * r : b_v + (size_t)(n - 2 * C) * r;
      float d = 0.f;
      for (int j = 0; j < r; ++j) d += Bm[j] * A[(size_t)j * C + k];
      v += s * d;
    }
    out[i] = v;
  }
}

}  // namespace gvk

extern "C" int gvk_lora_merge_f32(const float* w, const float* a_q, const float* b_q, const float* a_v, const float* b_v, float* out, int C, int r,
                                  float s, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(w && a_q && b_q && a_v && b_v && out && C > 0 && r > 0, "gvk_lora_merge_f32: bad arguments");
  GVK_LAUNCH(lora_merge_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, w, a_q, b_q, a_v, b_v, out, C, r, s);
  return check_launch("lora_merge_f32");
}

extern "C" int gvk_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(in && out && n >= 0, "gvk_cast_f32_bf16: null pointer");
  if (n == 0) return 0;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  GVK_LAUNCH(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, (bf16*)out, n);
  return check_launch("cast_f32_bf16");
}

extern "C" int gvk_pack_split_bf16(const float* a, int ca, const float* b, void* dst, int ld_dst, int col0, int rows, int weight_side,
                                   void* stream) {
  using namespace gvk;
  GVK_REQUIRE(a && dst && ca > 0 && rows > 0 && col0 >= 0 && col0 + 3 * ca + 2 <= ld_dst, "gvk_pack_split_bf16: bad arguments");
  const long n = (long)rows * (ca + (weight_side ? 1 : 0));
  GVK_LAUNCH(pack_split_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, ca, b, (bf16*)dst, ld_dst, col0, rows,
             weight_side);
  return check_launch("pack_split_bf16");
}

extern "C" int gvk_transpose_cast_f32_bf16(const float* in, void* out, int rows, int cols, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(in && out && rows > 0 && cols > 0, "gvk_transpose_cast_f32_bf16: bad arguments");
  GVK_LAUNCH(transpose_cast_kernel<bf16>, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, in,
                     (bf16*)out, rows, cols);
  return check_launch("transpose_cast_f32_bf16");
}

// ---- fp32 forms for the fp32 compute path (gemm_f32.hip / attention_f32.hip)
extern "C" int gvk_transpose_f32(const float* in, float* out, int rows, int cols, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(in && out && rows > 0 && cols > 0, "gvk_transpose_f32: bad arguments");
  GVK_LAUNCH(transpose_cast_kernel<float>, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, in, out, rows, cols);
  return check_launch("transpose_f32");
}

extern "C" int gvk_transpose_bf16(const void* in, void* out, int rows, int cols, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(in && out && rows > 0 && cols > 0, "gvk_transpose_bf16: bad arguments");
  GVK_LAUNCH((transpose_cast_kernel<bf16, bf16>), dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, (bf16*)out,
             rows, cols);
  return check_launch("transpose_bf16");
}

extern "C" int gvk_patchify_f32(const float* img, float* out, int B, int D, int H, int W, int pd, int ph, int pw, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(img && out && B > 0, "gvk_patchify_f32: null pointer");
  GVK_REQUIRE(D % pd == 0 && H % ph == 0 && W % pw == 0 && pw % 4 == 0 && W % 4 == 0,
              "gvk_patchify_f32: volume %dx%dx%d not divisible by patch %dx%dx%d (pw must be a multiple of 4)", D, H, W, pd, ph, pw);
  const int64_t total = (int64_t)B * D * H * (W / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  GVK_LAUNCH(patchify_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img, out, B, D, H, W, pd, ph, pw);
  return check_launch("patchify_f32");
}

namespace gvk { __global__ void copy_u32_kernel(unsigned* dst, const unsigned* src, long n4, long n); }   // runtime.hip

extern "C" int gvk_copy_async(void* dst, const void* src, size_t bytes, void* stream) {
  using namespace gvk;
  GVK_REQUIRE((dst && src) || bytes == 0, "gvk_copy_async: null pointer");
  if (bytes == 0) return 0;
  GVK_REQUIRE((((uintptr_t)dst | (uintptr_t)src) & 15) == 0 && bytes % 4 == 0, "gvk_copy_async: pointers must be 16-byte aligned and the size a multiple of 4");
  const long n = (long)(bytes / 4), n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  GVK_LAUNCH(copy_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned*)dst, (const unsigned*)src, n4, n);
  return check_launch("copy_async");
}

extern "C" int gvk_patchify_bf16(const float* img, void* out, int B, int D, int H, int W, int pd, int ph, int pw, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(img && out && B > 0, "gvk_patchify_bf16: null pointer");
  GVK_REQUIRE(D % pd == 0 && H % ph == 0 && W % pw == 0 && pw % 4 == 0 && W % 4 == 0,
              "gvk_patchify_bf16: volume %dx%dx%d not divisible by patch %dx%dx%d (pw must be a multiple of 4)", D, H, W, pd, ph, pw);
  const int64_t total = (int64_t)B * D * H * (W / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  GVK_LAUNCH(patchify_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img, (bf16*)out, B, D, H, W, pd, ph, pw);
  return check_launch("patchify_bf16");
}
