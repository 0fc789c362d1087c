// nn.Dropout as a stand-alone pass over row-major activations / gradients, for the sites that have no producing kernel to fuse into:
// emb_dropout (vision_transformer.py:157), VPT's prompt_dropout (vpt.py:129,148,152) and the gradient side of the Linear+Dropout
// pairs (the dgrad GEMM operand and the bias / weight-gradient operand of vision_transformer.py:34,54 must carry the forward's mask).
#include "common.hpp"
#include "dropout.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

struct DropRowsArgs {
  const float* x; float* out32; bf16* out16;
  int M, N, ld, rows_in, rows_out, row_off;
  unsigned long long seed; const unsigned long long* seed_ptr; unsigned int thresh; float inv_keep;
};

// logical row m (mask index m*N + n) lives in buffer row (m / rows_in) * rows_out + row_off + m % rows_in when rows_in > 0
__global__ __launch_bounds__(256) void dropout_rows_kernel(DropRowsArgs p) {
  const unsigned long long sd = p.seed + (p.seed_ptr != nullptr ? *p.seed_ptr : 0ull);
  const int n4 = p.N >> 2;
  const long long total = (long long)p.M * n4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / n4), n = (int)(i - (long long)m * n4) * 4;
    size_t row = (size_t)m;
    if (p.rows_in > 0) {
      const int s = m / p.rows_in;
      row = (size_t)s * p.rows_out + p.row_off + (m - s * p.rows_in);
    }
    f32x4 v = *(const f32x4*)(p.x + row * p.ld + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= drop_scale(sd, (unsigned long long)m * p.N + n + e, p.thresh, p.inv_keep);
    if (p.out32 != nullptr) *(f32x4*)(p.out32 + row * p.ld + n) = v;
    if (p.out16 != nullptr) {
      bf16x4 h = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
      *(bf16x4*)(p.out16 + row * p.ld + n) = h;
    }
  }
}

}  // namespace gvk

extern "C" int gvk_dropout_rows(const gvk_dropout_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->x && (d->out32 || d->out16), "gvk_dropout_rows: null pointer");
  GVK_REQUIRE(d->M > 0 && d->N > 0 && d->N % 4 == 0 && d->ld >= d->N && d->ld % 4 == 0, "gvk_dropout_rows: N and ld must be multiples of 4, ld >= N");
  GVK_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "gvk_dropout_rows: drop_p in [0, 1)");
  GVK_REQUIRE(d->rows_in == 0 || (d->rows_in > 0 && d->rows_out >= d->rows_in + d->row_off && d->row_off >= 0), "gvk_dropout_rows: bad row mapping");
  DropRowsArgs a{d->x, d->out32, (bf16*)d->out16, d->M, d->N, d->ld, d->rows_in, d->rows_out, d->row_off, d->seed,
                 (const unsigned long long*)d->seed_ptr, drop_threshold_u32(d->drop_p), d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f};
  long long blocks = ((long long)d->M * (d->N / 4) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  GVK_LAUNCH(dropout_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("dropout_rows");
}
