// fp32 GEMM, Y = A . W^T with the same fused epilogues as gemm_bf16.hip, for the fp32 configurations of the reference
// (BASELINE cfg4: adaptformer / melo, fp32, tolerance 1e-5 -- the bf16 operand path cannot meet that).
// v_mfma_f32_16x16x4_f32: exact fp32 products and fp32 accumulation at the fp32 vector rate (157 TFLOP/s peak on MI355X,
// 1/16 of bf16) -- this path exists for parity, not for the headline metric.
// 256 threads = 4 waves (2x2), 64x64 output tile, BK = 16; operands go global -> registers -> LDS ([row][20] floats: the
// 16 rows x 4 k-columns a fragment read touches fall on 64 distinct banks), register-prefetched one tile ahead.
// As in the bf16 kernel the weight is the MFMA A operand, so a lane owns 4 consecutive output columns of one row.
// Every "bf16" slot of the epilogue table carries fp32 here (out0 / out1 / aux are all float); GELU uses erff.
#include "common.hpp"
#include "dropout.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

struct GemmF32Args {
  const float* A;
  const float* W;
  float* out0;
  float* out1;
  const float* bias;
  const float* res;
  const float* aux;
  const float* pos;
  int M, N, K, lda, ldw, ldo, ldres, ldaux;
  int rows_in, rows_out, row_off;
  int nbn;
  unsigned long long seed; const unsigned long long* seed_ptr; unsigned int drop_thresh; float inv_keep;   // nn.Dropout behind the Linear: mask index m*N + n
};

constexpr int kFT = 64;        // tile rows / cols
constexpr int kFK = 16;        // k per stage
constexpr int kFS = 20;        // LDS row stride (floats)

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(GemmF32Args p) {
  __shared__ __attribute__((aligned(16))) float sA[2][kFT * kFS];
  __shared__ __attribute__((aligned(16))) float sW[2][kFT * kFS];
  const int tile_m = blockIdx.x / p.nbn, tile_n = blockIdx.x - tile_m * p.nbn;
  const int m0 = tile_m * kFT, n0 = tile_n * kFT;
  const int lane = lane_id(), wave = wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  // staging: thread t copies the float4 (row t/4, k 4*(t%4)) of each operand tile (rows beyond M read the padded panel rows)
  const int srow = threadIdx.x >> 2, sk = (threadIdx.x & 3) * 4;
  const float* ag = p.A + (size_t)(m0 + srow) * p.lda + sk;
  const float* wg = p.W + (size_t)(n0 + srow) * p.ldw + sk;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nt = p.K / kFK;
  f32x4 ra = *(const f32x4*)ag, rw = *(const f32x4*)wg;
  *(f32x4*)(&sA[0][srow * kFS + sk]) = ra;
  *(f32x4*)(&sW[0][srow * kFS + sk]) = rw;
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) {
      ra = *(const f32x4*)(ag + (size_t)(t + 1) * kFK);
      rw = *(const f32x4*)(wg + (size_t)(t + 1) * kFK);
    }
#pragma unroll
    for (int s = 0; s < kFK / 4; ++s) {
      float xa[2], wb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) xa[i] = sA[buf][(wm * 32 + i * 16 + l15) * kFS + 4 * s + lq];
#pragma unroll
      for (int j = 0; j < 2; ++j) wb[j] = sW[buf][(wn * 32 + j * 16 + l15) * kFS + 4 * s + lq];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j], xa[i], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nt) {
      *(f32x4*)(&sA[buf ^ 1][srow * kFS + sk]) = ra;
      *(f32x4*)(&sW[buf ^ 1][srow * kFS + sk]) = rw;
    }
    __syncthreads();
  }
  // ---- epilogue: lane owns row m (one per i) x 4 consecutive columns n (per j)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 32 + i * 16 + l15;
    if (m >= p.M) continue;
    size_t orow = (size_t)m;
    int prow = 0;
    if constexpr (EPI == GVK_EPI_PATCH_F32) {
      const int s = m / p.rows_in;
      prow = m - s * p.rows_in;
      orow = (size_t)s * p.rows_out + p.row_off + prow;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 32 + j * 16 + lq * 4;
      f32x4 v = acc[i][j];
      if (p.bias != nullptr) v += *(const f32x4*)(p.bias + n);
      f32x4 dm = {1.f, 1.f, 1.f, 1.f};                   // dropout scale mask of these four elements (this path is for parity, not speed)
      if (p.drop_thresh != 0u) {
        const unsigned long long sd = p.seed + *p.seed_ptr;
#pragma unroll
        for (int e = 0; e < 4; ++e) dm[e] = drop_scale(sd, (unsigned long long)m * p.N + n + e, p.drop_thresh, p.inv_keep);
      }
      if constexpr (EPI == GVK_EPI_STORE_BF16 || EPI == GVK_EPI_STORE_F32) {
        *(f32x4*)(p.out0 + (size_t)m * p.ldo + n) = v;
      } else if constexpr (EPI == GVK_EPI_BIAS_RES_F32 || EPI == GVK_EPI_BIAS_RES_F32_BF16) {
        v *= dm;
        v += *(const f32x4*)(p.res + (size_t)m * p.ldres + n);
        *(f32x4*)(p.out0 + (size_t)m * p.ldo + n) = v;
        if constexpr (EPI == GVK_EPI_BIAS_RES_F32_BF16) *(f32x4*)(p.out1 + (size_t)m * p.ldo + n) = v;
      } else if constexpr (EPI == GVK_EPI_BIAS_GELU_BF16) {
        if (p.out0 != nullptr) *(f32x4*)(p.out0 + (size_t)m * p.ldo + n) = v;
        f32x4 g = {gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
        g *= dm;
        *(f32x4*)(p.out1 + (size_t)m * p.ldo + n) = g;
      } else if constexpr (EPI == GVK_EPI_PATCH_F32) {
        v += *(const f32x4*)(p.pos + (size_t)prow * p.N + n);
        *(f32x4*)(p.out0 + orow * p.ldo + n) = v;
        if (p.out1 != nullptr) *(f32x4*)(p.out1 + (size_t)m * p.ldo + n) = v;
      } else if constexpr (EPI == GVK_EPI_GELU_BWD_BF16) {
        const f32x4 a = *(const f32x4*)(p.aux + (size_t)m * p.ldaux + n);
        v *= dm;
        const f32x4 o = {v[0] * gelu_erf_grad(a[0]), v[1] * gelu_erf_grad(a[1]), v[2] * gelu_erf_grad(a[2]), v[3] * gelu_erf_grad(a[3])};
        *(f32x4*)(p.out0 + (size_t)m * p.ldo + n) = o;
      } else if constexpr (EPI == GVK_EPI_BIAS_RELU_BF16) {
        const f32x4 o = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        *(f32x4*)(p.out0 + (size_t)m * p.ldo + n) = o;
      } else if constexpr (EPI == GVK_EPI_RELU_BWD_BF16) {
        const f32x4 a = *(const f32x4*)(p.aux + (size_t)m * p.ldaux + n);
        const f32x4 o = {a[0] > 0.f ? v[0] : 0.f, a[1] > 0.f ? v[1] : 0.f, a[2] > 0.f ? v[2] : 0.f, a[3] > 0.f ? v[3] : 0.f};
        *(f32x4*)(p.out0 + (size_t)m * p.ldo + n) = o;
      }
    }
  }
}

template <int EPI>
static int launch_f32(const GemmF32Args& a, hipStream_t s) {
  GVK_LAUNCH((gemm_nt_f32_kernel<EPI>), dim3(((a.M + kFT - 1) / kFT) * a.nbn), dim3(256), 0, s, a);
  return check_launch("gemm_nt_f32");
}

}  // namespace gvk

extern "C" int gvk_gemm_nt_f32(const gvk_gemm_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->a && d->w, "gvk_gemm_nt_f32: null operand");
  GVK_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gvk_gemm_nt_f32: empty shape");
  GVK_REQUIRE(d->scale_cols == 0, "gvk_gemm_nt_f32: scale_cols is a bf16-path option (the fp32 attention kernels take the raw q block)");
  GVK_REQUIRE(d->m_panels == 0 && d->m_stride == 0 && d->splitk_ws == nullptr, "gvk_gemm_nt_f32: strided row panels are a bf16-path option");
  GVK_REQUIRE(d->aux_is_grad == 0, "gvk_gemm_nt_f32: aux_is_grad is a bf16-path option");
  GVK_REQUIRE(d->ln_mean == nullptr && d->stat_part == nullptr, "gvk_gemm_nt_f32: the LayerNorm fold / row-statistic partials are bf16-path options");
  GVK_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f && (d->drop_p == 0.f || d->seed_ptr != nullptr), "gvk_gemm_nt_f32: drop_p in [0,1) and a seed word");
  GVK_REQUIRE(d->drop_p == 0.f || d->epilogue == GVK_EPI_BIAS_RES_F32 || d->epilogue == GVK_EPI_BIAS_GELU_BF16 || d->epilogue == GVK_EPI_GELU_BWD_BF16,
              "gvk_gemm_nt_f32: drop_p > 0 is supported by BIAS_RES_F32, BIAS_GELU_BF16 and GELU_BWD_BF16 only");
  GVK_REQUIRE(d->N % kFT == 0 && d->K % kFK == 0, "gvk_gemm_nt_f32: N=%d must be a multiple of 64 and K=%d of 16", d->N, d->K);
  GVK_REQUIRE(d->lda >= d->K && d->ldw >= d->K && d->lda % 4 == 0 && d->ldw % 4 == 0, "gvk_gemm_nt_f32: lda/ldw must be >= K and multiples of 4");
  GVK_REQUIRE(d->ldo % 4 == 0 && d->ldo >= d->N, "gvk_gemm_nt_f32: ldo=%d must be >= N and a multiple of 4", d->ldo);
  GemmF32Args a{};
  a.A = (const float*)d->a; a.W = (const float*)d->w; a.out0 = (float*)d->out0; a.out1 = (float*)d->out1; a.bias = d->bias; a.res = d->res;
  a.aux = (const float*)d->aux; a.pos = d->pos; a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldw = d->ldw; a.ldo = d->ldo;
  a.ldres = d->ldres; a.ldaux = d->ldaux; a.rows_in = d->rows_in; a.rows_out = d->rows_out; a.row_off = d->row_off; a.nbn = d->N / kFT;
  a.seed = d->seed; a.seed_ptr = (const unsigned long long*)d->seed_ptr; a.drop_thresh = drop_threshold_u32(d->drop_p);
  a.inv_keep = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
  hipStream_t s = (hipStream_t)stream;
  switch (d->epilogue) {
    case GVK_EPI_STORE_BF16:
      GVK_REQUIRE(d->out0, "gemm_f32 STORE: out0");
      return launch_f32<GVK_EPI_STORE_BF16>(a, s);
    case GVK_EPI_BIAS_RES_F32:
      GVK_REQUIRE(d->out0 && d->res && d->ldres >= d->N && d->ldres % 4 == 0, "gemm_f32 BIAS_RES: out0/res");
      return launch_f32<GVK_EPI_BIAS_RES_F32>(a, s);
    case GVK_EPI_BIAS_GELU_BF16:
      GVK_REQUIRE(d->out1, "gemm_f32 BIAS_GELU: out1");
      return launch_f32<GVK_EPI_BIAS_GELU_BF16>(a, s);
    case GVK_EPI_PATCH_F32:
      GVK_REQUIRE(d->out0 && d->pos && d->rows_in > 0 && d->rows_out >= d->rows_in + d->row_off, "gemm_f32 PATCH: out0/pos/rows");
      return launch_f32<GVK_EPI_PATCH_F32>(a, s);
    case GVK_EPI_GELU_BWD_BF16:
      GVK_REQUIRE(d->out0 && d->aux && d->ldaux >= d->N && d->ldaux % 4 == 0, "gemm_f32 GELU_BWD: out0/aux");
      return launch_f32<GVK_EPI_GELU_BWD_BF16>(a, s);
    case GVK_EPI_STORE_F32:
      GVK_REQUIRE(d->out0, "gemm_f32 STORE_F32: out0");
      return launch_f32<GVK_EPI_STORE_F32>(a, s);
    case GVK_EPI_BIAS_RES_F32_BF16:
      GVK_REQUIRE(d->out0 && d->out1 && d->res && d->ldres >= d->N && d->ldres % 4 == 0, "gemm_f32 BIAS_RES(2 outputs): out0/out1/res");
      return launch_f32<GVK_EPI_BIAS_RES_F32_BF16>(a, s);
    case GVK_EPI_BIAS_RELU_BF16:
      GVK_REQUIRE(d->out0, "gemm_f32 BIAS_RELU: out0");
      return launch_f32<GVK_EPI_BIAS_RELU_BF16>(a, s);
    case GVK_EPI_RELU_BWD_BF16:
      GVK_REQUIRE(d->out0 && d->aux && d->ldaux >= d->N && d->ldaux % 4 == 0, "gemm_f32 RELU_BWD: out0/aux");
      return launch_f32<GVK_EPI_RELU_BWD_BF16>(a, s);
    default:
      return set_error(-2, "gvk_gemm_nt_f32: unknown epilogue %d", d->epilogue);
  }
}
