// bf16 MFMA GEMM, 128 x 128 workgroup tile with the k axis split FOUR ways: eight waves = 2 column halves x 4 k quarters, every wave
// holding a 128 x 64 accumulator tile (8 x 4 MFMA tiles, 128 VGPRs).
//
// Hypothesis tested: the N = 768 GEMMs of the model (M = 4132: 198 tiles of 128 x 128, one per CU) are bound by LDS bandwidth rather than by MFMA
// issue or by latency (removing the steady-state LDS-DMA takes fc1 dgrad from 33.0 to 25.4 us; a second wave per SIMD at the
// same per-wave tile changes nothing).  With 64 x 64 per wave a 64-wide k-tile costs 64 KB of fragment reads + 32 KB of LDS-DMA writes per
// CU; with 128 x 64 per wave the fragment reads drop to 48 KB (every wave reads ALL 128 activation rows but only 64 weight rows, for a
// quarter of the k range) -- the per-wave tile of gemm8p_bf16.hip on a workgroup tile that still gives 198 workgroups.  The price is the
// final reduction of four partial tiles through LDS (two rounds, fixed order) before the shared epilogue.
//
// Result: correct and reproducible, but 33.0 vs 31.7 us in isolation and 654 vs 716 volumes/s inside the step (128 KB
// of LDS and 215 VGPRs per workgroup leave no room beside it for the side-stream kernels; operands are requested only one stage ahead).
// Opt-in: tile code 4128128 / GAVIKO_HIP_GEMM_K4=1.
//
// LDS: two stages of 64 KB, each two 64-k UNITS (activation 128 x 64 | weight 128 x 64, the swizzled images of gemm_nt_kernel), filled by
// LDS-DMA one stage ahead; one barrier per 128 k.  K is a multiple of 64; an odd unit count leaves the k quarters 2 and 3 idle in the last
// stage.
#include "gemm_epilogue.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

template <int EPI>
__global__ __launch_bounds__(512) void gemm_k4_kernel(GemmArgs p) {
  constexpr int BM = 128, BN = 128, UK = 64;            // tile, k per unit
  constexpr int ROWB = UK * 2, UNIT_A = BM * ROWB, UNIT_W = BN * ROWB, UNIT = UNIT_A + UNIT_W, STAGE = 2 * UNIT;   // 16 + 16 KB, 64 KB
  constexpr int MT = 8, NT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware bijective remap + grouped rasterisation, as gemm_nt_kernel
  const int nwg = p.nbm * p.nbn;
  int wg;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int GROUP_M = 8;
  const int gsz = GROUP_M * p.nbn;
  const int grp_t = wg / gsz, first_m = grp_t * GROUP_M;
  const int gm = min(p.nbm - first_m, GROUP_M);
  const int rem = wg - grp_t * gsz;
  const int tile_m = first_m + rem % gm, tile_n = rem / gm;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int lane = lane_id(), wave = wave_id();
  const int sn = wave & 1, kq = wave >> 1;              // column half; k quarter of every 128-k stage
  const int l15 = lane & 15, lq = lane >> 4;

  const bf16* __restrict__ Ag = p.A + (size_t)m0 * p.lda;
  const bf16* __restrict__ Wg = p.W + (size_t)n0 * p.ldw;
  const int nunits = p.K / UK, nst = (nunits + 1) >> 1;
  // LDS-DMA of one stage: both units, 8 KB per wave (rows (r * 8 + wave) * 8 + lane / 8 of each 128-row image, source-side XOR swizzle)
  auto dma = [&](int st, int buf) {
    const int rsub = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (2 * st + u < nunits) {                         // workgroup-uniform
        char* sA = smem + buf * STAGE + u * UNIT;
        char* sW = sA + UNIT_A;
        const int k0 = (2 * st + u) * UK;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int row = (r * 8 + wave) * 8 + rsub;
          glds16(Ag + (size_t)row * p.lda + k0 + ((slot ^ swz_a128(row)) << 3), sA + (r * 8 + wave) * 1024);
          glds16(Wg + (size_t)row * p.ldw + k0 + ((slot ^ swz_w(row)) << 3), sW + (r * 8 + wave) * 1024);
        }
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  dma(0, 0);
  if (nst > 1) {
    dma(1, 1);
    // stage 0 has landed when at most this wave's share of stage 1 is in flight (8 instructions, or 4 when stage 1 is a single unit)
    if (nunits >= 4) __builtin_amdgcn_s_waitcnt(0x0F70 | 8); else __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  __builtin_amdgcn_s_barrier();
  const int myu = kq >> 1, chunk = (kq & 1) * 4 + lq;    // this wave's unit within a stage, and 16-byte chunk of the unit's rows
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (2 * st + myu < nunits) {                         // wave-uniform (false only for k quarters 2, 3 in an odd last stage)
      const char* sA = smem + buf * STAGE + myu * UNIT;
      const char* sW = sA + UNIT_A;
      bf16x8 xa[MT], wb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = i * 16 + l15;
        xa[i] = *(const bf16x8*)(sA + row * ROWB + ((chunk ^ swz_a128(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = sn * 64 + 32 * (j >> 1) + 8 * (l15 >> 2) + 4 * (j & 1) + (l15 & 3);
        wb[j] = *(const bf16x8*)(sW + row * ROWB + ((chunk ^ swz_w(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
    }
    if (st + 1 < nst) {
      // this wave's share of stage st+1 (requested one stage ago) has landed and its reads of stage st are complete; the barrier extends
      // both to the workgroup, after which stage st's buffer is requested for stage st+2
      __builtin_amdgcn_s_waitcnt(0x0070);                // vmcnt(0) lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
      if (st + 2 < nst) dma(st + 2, buf);
    }
  }

  // ---- four partial tiles -> one.  Round 1: quarters 2, 3 hand theirs to 0, 1 (4 x 32 KB = the two stage buffers).  Round 2: quarters 0 and 1
  // exchange halves -- 0 keeps rows 0..63, 1 keeps rows 64..127 -- so that FOUR waves run the epilogue, 64 x 64 each.
  __syncthreads();
  f32x4* red = (f32x4*)smem;                             // [slot][i * NT + j][lane], a slot = one wave's 128 x 64 tile = 32 KB
  const int tiles = MT * NT;
  if (kq >= 2) {
    f32x4* dst = red + ((kq - 2) * 2 + sn) * tiles * 64;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) dst[(i * NT + j) * 64 + lane] = acc[i][j];
  }
  __syncthreads();
  if (kq >= 2) return;
  {
    const f32x4* src = red + (kq * 2 + sn) * tiles * 64;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] += src[(i * NT + j) * 64 + lane];
  }
  __syncthreads();                                       // (four waves left; the others have exited)
  {
    // quarter 0 gives away its lower half (m tiles 4..7), quarter 1 its upper half (m tiles 0..3): 16 KB per wave
    f32x4* dst = red + (kq * 2 + sn) * (tiles / 2) * 64;
    const int give = kq == 0 ? 4 : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) dst[(i * NT + j) * 64 + lane] = acc[give + i][j];
  }
  __syncthreads();
  f32x4 fin[4][NT];
  {
    const f32x4* src = red + ((kq ^ 1) * 2 + sn) * (tiles / 2) * 64;
    const int keep = kq == 0 ? 0 : 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) fin[i][j] = acc[keep + i][j] + src[(i * NT + j) * 64 + lane];
  }
  gemm_epilogue<EPI, false, 4, NT>(p, fin, m0 + kq * 64, n0 + sn * 64, l15, lq);
}

bool gemm_k4_supports(int epilogue) {
  return epilogue == GVK_EPI_STORE_BF16 || epilogue == GVK_EPI_BIAS_RES_F32 || epilogue == GVK_EPI_STORE_F32;
}

template <int EPI>
static int launch_k4(const GemmArgs& a, hipStream_t stream) {
  constexpr int lds = 2 * 2 * (128 + 128) * 64 * 2;     // two stages of two units: 128 KiB
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_k4_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(gemm_k4): %s", hipGetErrorString(e));
    attr_set = true;
  }
  GemmArgs p = a;
  p.nbm = (a.M + 127) / 128;
  p.nbn = a.N / 128;
  p.a_rows = (a.M + 127) / 128 * 128;
  GVK_LAUNCH((gemm_k4_kernel<EPI>), dim3(p.nbm * p.nbn), dim3(512), lds, stream, p);
  return check_launch("gemm_nt_bf16 (k4)");
}

int launch_gemm_k4(const GemmArgs& a, int epilogue, hipStream_t stream) {
  if (a.N % 128 != 0 || a.K % 64 != 0 || a.K < 128 || a.drop_thresh != 0u)
    return set_error(-2, "gvk_gemm_nt_bf16: the four-way split-k 128x128 tile needs N %% 128 == 0, K %% 64 == 0, K >= 128 and no dropout");
  switch (epilogue) {
    case GVK_EPI_STORE_BF16: return launch_k4<GVK_EPI_STORE_BF16>(a, stream);
    case GVK_EPI_BIAS_RES_F32: return launch_k4<GVK_EPI_BIAS_RES_F32>(a, stream);
    case GVK_EPI_STORE_F32: return launch_k4<GVK_EPI_STORE_F32>(a, stream);
    default: return set_error(-2, "gvk_gemm_nt_bf16: the four-way split-k 128x128 tile is built for STORE_BF16, BIAS_RES_F32 and STORE_F32");
  }
}

}  // namespace gvk
