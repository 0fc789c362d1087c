// bf16 MFMA GEMM for the shapes whose 128 x 128 tiling fills the chip only ONCE (N = 768 at M = 4132: 198 tiles on 256 CUs): the same tile
// on EIGHT waves that split the k axis of every k-tile between them.
//
// With one 4-wave workgroup per CU (gemm_nt_kernel, three LDS stages) every SIMD runs ONE wave, and that wave's k-tile is a chain --
// fragment reads, 32 MFMAs, LDS-DMA issue, barrier -- with nothing to overlap it: measured 1300 cycles per k-tile against 512 of MFMA
// issue, and still 1000 with the LDS-DMA removed (tools/bench_gemm.py with an ablation build: fc1 dgrad 33.0 -> 25.4 us).  A smaller tile
// gives more workgroups but more operand traffic per flop (64 x 128: -6 % on the step).  Here the tile, its LDS image and its staging are
// unchanged, but waves 0-3 multiply the FIRST 32-wide half of every 64-wide k-tile and waves 4-7 the SECOND half, each into its own
// 64 x 64 accumulators: every SIMD holds two waves whose chains interleave, each wave issues half of the LDS-DMA instructions, and the LDS
// bytes read per flop are those of the 4-wave form (a 4 x 2 wave grid over the tile would read 1.5 x as much).  The two partial tiles are
// summed through LDS once, after the main loop (64 KB, fixed order), and waves 0-3 run the shared epilogue.
//
// Result: correct, reproducible, 0-9 % faster in isolation, NO gain inside the step -- the second wave per SIMD
// does not help: the single wave's latency chain is not what bounds the k-tile (nor are the LDS bytes: gemm_k4_bf16.hip).  Selected by
// tile code 9128128 or GAVIKO_HIP_GEMM_K2=1.
#include "gemm_epilogue.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

template <int EPI>
__global__ __launch_bounds__(512) void gemm_k2_kernel(GemmArgs p) {
  constexpr int BM = 128, BN = 128, BK = 64, NS = 3, NW = 8;
  constexpr int WM = 64, WN = 64, MT = 4, NT = 4;
  constexpr int ROWB = BK * 2, RPI = 1024 / ROWB, CPR = ROWB / 16;
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
  constexpr int PER_TILE = BM / (NW * RPI) + BN / (NW * RPI);          // LDS-DMA instructions per wave and k-tile: 4
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
  const int kh = wave >> 2, w4 = wave & 3;              // k half of every k-tile this wave multiplies; position in the 2 x 2 wave grid
  const int wm = w4 >> 1, wn = w4 & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  const bf16* __restrict__ Ag = p.A + (size_t)m0 * p.lda;
  const bf16* __restrict__ Wg = p.W + (size_t)n0 * p.ldw;
  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * STAGE;
    char* sW = sA + A_BYTES;
    const int k0 = kt * BK;
    const int rsub = lane / CPR, slot = lane % CPR;
#pragma unroll
    for (int r = 0; r < BM / (NW * RPI); ++r) {
      const int row = (r * NW + wave) * RPI + rsub;
      glds16(Ag + (size_t)row * p.lda + k0 + ((slot ^ swz_a128(row)) << 3), sA + (r * NW + wave) * 1024);
    }
#pragma unroll
    for (int r = 0; r < BN / (NW * RPI); ++r) {
      const int row = (r * NW + wave) * RPI + rsub;
      glds16(Wg + (size_t)row * p.ldw + k0 + ((slot ^ swz_w(row)) << 3), sW + (r * NW + wave) * 1024);
    }
  };
  // this wave's fragments of one k-tile: its 32-wide half only
  auto load_frags = [&](int buf, bf16x8 (&xa)[MT], bf16x8 (&wb)[NT]) {
    const char* sA = smem + buf * STAGE;
    const char* sW = sA + A_BYTES;
    const int chunk = kh * 4 + lq;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = wm * WM + i * 16 + l15;
      xa[i] = *(const bf16x8*)(sA + row * ROWB + ((chunk ^ swz_a128(row)) << 4));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * WN + 32 * (j >> 1) + 8 * (l15 >> 2) + 4 * (j & 1) + (l15 & 3);
      wb[j] = *(const bf16x8*)(sW + row * ROWB + ((chunk ^ swz_w(row)) << 4));
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt = p.K / BK;                               // >= NS (the launcher checks K >= 192)
#pragma unroll
  for (int s = 0; s < NS; ++s) stage(s, s);
  __builtin_amdgcn_s_waitcnt(0x0F70 | (((NS - 1) * PER_TILE) & 0xF) | ((((NS - 1) * PER_TILE) >> 4) << 14));   // tile 0 in; two may be in flight
  __builtin_amdgcn_s_barrier();
  bf16x8 xaA[MT], wbA[NT], xaB[MT], wbB[NT];
  load_frags(0, xaA, wbA);
  // One k-tile.  CUR holds this wave's fragments of tile T (read one tile ago); its 16 MFMAs are issued first and run while the wave
  // waits for tile T+1 (counted vmcnt: AHEAD tiles requested after it may stay in flight), meets the others at the barrier -- every wave
  // has tile T in registers by then, so its buffer is re-requested for tile T+NS -- and reads its fragments of tile T+1 into NXT.
#define GVK_K2_TILE(T, CXA, CWB, NXA, NWB)                                                                                   \
  {                                                                                                                           \
    const int nxt_ = b == NS - 1 ? 0 : b + 1;                                                                                  \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): CUR has arrived */                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                        \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                            \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                            \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(CWB[j], CXA[i], acc[i][j], 0, 0, 0);                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                        \
    if ((T) + 1 < nt) {                                                                                                       \
      if ((T) + 2 < nt) __builtin_amdgcn_s_waitcnt(0x0F70 | PER_TILE); /* tile T+2 may stay in flight */                       \
      else __builtin_amdgcn_s_waitcnt(0x0F70);                                                                                \
      __builtin_amdgcn_s_barrier();                                                                                           \
      if ((T) + NS < nt) stage(b, (T) + NS);                                                                                  \
      load_frags(nxt_, NXA, NWB);                                                                                             \
    }                                                                                                                         \
    b = nxt_;                                                                                                                 \
  }
  int b = 0;
  for (int t = 0; t < nt; t += 2) {
    GVK_K2_TILE(t, xaA, wbA, xaB, wbB)
    if (t + 1 < nt) GVK_K2_TILE(t + 1, xaB, wbB, xaA, wbA)
  }
#undef GVK_K2_TILE

  // ---- the two k halves meet: waves 4-7 hand their partial tile over through LDS (every stage buffer is free: each wave's last fragment
  // reads completed before its last MFMAs were issued), waves 0-3 add it and run the epilogue
  __syncthreads();
  f32x4* red = (f32x4*)smem;                             // [w4][i * NT + j][lane]: 64 KiB of the 96
  if (kh == 1) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) red[(w4 * (MT * NT) + i * NT + j) * 64 + lane] = acc[i][j];
  }
  __syncthreads();
  if (kh == 1) return;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] += red[(w4 * (MT * NT) + i * NT + j) * 64 + lane];
  gemm_epilogue<EPI, false, MT, NT>(p, acc, m0 + wm * WM, n0 + wn * WN, l15, lq);
}

bool gemm_k2_supports(int epilogue) {
  return epilogue == GVK_EPI_STORE_BF16 || epilogue == GVK_EPI_BIAS_RES_F32 || epilogue == GVK_EPI_STORE_F32;
}

template <int EPI>
static int launch_k2(const GemmArgs& a, hipStream_t stream) {
  constexpr int lds = 3 * (128 + 128) * 64 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_k2_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(gemm_k2): %s", hipGetErrorString(e));
    attr_set = true;
  }
  GemmArgs p = a;
  p.nbm = (a.M + 127) / 128;
  p.nbn = a.N / 128;
  p.a_rows = (a.M + 127) / 128 * 128;
  GVK_LAUNCH((gemm_k2_kernel<EPI>), dim3(p.nbm * p.nbn), dim3(512), lds, stream, p);
  return check_launch("gemm_nt_bf16 (k2)");
}

int launch_gemm_k2(const GemmArgs& a, int epilogue, hipStream_t stream) {
  if (a.N % 128 != 0 || a.K % 64 != 0 || a.K < 192 || a.drop_thresh != 0u)
    return set_error(-2, "gvk_gemm_nt_bf16: the split-k 128x128 tile needs N %% 128 == 0, K %% 64 == 0, K >= 192 and no dropout");
  switch (epilogue) {
    case GVK_EPI_STORE_BF16: return launch_k2<GVK_EPI_STORE_BF16>(a, stream);
    case GVK_EPI_BIAS_RES_F32: return launch_k2<GVK_EPI_BIAS_RES_F32>(a, stream);
    case GVK_EPI_STORE_F32: return launch_k2<GVK_EPI_STORE_F32>(a, stream);
    default: return set_error(-2, "gvk_gemm_nt_bf16: the split-k 128x128 tile is built for STORE_BF16, BIAS_RES_F32 and STORE_F32");
  }
}

}  // namespace gvk
