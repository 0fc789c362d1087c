// bf16 MFMA GEMM, Y = A . W^T, 256 x 256 x 64 tile on EIGHT waves -- the main loop for the wide Linears of the backbone
// (vision_transformer.py:62 to_qkv, :31 fc1 and the dgrad of :34 fc2), one workgroup per CU.
//
// Waves: 2 (M) x 4 (N); wave (wr, wc) owns rows 128 wr .. +127 and columns 64 wc .. +63 of the tile: 8 x 4 accumulator tiles of
// v_mfma_f32_16x16x32_bf16 (128 VGPRs).  A k-tile (BK = 64) is computed in FOUR phases, one 64 x 32 quadrant (mq, nq) of the wave's
// output each = 16 MFMAs: (0,0) (0,1) (1,1) (1,0), so only one operand sub-tile changes between consecutive phases.
//
// The two wave groups wr = 0 / 1 (waves 0-3 / 4-7; waves w and w+4 share a SIMD) run the same program ONE BARRIER APART: while one group
// issues its 16 MFMAs the other reads its fragments from LDS and issues its LDS-DMA, so each SIMD's matrix pipe alternates between its two
// waves (MI355X_MICROARCH.md "Two waves per SIMD").  Per phase and wave:
//     ds_read fragments of this phase | 2 x global_load_lds (one 16-KiB unit) | [counted vmcnt] | s_barrier | lgkmcnt(0) | 16 MFMA | s_barrier
//
// LDS: two k-tile buffers of four 16-KiB UNITS (128 rows x 128 B), staged in the order they are first read:
//     u0 = W rows of the nq = 0 column halves   (read in phase 0, kept in registers for phase 3)
//     u1 = A rows of the mq = 0 row halves      (phase 0)
//     u2 = W rows, nq = 1                       (phase 1)
//     u3 = A rows, mq = 1                       (phase 2)
// The stream of units S_j (j = 4 t + u) runs D units ahead of the reads.  Two placements of the LDS-DMA (template parameter VAR):
//   VAR 1 (tile code 7256256, opt-in: measured 2546 vs 2426 cycles per k-tile): phase g issues S_{g+7} INSIDE its MFMA cluster (an LDS-DMA costs ~60 issue cycles among MFMAs against 100-185 in a
//          section that also carries ds_reads; the load section shrinks to the ds_reads, so the two groups' clusters run back to back);
//   VAR 0 (tile code 8256256, what dispatch_tile picks): phase g issues S_{g+6} in its load section.
// WAR: a unit is restaged no earlier than the MFMA section of the phase AFTER its last ds_read (VAR 1) / two phases after it (VAR 0): the
// reading group retired those reads with lgkmcnt(0) before its own MFMA section, and the other group is exactly one barrier away.
// RAW: each load section ends with a COUNTED vmcnt that leaves every unit in flight except those the NEXT phase reads (4 units = 64 KiB
// stay in flight), then the barrier, then -- one phase later -- the reads ("wait, barrier, read one phase later").
// Rows are XOR-swizzled on the DMA's per-lane SOURCE address (the destination is lane-linear); see gemm_epilogue.hpp for the keys.
#include "gemm_epilogue.hpp"

namespace gvk {

namespace {
constexpr int kUnit = 128 * 128;                  // bytes
constexpr int kBuf = 4 * kUnit;                   // one k-tile: 64 KiB
constexpr int kLds = 2 * kBuf;

#define GVK_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 0xF) | (((n) >> 4) << 14))
#define GVK_LGKMCNT0() __builtin_amdgcn_s_waitcnt(0xC07F)

// tools/probe/probe_gemm8p.hip compiles this file with GVK_STAMPS: shader-clock stamps kept in SGPRs until the end of the kernel
// (p.aux then points at a uint64 [workgroup][wave][32] buffer), and GVK_ABLATE bits that remove one ingredient of the main loop
// (1: no LDS-DMA in the loop, 2: no MFMA, 4: no fragment reads) -- timing only, the results are wrong.  Compiled out of the library.
#ifdef GVK_STAMPS
#define GVK_STAMP(k) st_[k] = __builtin_amdgcn_s_memtime();
#ifndef GVK_ABLATE
#define GVK_ABLATE 0
#endif
#else
#define GVK_STAMP(k)
#define GVK_ABLATE 0
#endif
}  // namespace

template <int EPI, int VAR>
__global__ __launch_bounds__(512) void gemm8p_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 256, BK = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware bijective remap + grouped rasterisation (as gemm_nt_kernel): every XCD works on a contiguous run of tiles whose row panels
  // stay in its L2 while the weight column tiles stream
  const int nwg = p.nbm * p.nbn;
  int wg;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int GROUP_M = 8;
  const int gsz = GROUP_M * p.nbn;
  const int grp = wg / gsz, first_m = grp * GROUP_M;
  const int gm = min(p.nbm - first_m, GROUP_M);
  const int rem = wg - grp * gsz;
  const int tile_m = first_m + rem % gm, tile_n = rem / gm;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int lane = lane_id();
  const int wave = wave_id();
  const int wr = wave >> 2, wc = wave & 3;
#ifdef GVK_STAMPS
  unsigned long long st_[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) st_[k] = 0;
  GVK_STAMP(0)
  const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
  const int l15 = lane & 15, lq = lane >> 4;

  // ---- staging: per unit two 1-KiB LDS-DMA instructions per wave; instruction r of wave w fills unit rows (8 r + w) 8 .. +7
  const bf16* __restrict__ Ag = p.A + (size_t)m0 * p.lda;
  const bf16* __restrict__ Wg = p.W + (size_t)n0 * p.ldw;
  int aoff[2][2], woff[2][2];                     // element offsets of this lane's 16-byte piece, [mq or nq][r]
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int lr = (r * 8 + wave) * 8 + (lane >> 3), slot = lane & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int mrow = min(128 * (lr >> 6) + 64 * h + (lr & 63), p.a_rows - 1 - m0);   // activations are padded to 128 rows, not to the tile
      aoff[h][r] = mrow * p.lda + ((slot ^ swz_a128(lr)) << 3);
      const int nrow = 64 * (lr >> 5) + 32 * h + (lr & 31);
      woff[h][r] = nrow * p.ldw + ((slot ^ swz_w(lr)) << 3);
    }
  }
  // unit U (0..3) of k-tile T
#define GVK_STAGE(U, T)                                                                                      \
  {                                                                                                          \
    const int t_ = (T);                                                                                      \
    char* dst_ = smem + (t_ & 1) * kBuf + (U) * kUnit + wave * 1024;                                         \
    const int k0_ = t_ * BK;                                                                                 \
    if constexpr ((U) == 0 || (U) == 2) {                                                                    \
      glds16(Wg + woff[(U) >> 1][0] + k0_, dst_);                                                            \
      glds16(Wg + woff[(U) >> 1][1] + k0_, dst_ + 8192);                                                     \
    } else {                                                                                                 \
      glds16(Ag + aoff[(U) >> 1][0] + k0_, dst_);                                                            \
      glds16(Ag + aoff[(U) >> 1][1] + k0_, dst_ + 8192);                                                     \
    }                                                                                                        \
  }

  // one of the two LDS-DMA instructions of a unit (VAR 1 issues them between MFMAs)
#define GVK_STAGE_HALF(U, T, R)                                                                              \
  if constexpr ((GVK_ABLATE & 1) == 0) {                                                                     \
    const int t_ = (T);                                                                                      \
    char* dst_ = smem + (t_ & 1) * kBuf + (U) * kUnit + wave * 1024 + (R) * 8192;                            \
    if constexpr ((U) == 0 || (U) == 2) glds16(Wg + woff[(U) >> 1][R] + t_ * BK, dst_);                      \
    else glds16(Ag + aoff[(U) >> 1][R] + t_ * BK, dst_);                                                     \
  }

  // ---- fragment reads: lane-dependent byte offsets inside a unit; everything else is an immediate
  //   A: unit row 64 wr + 16 i + l15, chunk 4 ks + lq;   W: unit row 32 wc + 8 (l15 >> 2) + 4 j + (l15 & 3), same chunk
  int ra[2], rw[2];
  {
    const int arow = 64 * wr + l15, wrow = 32 * wc + 8 * (l15 >> 2) + (l15 & 3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      ra[ks] = arow * 128 + (((4 * ks + lq) ^ swz_a128(arow)) << 4);
      rw[ks] = wrow * 128 + (((4 * ks + lq) ^ swz_w(wrow)) << 4);
    }
  }
  bf16x8 xa[4][2], w0[2][2], w1[2][2];
#define GVK_READ_A(MQ, BASE)                                                                                 \
  if constexpr ((GVK_ABLATE & 4) == 0)                                                                       \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                           \
      xa[i][ks] = *(const bf16x8*)((BASE) + ((MQ) ? 3 : 1) * kUnit + ra[ks] + i * 2048);
#define GVK_READ_W(NQ, WREG, BASE)                                                                           \
  if constexpr ((GVK_ABLATE & 4) == 0)                                                                       \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                           \
      WREG[j][ks] = *(const bf16x8*)((BASE) + ((NQ) ? 2 : 0) * kUnit + rw[ks] + j * 512);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
// MFMAs c*4 .. c*4+3 of a phase's 16: k sub-step c >> 1, row tiles 2 (c & 1), 2 (c & 1) + 1, both column tiles
#define GVK_MMA4(MQ, NQ, WREG, C_)                                                                           \
  if constexpr ((GVK_ABLATE & 2) == 0)                                                                       \
  _Pragma("unroll") for (int i = 2 * ((C_) & 1); i < 2 * ((C_) & 1) + 2; ++i)                                \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
      acc[4 * (MQ) + i][2 * (NQ) + j] =                                                                      \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(WREG[j][(C_) >> 1], xa[i][(C_) >> 1], acc[4 * (MQ) + i][2 * (NQ) + j], 0, 0, 0);

  // barrier | reads retired | the phase's 16 MFMAs (VAR 1: with the two LDS-DMA of unit SU of k-tile ST_ between them) | barrier.
  // The setprio pair also keeps hipcc from moving MFMAs across the barriers.
#define GVK_COMPUTE(MQ, NQ, WREG, DO_STAGE, SU, ST_, SB)                                                     \
  __builtin_amdgcn_sched_barrier(0);                                                                         \
  __builtin_amdgcn_s_barrier();                                                                              \
  if constexpr ((SB) >= 0) { GVK_STAMP((SB) < 0 ? 0 : (SB)) }                                                \
  GVK_LGKMCNT0();                                                                                            \
  if constexpr ((SB) >= 0) { GVK_STAMP((SB) < 0 ? 0 : (SB) + 1) }                                            \
  __builtin_amdgcn_sched_barrier(0);                                                                         \
  __builtin_amdgcn_s_setprio(1);                                                                             \
  GVK_MMA4(MQ, NQ, WREG, 0)                                                                                  \
  if constexpr (VAR == 1 && (DO_STAGE)) {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    GVK_STAGE_HALF(SU, ST_, 0)                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  }                                                                                                          \
  GVK_MMA4(MQ, NQ, WREG, 1)                                                                                  \
  GVK_MMA4(MQ, NQ, WREG, 2)                                                                                  \
  if constexpr (VAR == 1 && (DO_STAGE)) {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    GVK_STAGE_HALF(SU, ST_, 1)                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  }                                                                                                          \
  GVK_MMA4(MQ, NQ, WREG, 3)                                                                                  \
  __builtin_amdgcn_s_setprio(0);                                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                                         \
  if constexpr ((SB) >= 0) { GVK_STAMP((SB) < 0 ? 0 : (SB) + 2) }                                            \
  __builtin_amdgcn_s_barrier();                                                                              \
  if constexpr ((SB) >= 0) { GVK_STAMP((SB) < 0 ? 0 : (SB) + 3) }                                            \
  __builtin_amdgcn_sched_barrier(0);

  // one k-tile.  ST = which of its four phases still have a unit to issue (bit p = phase p); VM0 / VM1 / VM3 = the counted vmcnt
  // closing the load sections of phases 0 / 1 / 3 (63 = no wait; phase 2's successor reads nothing new).
  // Unit issued by phase p of k-tile t:  VAR 0: S_{4t+p+6} = u2, u3 of t+1, u0, u1 of t+2;   VAR 1: S_{4t+p+7} = u3 of t+1, u0, u1, u2 of t+2.
#define GVK_KTILE(T, ST, VM0, VM1, VM3, SB)                                                                      \
  {                                                                                                          \
    const int tt_ = (T);                                                                                     \
    const char* base_ = smem + (tt_ & 1) * kBuf;                                                             \
    GVK_READ_W(0, w0, base_)                                                                                 \
    GVK_READ_A(0, base_)                                                                                     \
    if constexpr (VAR == 0 && ((ST) & 1) != 0 && (GVK_ABLATE & 1) == 0) GVK_STAGE(2, tt_ + 1)                                         \
    GVK_VMCNT(VM0);                                                                                          \
    GVK_COMPUTE(0, 0, w0, ((ST) & 1) != 0, 3, tt_ + 1, SB)                                                       \
    GVK_READ_W(1, w1, base_)                                                                                 \
    if constexpr (VAR == 0 && ((ST) & 2) != 0 && (GVK_ABLATE & 1) == 0) GVK_STAGE(3, tt_ + 1)                                         \
    GVK_VMCNT(VM1);                                                                                          \
    GVK_COMPUTE(0, 1, w1, ((ST) & 2) != 0, 0, tt_ + 2, (SB) < 0 ? -1 : (SB) + 4)                                                       \
    GVK_READ_A(1, base_)                                                                                     \
    if constexpr (VAR == 0 && ((ST) & 4) != 0 && (GVK_ABLATE & 1) == 0) GVK_STAGE(0, tt_ + 2)                                         \
    GVK_COMPUTE(1, 1, w1, ((ST) & 4) != 0, 1, tt_ + 2, (SB) < 0 ? -1 : (SB) + 8)                                                       \
    if constexpr (VAR == 0 && ((ST) & 8) != 0 && (GVK_ABLATE & 1) == 0) GVK_STAGE(1, tt_ + 2)                                         \
    GVK_VMCNT(VM3);                                                                                          \
    GVK_COMPUTE(1, 0, w0, ((ST) & 8) != 0, 2, tt_ + 2, (SB) < 0 ? -1 : (SB) + 12)                                                       \
  }

  const int nt = p.K / BK;                         // >= 2 (checked on the host)
  GVK_STAGE(0, 0) GVK_STAGE(1, 0) GVK_STAGE(2, 0) GVK_STAGE(3, 0)
  GVK_STAGE(0, 1) GVK_STAGE(1, 1)
  if constexpr (VAR == 1) {
    GVK_STAGE(2, 1)
    GVK_VMCNT(10);                                 // u0, u1 of k-tile 0 landed; five units may still be in flight
  } else {
    GVK_VMCNT(8);
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();       // the stagger: group 1 runs one barrier behind group 0 from here on
  __builtin_amdgcn_sched_barrier(0);
  GVK_STAMP(1)
  int t = 0;
#ifdef GVK_STAMPS
  for (; t < nt - 3; ++t) GVK_KTILE(t, 15, 8, 8, 8, -1)
  GVK_STAMP(2)
  if (nt >= 3) { GVK_KTILE(t, 15, 8, 8, 8, 4) ++t; }   // one steady-state k-tile with a stamp at every barrier: st_[4 .. 19]
  GVK_STAMP(20)
#else
  for (; t < nt - 2; ++t) GVK_KTILE(t, 15, 8, 8, 8, -1)
#endif
  if constexpr (VAR == 1) {
    GVK_KTILE(t, 1, 8, 8, 4, -1)                   // k-tile nt-2: one unit of the stream is left (u3 of the last k-tile)
    ++t;
    GVK_KTILE(t, 0, 2, 0, 63, -1)                  // k-tile nt-1
  } else {
    GVK_KTILE(t, 3, 8, 8, 4, -1)                   // k-tile nt-2: the last two units of the stream
    ++t;
    GVK_KTILE(t, 0, 2, 0, 63, -1)                  // k-tile nt-1
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();       // balance the barrier count
#undef GVK_KTILE
#undef GVK_COMPUTE
#undef GVK_MMA4
#undef GVK_STAGE_HALF
#undef GVK_READ_A
#undef GVK_READ_W
#undef GVK_STAGE

  GVK_STAMP(21)
#ifdef GVK_STAMPS
  if constexpr ((GVK_ABLATE & 2) != 0) {           // keep the fragments alive without the MFMAs
#pragma unroll
    for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(xa[i][0]), "v"(xa[i][1])); }
    asm volatile("" ::"v"(w0[0][0]), "v"(w0[1][1]), "v"(w1[0][0]), "v"(w1[1][1]));
  }
  const bf16* aux_keep = p.aux;
  p.aux = nullptr;
#endif
  gemm_epilogue<EPI, false, 8, 4>(p, acc, m0 + 128 * wr, n0 + 64 * wc, l15, lq);
#ifdef GVK_STAMPS
  __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the epilogue's stores have left
  GVK_STAMP(22)
  st_[23] = rt0_;
  st_[24] = __builtin_amdgcn_s_memrealtime();
  if (aux_keep != nullptr && lane == 0) {
    unsigned long long* o = (unsigned long long*)aux_keep + ((size_t)blockIdx.x * 8 + wave) * 32;
#pragma unroll
    for (int k = 0; k < 32; ++k) o[k] = st_[k];
  }
#endif
}

template <int EPI, int VAR>
static int launch8p(const GemmArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_kernel<EPI, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(gemm8p): %s", hipGetErrorString(e));
    attr_set = true;
  }
  GemmArgs p = a;
  p.nbm = (a.M + 255) / 256;
  p.nbn = a.N / 256;
  p.a_rows = (a.M + 127) / 128 * 128;
  GVK_LAUNCH((gemm8p_kernel<EPI, VAR>), dim3(p.nbm * p.nbn), dim3(512), kLds, stream, p);
  return check_launch("gemm8p_bf16");
}

bool gemm8p_supports(int epilogue) {
  switch (epilogue) {
    case GVK_EPI_STORE_BF16: case GVK_EPI_BIAS_GELU_BF16: case GVK_EPI_GELU_BWD_BF16: case GVK_EPI_STORE_F32: case GVK_EPI_BIAS_RES_F32:
      return true;
    default:
      return false;
  }
}

template <int VAR>
static int launch8p_var(const GemmArgs& a, int epilogue, hipStream_t stream) {
  if (a.N % 256 != 0 || a.K % 64 != 0 || a.K < 128 || a.drop_thresh != 0u)
    return set_error(-2, "gemm8p: needs N %% 256 == 0, K %% 64 == 0, K >= 128 and no dropout (N=%d K=%d)", a.N, a.K);
  switch (epilogue) {
    case GVK_EPI_STORE_BF16: return launch8p<GVK_EPI_STORE_BF16, VAR>(a, stream);
    case GVK_EPI_BIAS_GELU_BF16: return launch8p<GVK_EPI_BIAS_GELU_BF16, VAR>(a, stream);
    case GVK_EPI_GELU_BWD_BF16: return launch8p<GVK_EPI_GELU_BWD_BF16, VAR>(a, stream);
    case GVK_EPI_STORE_F32: return launch8p<GVK_EPI_STORE_F32, VAR>(a, stream);
    case GVK_EPI_BIAS_RES_F32: return launch8p<GVK_EPI_BIAS_RES_F32, VAR>(a, stream);
    default: return set_error(-2, "gemm8p: epilogue %d is not built on the 256x256 eight-phase kernel", epilogue);
  }
}

int launch_gemm8p(const GemmArgs& a, int epilogue, int variant, hipStream_t stream) {
  return variant == 0 ? launch8p_var<0>(a, epilogue, stream) : launch8p_var<1>(a, epilogue, stream);
}

}  // namespace gvk
