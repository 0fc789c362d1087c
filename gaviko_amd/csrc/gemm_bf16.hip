// bf16 MFMA GEMM, Y = A . W^T with fused epilogues, for gfx950.
//
// Structure: 256-thread workgroup = 4 waves (2x2), BMxBN output tile, BK = 64.  Both operands are K-contiguous,
// staged global -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4), double-buffered.  An LDS tile row is
// 64 bf16 = 128 B = eight 16-B chunks; chunk c of row r is stored at chunk position c ^ ((r>>1)&7) (the XOR is
// applied on the per-lane SOURCE address because the DMA destination is lane-linear) so that the 16 rows a
// ds_read_b128 lane group touches land on 16 distinct 16-B slots of the 256-B bank row.
// MFMA: v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment as the A operand and the ACTIVATION fragment as the B
// operand, i.e. each wave computes D[n][m]; a lane owns 4 consecutive n of one row m per 16x16 tile.  The weight rows
// fed to tiles 2p and 2p+1 are interleaved (tile j, A row a  <->  n = 32(j>>1) + 8(a>>2) + 4(j&1) + (a&3)), so across the
// pair a lane owns EIGHT consecutive n: every epilogue access is a 16-byte (bf16) or 32-byte (fp32) contiguous piece of
// the row-major output.  (With 4-wide pieces the two-output GELU epilogue took 44 us against 31 us for the same GEMM
// with one output: store-instruction bound, not byte bound.)  The weight tile has its own swizzle key so that the
// interleaved rows stay conflict-free.
#include "gemm_epilogue.hpp"

namespace gvk {

// LDS swizzles (applied to the 16-byte chunk index of a tile row; conflict-free for the ds_read_b128 lane groups):
//   BK = 64: 128-byte rows, chunk ^ ((row >> 1) & 7);   BK = 32: 64-byte rows, chunk ^ {0,2,3,1}[(row >> 2) & 3]
template <int BK>
__device__ __forceinline__ int swz_chunk(int row) {
  if constexpr (BK == 64) return (row >> 1) & 7;
  else return (0x78 >> (((row >> 2) & 3) * 2)) & 3;      // 0b01'11'10'00 -> 0,2,3,1
}

// NS = LDS stages.  2: the tuned default (two workgroups per CU hide each other's stalls).  3: for the shapes that run ONE workgroup
// per CU (128 x 128 tiles of the N = 768 GEMMs: 198 tiles) -- there a third tile in flight is what hides the L2 round trip.
// NW = waves per workgroup: 4 (2 x 2 over the tile) or 8 (4 x 2: the 256 x 256 tile of the wide shapes, 64 x 128 per wave).
// SPLITK (strided row panels only): a handful of 64-row tiles with 12 - 49 k-tiles each runs at the latency of ONE workgroup's k loop
// (0.6 - 0.75 us per k-tile: 36 us for K = 3072 on 24 of 256 CUs); cut into pieces over idle CUs the loop is ~6 k-tiles long.  The
// pieces' partial tiles are summed in piece order by the last arriver (write-through stores, drained, agent-scope ticket, acquire:
// the hand-off of paramgrad.hip), so the result does not depend on which workgroup that is.
template <int BM, int BN, int EPI, int BK = 64, bool DROP = false, int NS = 2, int NW = 4, bool SPLITK = false>
__global__ __launch_bounds__(64 * NW) void gemm_nt_kernel(GemmArgs p) {
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(!SPLITK || (NW == 4 && !DROP), "split-K is built on the 4-wave loops");
  static_assert(BK == 64, "the interleaved weight-row mapping is built for 128-byte tile rows");
  static_assert(NS >= 2 && NS <= 4, "2..4 LDS stages");
  constexpr int WM = BM / (NW / 2), WN = BN / 2;
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int ROWB = BK * 2;                       // bytes per LDS tile row
  constexpr int RPI = 1024 / ROWB;                   // rows covered by one 1-KiB wave LDS-DMA instruction
  constexpr int CPR = ROWB / 16;                     // 16-byte chunks per row
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD (own L2); give each XCD a contiguous run of tiles.
  const int nwg = p.nbm * p.nbn;
  int wg;
  [[maybe_unused]] int piece = 0;
  if constexpr (SPLITK) {                                  // the same remap over (tile, piece) units: the pieces of a tile are neighbours in an XCD's run
    const int nu = nwg * p.ksplit;
    const int bid = blockIdx.x, xcd = bid & 7, q = nu >> 3, r = nu & 7;
    const int u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    wg = u / p.ksplit;
    piece = u - wg * p.ksplit;
  } else {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // Grouped rasterisation inside the XCD's run: group_m row panels x all column tiles, m fastest.  The L2 (4 MiB / XCD) then holds
  // the group's A panels while each weight column tile is streamed once per group instead of once per row panel
  // (rocprofv3 FETCH_SIZE of the fc2-dgrad GEMM: 110 MB with the plain n-fastest order vs 36 MB algorithmic).
  // group_m is chosen by the launcher so that ONE group is about one XCD's run of tiles (all of which are resident together): every column
  // tile of a row panel then runs on the same XCD at the same time and the activation panel is fetched from HBM / MALL once, not once per
  // XCD that holds a piece of the group (round 2, fixed groups of 8 panels: the fc2-forward activation panel was fetched about twice).
  const int GROUP_M = p.group_m;
  const int gsz = GROUP_M * p.nbn;
  const int grp = wg / gsz, first_m = grp * GROUP_M;
  const int gm = min(p.nbm - first_m, GROUP_M);
  const int rem = wg - grp * gsz;
  int tile_m = first_m + rem % gm, tile_n = rem / gm;
  if (p.xcd_panels > 0) {
    // One-round launches (every tile resident at once, <= one per CU): the XCD's contiguous run of tiles in PANEL-MAJOR order, so that a row
    // panel's column tiles sit on one XCD and an activation panel is fetched by one L2 -- except the panel a run boundary cuts (7 of 33
    // at M = 4132).  The grouped order above (m fastest inside groups of four panels, built for multi-round launches) let a run of 24.75
    // tiles spill a few tiles into the next group and pull ALL its panels: rocprofv3 FETCH_SIZE of the fc1 dgrad 81 MB against
    // 25.4 (activation) + 8 x 4.7 (the weight once per XCD) = 63.  (Whole panels per XCD -- 5,4,4,... -- fetch the least, 63.8 MB, but give
    // one XCD 30 tiles against 24: its L2 -> LDS stream then sets the kernel time, +2.3 us on the K = 3072 shapes.)
    tile_m = wg / p.nbn;
    tile_n = wg - tile_m * p.nbn;
  }
  const int m0 = tile_m * p.m_stride, n0 = tile_n * BN;      // (m_stride = BM except for strided row panels)

  const int lane = lane_id();
  const int wave = wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  const bf16* __restrict__ Ag = p.A + (size_t)m0 * p.lda;
  const bf16* __restrict__ Wg = p.W + (size_t)n0 * p.ldw;
  if constexpr (SPLITK) {                                  // this piece's k range starts at k-tile piece * kt_per
    Ag += (size_t)piece * p.kt_per * BK;
    Wg += (size_t)piece * p.kt_per * BK;
  }

  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * STAGE;
    char* sW = sA + A_BYTES;
    const int k0 = kt * BK;
    const int rsub = lane / CPR, slot = lane % CPR;
#pragma unroll
    for (int r = 0; r < BM / (NW * RPI); ++r) {
      const int row = (r * NW + wave) * RPI + rsub;
      const int chunk = slot ^ swz_chunk<BK>(row);
      int srow = row;
      if constexpr (BM > 128 || 128 % BM != 0) srow = min(row, p.a_rows - 1 - m0);     // activations are padded to 128 rows, not to the tile: re-read the last one
      glds16(Ag + (size_t)srow * p.lda + k0 + chunk * 8, sA + (r * NW + wave) * 1024);
    }
#pragma unroll
    for (int r = 0; r < BN / (NW * RPI); ++r) {
      const int row = (r * NW + wave) * RPI + rsub;
      const int chunk = slot ^ swz_w(row);
      glds16(Wg + (size_t)row * p.ldw + k0 + chunk * 8, sW + (r * NW + wave) * 1024);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int nt = p.K / BK;
  if constexpr (SPLITK) nt = min(p.kt_per, nt - piece * p.kt_per);
  if constexpr (BK == 64) {
    // Software-pipelined main loop.  Fragments of the next 32-wide k sub-step are read from LDS while the 16 MFMAs of the
    // current one run, so no MFMA ever waits on an LDS read it was issued behind; the single barrier per tile sits between
    // the two sub-steps.  After that barrier every wave holds tile t entirely in registers, so the LDS-DMA of tile t+2 can
    // already overwrite tile t's buffer: two buffers, two tiles of prefetch distance.
#define GVK_LOAD_FRAGS(SA, SW, KS, XA, WB)                                                                  \
  {                                                                                                         \
    const int chunk_ = (KS) * 4 + lq;                                                                       \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                        \
      const int row = wm * WM + i * 16 + l15;                                                               \
      XA[i] = *(const bf16x8*)((SA) + row * ROWB + ((chunk_ ^ swz_chunk<BK>(row)) << 4));                   \
    }                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                        \
      const int row = wn * WN + 32 * (j >> 1) + 8 * (l15 >> 2) + 4 * (j & 1) + (l15 & 3);                   \
      WB[j] = *(const bf16x8*)((SW) + row * ROWB + ((chunk_ ^ swz_w(row)) << 4));                           \
    }                                                                                                       \
  }
#define GVK_MMA(XA, WB)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                            \
  _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                            \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WB[j], XA[i], acc[i][j], 0, 0, 0);
    bf16x8 xa0[MT], wb0[NT], xa1[MT], wb1[NT];
    constexpr int PER_TILE = BM / (NW * RPI) + BN / (NW * RPI);
    if constexpr (NS == 2) {
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    // tile 0 must have landed; tile 1 may still be in flight (PER_TILE LDS-DMA instructions per wave and tile)
    if (nt > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (PER_TILE & 0xF) | ((PER_TILE >> 4) << 14));   // vmcnt(PER_TILE)
    else __builtin_amdgcn_s_waitcnt(0x0F70);                                                        // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    GVK_LOAD_FRAGS(smem, smem + A_BYTES, 0, xa0, wb0)
    __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): the loop is entered with nothing outstanding on LDS
    // the last two tiles prefetch nothing: peeled, so that the steady-state body is one branch-free block
#define GVK_TILE(T, PREFETCH)                                                                               \
    {                                                                                                       \
      const int buf = (T) & 1;                                                                              \
      const char* sA = smem + buf * STAGE;                                                                  \
      const char* sW = sA + A_BYTES;                                                                        \
      GVK_LOAD_FRAGS(sA, sW, 1, xa1, wb1)                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      GVK_MMA(xa0, wb0)                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      __syncthreads(); /* vmcnt(0): tile T+1 landed; lgkmcnt(0): this wave's reads of tile T are done */    \
      if (PREFETCH) stage(buf, (T) + 2);                                                                    \
      GVK_LOAD_FRAGS(smem + (buf ^ 1) * STAGE, smem + (buf ^ 1) * STAGE + A_BYTES, 0, xa0, wb0)             \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      GVK_MMA(xa1, wb1)                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) -- free: those reads were issued 16 MFMAs ago */    \
    }
    int t = 0;
    for (; t < nt - 2; ++t) GVK_TILE(t, true)
    for (; t < nt; ++t) GVK_TILE(t, false)
#undef GVK_TILE
    } else {
    // NS stages: tiles 0..NS-1 are requested up front, tile t+NS as soon as tile t's buffer is free.  A wave's LDS-DMA instructions
    // complete in order, so "tile t+1 has landed" is vmcnt(PER_TILE * #tiles requested after t+1): NS-2 in the steady state.
    static_assert((NS - 1) * PER_TILE <= 63, "vmcnt is a 6-bit counter");
#pragma unroll
    for (int sgi = 0; sgi < NS; ++sgi)
      if (sgi < nt) stage(sgi, sgi);
    {
      const int ahead = min(nt, NS) - 1;                   // tiles that may still be in flight once tile 0 is in
      if (ahead >= 3) __builtin_amdgcn_s_waitcnt(0x0F70 | ((3 * PER_TILE) & 0xF) | (((3 * PER_TILE) >> 4) << 14));
      else if (ahead == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * PER_TILE) & 0xF) | (((2 * PER_TILE) >> 4) << 14));
      else if (ahead == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (PER_TILE & 0xF) | ((PER_TILE >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    GVK_LOAD_FRAGS(smem, smem + A_BYTES, 0, xa0, wb0)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    // AHEAD = tiles requested after tile T+1 when tile T reaches its barrier (a literal: s_waitcnt takes an immediate); LAST: no tile T+1
#define GVK_TILE3(T, BUF, PREFETCH, AHEAD, LAST)                                                            \
    {                                                                                                       \
      const int buf = (BUF);                                                                                \
      const int nxt = buf == NS - 1 ? 0 : buf + 1;                                                          \
      const char* sA = smem + buf * STAGE;                                                                  \
      const char* sW = sA + A_BYTES;                                                                        \
      GVK_LOAD_FRAGS(sA, sW, 1, xa1, wb1)                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      GVK_MMA(xa0, wb0)                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      /* this wave's share of tile T+1 landed, its reads of tile T done; the barrier extends both to all waves */ \
      __builtin_amdgcn_s_waitcnt(0x0070 | (((AHEAD) * PER_TILE) & 0xF) | ((((AHEAD) * PER_TILE) >> 4) << 14)); \
      __builtin_amdgcn_s_barrier();                                                                         \
      if (PREFETCH) stage(buf, (T) + NS);                                                                   \
      if (!(LAST)) GVK_LOAD_FRAGS(smem + nxt * STAGE, smem + nxt * STAGE + A_BYTES, 0, xa0, wb0)            \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      GVK_MMA(xa1, wb1)                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                                                   \
    }
    int t = 0, b = 0;
    for (; t < nt - NS; ++t) {
      GVK_TILE3(t, b, true, NS - 2, false)
      b = b == NS - 1 ? 0 : b + 1;
    }
    for (; t < nt; ++t) {                                  // the last (up to) NS tiles request nothing
      const int r = nt - 1 - t;                            // tiles left after this one; r - 1 of them may still be in flight
      if (NS > 3 && r == 3) GVK_TILE3(t, b, false, 2, false)
      else if (r == 2) GVK_TILE3(t, b, false, 1, false)
      else if (r == 1) GVK_TILE3(t, b, false, 0, false)
      else GVK_TILE3(t, b, false, 0, true)
      b = b == NS - 1 ? 0 : b + 1;
    }
#undef GVK_TILE3
    }
#undef GVK_LOAD_FRAGS
#undef GVK_MMA
  }

  if constexpr (SPLITK) {
    // partial tile -> memory (write-through), ticket, and the last arriver of the tile sums the pieces in piece order
    const int tile_id = tile_m * p.nbn + tile_n;
    constexpr int kPiece = NW * MT * NT * 64 * 4;          // floats of one partial tile
    float* const base = p.sk_part + ((size_t)tile_id * p.ksplit) * kPiece;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, p.ksplit * kPiece * 4, 0x00020000);
    const int off = ((wave * MT * NT) * 64 + lane) * 16;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, (piece * kPiece) * 4 + off + (i * NT + j) * 1024, 0, 16);   // aux 16 = sc1
    int* const s_last = (int*)smem;                        // (the tiles are dead; a static __shared__ word would shift the dynamic region off its 16-byte alignment)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY storing wave drains its write-through stores ...
    __syncthreads();                                       // ... before the one lane that signals for all of them
    if (threadIdx.x == 0) {
      const int t = __hip_atomic_fetch_add(p.sk_tick + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (t == p.ksplit - 1) ? 1 : 0;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // drop this CU's stale lines before any wave of it reads a partial
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.sk_tick + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this stream
      }
      *s_last = last;
    }
    __syncthreads();
    if (!*s_last) return;
    if (p.ksplit == 2) {
      // two pieces: own accumulators + the other piece's partial -- a + b is the same bits in either order, so who arrives last does not matter
      const int pc = piece ^ 1;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] += *(const f32x4*)((const char*)base + (size_t)pc * kPiece * 4 + off + (i * NT + j) * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int pc = 0; pc < p.ksplit; ++pc) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] += *(const f32x4*)((const char*)base + (size_t)pc * kPiece * 4 + off + (i * NT + j) * 1024);
      }
    }
  }
  gemm_epilogue<EPI, DROP, MT, NT>(p, acc, m0 + wm * WM, n0 + wn * WN, l15, lq);
}

template <int BM, int BN, int EPI, bool DROP = false, int NS = 2, int NW = 4, bool SPLITK = false>
static int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  constexpr int BK = 64;
  constexpr int lds = NS * (BM + BN) * BK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, EPI, BK, DROP, NS, NW, SPLITK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(gemm %dx%d): %s", BM, BN, hipGetErrorString(e));
    attr_set = true;
  }
  GemmArgs p = a;
  p.nbm = (a.M + BM - 1) / BM;
  p.nbn = a.N / BN;
  p.a_rows = (a.M + 127) / 128 * 128;
  p.m_stride = BM;
  if (a.m_stride > 0) {                                  // strided row panels (gvk_gemm_desc.m_panels): nbm was set by the caller
    if (a.m_stride < BM || (long)(a.nbm - 1) * a.m_stride + BM > p.a_rows)
      return set_error(-2, "gvk_gemm_nt_bf16: %d row panels of %d rows at stride %d do not fit %d (padded) rows", a.nbm, BM, a.m_stride, p.a_rows);
    p.nbm = a.nbm;
    p.m_stride = a.m_stride;
  }
  {
    // row panels per group ~ (tiles per XCD) / (column tiles), within 1..8: N = 768 at M = 4132 -> 198 tiles, 24.75 per XCD, 6 columns -> 4
    const int per_xcd = (p.nbm * p.nbn) >> 3;
    int g = (per_xcd + p.nbn / 2) / p.nbn;
    g = g < 1 ? 1 : g > 8 ? 8 : g;
    static const int force = diag_env("GAVIKO_HIP_GEMM_GROUP_M") ? atoi(diag_env("GAVIKO_HIP_GEMM_GROUP_M")) : 0;     // A/B switch (8 = the round-2 mapping)
    p.group_m = force > 0 ? force : g;
  }
  const int grid = p.nbm * p.nbn;
  p.xcd_panels = 0;
  {
    // panel-major tile order for the one-round launches of the three-stage (one workgroup per CU) kernel
    static const int mode = diag_env("GAVIKO_HIP_GEMM_XCD_PANELS") ? atoi(diag_env("GAVIKO_HIP_GEMM_XCD_PANELS")) : 1;     // A/B switch
    if (mode != 0 && NS >= 3 && grid <= 256) p.xcd_panels = 1;
  }
  if constexpr (SPLITK) {
    // pieces of at least two k-tiles, as many as keep the launch within one round of the chip
    const int nt = a.K / BK;
    int s = 256 / (grid > 0 ? grid : 1);
    s = s < 1 ? 1 : s > 8 ? 8 : s;
    if (a.ksplit > 0) s = a.ksplit;                        // the caller's piece count (gvk_gemm_desc.ksplit)
    if (s > nt / 2) s = nt / 2 > 0 ? nt / 2 : 1;
    p.kt_per = (nt + s - 1) / s;
    p.ksplit = (nt + p.kt_per - 1) / p.kt_per;
    constexpr size_t kPiece = (size_t)NW * (BM / (NW / 2) / 16) * (BN / 2 / 16) * 64 * 4 * 4;      // bytes of one partial tile
    const size_t need = 1024 + (size_t)grid * p.ksplit * kPiece;
    if (a.sk_part == nullptr || a.sk_bytes < need || grid > 256)
      return set_error(-2, "gvk_gemm_nt_bf16: split-K workspace of %zu bytes needed (%zu given; at most 256 tiles)", need, a.sk_bytes);
    p.sk_tick = (int*)a.sk_part;                           // [256 ticket words | partial tiles]
    p.sk_part = (float*)((char*)a.sk_part + 1024);
    GVK_LAUNCH((gemm_nt_kernel<BM, BN, EPI, BK, DROP, NS, NW, SPLITK>), dim3(grid * p.ksplit), dim3(64 * NW), lds, stream, p);
    return check_launch("gemm_nt_bf16 (split-K panels)");
  }
  GVK_LAUNCH((gemm_nt_kernel<BM, BN, EPI, BK, DROP, NS, NW, SPLITK>), dim3(grid), dim3(64 * NW), lds, stream, p);
  return check_launch("gemm_nt_bf16");
}

static long wide_lo() {
  // 153 tiles (the qkv shape at M = 4132) included: +0.7 % on the ViT-B step; 144 tiles (fc1 / fc2 dgrad of ViT-L at M = 2066) included:
  // 196.7 -> 200.0 volumes/s at cfg5, while 108 tiles (its qkv shape) lose on the big tile (198.0)
  static const long v = diag_env("GAVIKO_HIP_GEMM_WIDE_LO") ? atol(diag_env("GAVIKO_HIP_GEMM_WIDE_LO")) : 140;
  return v;
}

template <int EPI>
static int dispatch_tile(const GemmArgs& a, int tile, hipStream_t stream) {
  if (a.m_stride > 0) {                                  // strided row panels: a few short tiles, latency-bound -- 64 x 128 with three stages
    if (tile == 0) tile = (a.N % 128 == 0 && a.drop_thresh == 0u) ? 4064128 : 64064;       // (four stages: these launches are bound by the L2 round trip of a k-step)
    if (tile == 8256256 || tile == 7256256 || tile == 256256) return set_error(-2, "gvk_gemm_nt_bf16: strided row panels run on the 4-wave tiles only");
  }
  if (tile == 0) {
    const int bn = (a.N % 128 == 0) ? 128 : 64;
    // fill the 256 CUs: fall back to 64-row tiles when 128-row tiles give < ~1.5 workgroups per CU
    const long t128 = (long)((a.M + 127) / 128) * (a.N / bn);
    // A/B switch GAVIKO_HIP_GEMM_N768: 64 = 64x128 tiles (two workgroups on most CUs), 128 = 128x128 two-stage, default = 128x128
    // with three stages at one workgroup per CU
    static const int n768 = diag_env("GAVIKO_HIP_GEMM_N768") ? atoi(diag_env("GAVIKO_HIP_GEMM_N768")) : 3128;
    const int bm = (t128 >= 384) ? 128 : 64;
    tile = bm * 1000 + bn;
    if constexpr (EPI == GVK_EPI_STORE_BF16 || EPI == GVK_EPI_BIAS_GELU_BF16 || EPI == GVK_EPI_GELU_BWD_BF16) {
      // wide shapes whose 256 x 256 tiles give (just under) one workgroup per CU: eight waves share one staging of twice the rows
      static const int wide = diag_env("GAVIKO_HIP_GEMM_WIDE") ? atoi(diag_env("GAVIKO_HIP_GEMM_WIDE")) : 8;   // A/B switch: 0 = off, 256 = the one-barrier 256x256 kernel, 8 = gemm8p
      const long t256 = (long)((a.M + 255) / 256) * (a.N / 256);
      static const bool wide_bwd = diag_env("GAVIKO_HIP_GEMM_WIDE_BWD") == nullptr || diag_env("GAVIKO_HIP_GEMM_WIDE_BWD")[0] != '0';
      if (wide != 0 && a.N % 256 == 0 && a.drop_thresh == 0u && t256 >= wide_lo() && t256 <= 256 && (EPI != GVK_EPI_GELU_BWD_BF16 || wide_bwd))
        tile = (wide == 256 || a.K < 128) ? 256256 : 8256256;
    }
    // N = 768-type shapes (64 x 128 by the fill rule above) with K >= 512 run THREE LDS stages: as 128 x 128 tiles at one workgroup per CU
    // when those fill at least half the chip (M = 4132: 198 tiles), as 64 x 128 tiles below that (M = 2066: 102 tiles of 128 rows would leave
    // 60 % of the CUs idle; 198 tiles of 64: 465 -> 488 volumes/s at B = 2, while at B = 4 the small tile costs 6 %)
    static const long t64_hi = diag_env("GAVIKO_HIP_GEMM_T64HI") ? atol(diag_env("GAVIKO_HIP_GEMM_T64HI")) : 130;
    if (bm == 64 && bn == 128 && t128 <= 256 && a.K >= 512 && a.drop_thresh == 0u) {
      if (n768 == 3128) {
        // One workgroup per CU, three stages: the tile's life is bound by its own L2 -> LDS bytes, (BM + 128) per k-step, and a launch takes
        // ceil(tiles / 256) such lives.  Row tile = the one of {128, 96, 64} with the smallest rounds x (BM + 128):
        //   M = 4132, N = 768 : 198 / 264 / 390 tiles -> 128 (one round of 256);   M = 2066, N = 768 : 102 / 132 / 198 -> 64 (one round of 192)
        //   M = 2066, N = 1024 (ViT-L, B = 2): 136 / 176 / 264 -> 96 (one round of 224; 64-row tiles would need two)
        long best = -1; int best_bm = 128;
        for (int cand : {128, 96, 64}) {
          const long tiles = (long)((a.M + cand - 1) / cand) * (a.N / 128);
          const long cost = ((tiles + 255) / 256) * (cand + 128);
          if (best < 0 || cost < best) { best = cost; best_bm = cand; }
        }
        static const int force_bm = diag_env("GAVIKO_HIP_GEMM_BM") ? atoi(diag_env("GAVIKO_HIP_GEMM_BM")) : 0;      // A/B switch: 128 / 96 / 64
        if (force_bm == 128 || force_bm == 96 || force_bm == 64) best_bm = force_bm;
        else if (t128 <= t64_hi && best_bm == 128) best_bm = 64;                                           // (the round-2 rule, kept for the shapes it was tuned on)
        // (eight-wave split-k forms of this tile and a stream-K launch were built and measured slower in rounds 2-3: DESIGN.md 7b.1, 7c.5b)
        tile = best_bm == 64 ? 3064128 : best_bm == 96 ? 3096128 : 3128128;
      }
      else if (n768 == 128) tile = 128128;
      else if (n768 == 3064) tile = 3064128;
    }
  }
  if (a.drop_thresh != 0u) {
    if constexpr (EPI == GVK_EPI_BIAS_RES_F32 || EPI == GVK_EPI_BIAS_GELU_BF16 || EPI == GVK_EPI_GELU_BWD_BF16) {
      switch (tile) {
        case 128128: return launch_gemm<128, 128, EPI, true>(a, stream);
        case 128064: return launch_gemm<128, 64, EPI, true>(a, stream);
        case 64128: return launch_gemm<64, 128, EPI, true>(a, stream);
        case 64064: return launch_gemm<64, 64, EPI, true>(a, stream);
        default: return set_error(-2, "gvk_gemm_nt_bf16: unsupported tile %d", tile);
      }
    } else {
      return set_error(-2, "gvk_gemm_nt_bf16: drop_p > 0 is supported by BIAS_RES_F32, BIAS_GELU_BF16 and GELU_BWD_BF16 only");
    }
  }
  switch (tile) {
    case 8256256: return launch_gemm8p(a, EPI, 0, stream);      // eight-phase kernel, LDS-DMA issued in the load sections (default)
    case 7256256: return launch_gemm8p(a, EPI, 1, stream);      // ... issued inside the MFMA clusters (2546 vs 2426 cycles per k-tile)
    case 2128128:                                                            // 128 x 128, two stages, every tile's K loop cut into pieces (two workgroups share a CU)
      if (a.sk_part == nullptr) return set_error(-2, "gvk_gemm_nt_bf16: tile 2128128 needs splitk_ws");
      return launch_gemm<128, 128, EPI, false, 2, 4, true>(a, stream);
    case 3128128: return launch_gemm<128, 128, EPI, false, 3>(a, stream);
    case 3064128: return launch_gemm<64, 128, EPI, false, 3>(a, stream);     // 64 x 128 with three stages (A/B switch GAVIKO_HIP_GEMM_N768=3064)
    case 4064128:                                                            // ... four stages: the strided row-panel launches (a few tiles, 12-49 k-steps each),
      if (a.sk_part != nullptr) return launch_gemm<64, 128, EPI, false, 4, 4, true>(a, stream);      // their k loops cut into pieces when a workspace is given
      return launch_gemm<64, 128, EPI, false, 4>(a, stream);
    case 3096128: return launch_gemm<96, 128, EPI, false, 3>(a, stream);     // 96 x 128 with three stages (M = 2066, N = 1024: 176 tiles in one round)
    case 256256:
      if constexpr (EPI == GVK_EPI_STORE_BF16 || EPI == GVK_EPI_BIAS_GELU_BF16 || EPI == GVK_EPI_GELU_BWD_BF16) return launch_gemm<256, 256, EPI, false, 2, 8>(a, stream);
      else return set_error(-2, "gvk_gemm_nt_bf16: the 256x256 tile is built for STORE_BF16, BIAS_GELU_BF16 and GELU_BWD_BF16");
    case 128128: return launch_gemm<128, 128, EPI>(a, stream);
    case 128064: return launch_gemm<128, 64, EPI>(a, stream);
    case 64128: return launch_gemm<64, 128, EPI>(a, stream);
    case 64064: return launch_gemm<64, 64, EPI>(a, stream);
    default: return set_error(-2, "gvk_gemm_nt_bf16: unsupported tile %d", tile);
  }
}

}  // namespace gvk

extern "C" int gvk_gemm_stat_parts(int N) { return N / 64; }

extern "C" int gvk_gemm_nt_bf16(const gvk_gemm_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d != nullptr, "gvk_gemm_nt_bf16: null descriptor");
  GVK_REQUIRE(d->a && d->w && d->M > 0 && d->N > 0 && d->K > 0, "gvk_gemm_nt_bf16: null operand or empty shape");
  GVK_REQUIRE(d->K % 64 == 0, "gvk_gemm_nt_bf16: K=%d must be a multiple of 64", d->K);
  GVK_REQUIRE(d->N % 64 == 0, "gvk_gemm_nt_bf16: N=%d must be a multiple of 64", d->N);
  GVK_REQUIRE(d->lda >= d->K && d->ldw >= d->K && d->lda % 8 == 0 && d->ldw % 8 == 0,
              "gvk_gemm_nt_bf16: lda/ldw must be >= K and multiples of 8");
  GVK_REQUIRE(d->ldo % 8 == 0 && d->ldo >= d->N, "gvk_gemm_nt_bf16: ldo=%d must be >= N and a multiple of 8 (16-byte row pieces)", d->ldo);
  if (d->tile != 0) {
    const int bn = d->tile % 1000;
    GVK_REQUIRE(d->tile < 1000000 || d->K % 32 == 0, "gvk_gemm_nt_bf16: K not a multiple of 32");
    GVK_REQUIRE(bn > 0 && d->N % bn == 0, "gvk_gemm_nt_bf16: N=%d not a multiple of the tile's BN", d->N);
  }
  GemmArgs a{};
  a.A = (const bf16*)d->a; a.W = (const bf16*)d->w; a.out0 = d->out0; a.out1 = d->out1;
  a.bias = d->bias; a.res = d->res; a.aux = (const bf16*)d->aux; a.pos = d->pos;
  a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldw = d->ldw; a.ldo = d->ldo;
  a.ldres = d->ldres; a.ldaux = d->ldaux; a.rows_in = d->rows_in; a.rows_out = d->rows_out; a.row_off = d->row_off;
  GVK_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f && (d->drop_p == 0.f || d->seed_ptr != nullptr), "gvk_gemm_nt_bf16: drop_p in [0,1) and a seed word");
  a.seed = d->seed; a.seed_ptr = (const unsigned long long*)d->seed_ptr; a.drop_thresh = drop_threshold_u32(d->drop_p);
  a.inv_keep = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
  GVK_REQUIRE(d->scale_cols == 0 || (d->epilogue == GVK_EPI_STORE_BF16 && d->scale_cols > 0 && d->scale_cols % 8 == 0 && d->scale_cols <= d->N),
              "gvk_gemm_nt_bf16: scale_cols=%d needs the STORE_BF16 epilogue, a multiple of 8 and <= N", d->scale_cols);
  a.scale_cols = d->scale_cols; a.col_scale = d->col_scale;
  GVK_REQUIRE(d->ln_mean == nullptr || (d->epilogue == GVK_EPI_STORE_BF16 && d->ln_rstd && d->ln_c1 && d->bias),
              "gvk_gemm_nt_bf16: the LayerNorm fold needs the STORE_BF16 epilogue, ln_rstd, ln_c1 and bias (= sum_c beta[c] W[n][c])");
  a.ln_mean = d->ln_mean; a.ln_rstd = d->ln_rstd; a.ln_c1 = d->ln_c1;
  GVK_REQUIRE(d->stat_part == nullptr || (d->epilogue == GVK_EPI_BIAS_RES_F32_BF16 && d->N % 128 == 0 && (d->tile == 0 || d->tile % 1000 == 128)),
              "gvk_gemm_nt_bf16: stat_part needs the BIAS_RES_F32_BF16 epilogue on 128-column tiles (64-column groups)");
  a.stat_part = d->stat_part; a.stat_pivot = d->stat_pivot;
  GVK_REQUIRE((d->m_panels == 0 && d->m_stride == 0) || (d->m_panels > 0 && d->m_stride > 0 && d->epilogue != GVK_EPI_PATCH_F32),
              "gvk_gemm_nt_bf16: m_panels=%d / m_stride=%d: both positive (or both 0), not with the PATCH epilogue", d->m_panels, d->m_stride);
  a.m_stride = d->m_stride; a.nbm = d->m_panels;
  GVK_REQUIRE(d->splitk_ws == nullptr || ((d->m_panels > 0 || d->tile == 2128128) && ((uintptr_t)d->splitk_ws & 255) == 0),
              "gvk_gemm_nt_bf16: splitk_ws goes with strided row panels or tile 2128128 (256-byte aligned)");
  GVK_REQUIRE(d->ksplit >= 0 && d->ksplit <= 8 && (d->ksplit == 0 || d->splitk_ws != nullptr), "gvk_gemm_nt_bf16: ksplit=%d: 0 (auto) .. 8, with splitk_ws", d->ksplit);
  a.sk_part = (float*)d->splitk_ws; a.sk_bytes = d->splitk_ws_bytes; a.ksplit = d->ksplit;
  GVK_REQUIRE(d->aux_is_grad == 0 || (d->drop_p == 0.f && ((d->epilogue == GVK_EPI_BIAS_GELU_BF16 && d->out0) || d->epilogue == GVK_EPI_GELU_BWD_BF16)),
              "gvk_gemm_nt_bf16: aux_is_grad goes with BIAS_GELU_BF16 (out0 set) / GELU_BWD_BF16 and no dropout");
  a.aux_grad = d->aux_is_grad;
  hipStream_t s = (hipStream_t)stream;
  switch (d->epilogue) {
    case GVK_EPI_STORE_BF16:
      GVK_REQUIRE(d->out0, "gemm STORE_BF16: out0 null");
      return dispatch_tile<GVK_EPI_STORE_BF16>(a, d->tile, s);
    case GVK_EPI_BIAS_RES_F32:
      GVK_REQUIRE(d->out0 && d->res && d->ldres >= d->N && d->ldres % 4 == 0, "gemm BIAS_RES_F32: out0/res");
      return dispatch_tile<GVK_EPI_BIAS_RES_F32>(a, d->tile, s);
    case GVK_EPI_BIAS_RES_F32_BF16:
      GVK_REQUIRE(d->out0 && d->out1 && d->res && d->ldres >= d->N && d->ldres % 4 == 0, "gemm BIAS_RES_F32_BF16: out0/out1/res");
      return dispatch_tile<GVK_EPI_BIAS_RES_F32_BF16>(a, d->tile, s);
    case GVK_EPI_BIAS_GELU_BF16:
      GVK_REQUIRE(d->out1, "gemm BIAS_GELU_BF16: out1 null");
      return dispatch_tile<GVK_EPI_BIAS_GELU_BF16>(a, d->tile, s);
    case GVK_EPI_PATCH_F32:
      GVK_REQUIRE(d->out0 && d->pos && d->rows_in > 0 && d->rows_out >= d->rows_in + d->row_off && d->ldo == d->N,
                  "gemm PATCH_F32: out0/pos/rows_in/rows_out/row_off inconsistent (ldo must equal N)");
      return dispatch_tile<GVK_EPI_PATCH_F32>(a, d->tile, s);
    case GVK_EPI_GELU_BWD_BF16:
      GVK_REQUIRE(d->out0 && d->aux && d->ldaux >= d->N && d->ldaux % 8 == 0, "gemm GELU_BWD_BF16: out0/aux");
      return dispatch_tile<GVK_EPI_GELU_BWD_BF16>(a, d->tile, s);
    case GVK_EPI_STORE_F32:
      GVK_REQUIRE(d->out0, "gemm STORE_F32: out0 null");
      return dispatch_tile<GVK_EPI_STORE_F32>(a, d->tile, s);
    case GVK_EPI_BIAS_RELU_BF16:
      GVK_REQUIRE(d->out0, "gemm BIAS_RELU_BF16: out0 null");
      return dispatch_tile<GVK_EPI_BIAS_RELU_BF16>(a, d->tile, s);
    case GVK_EPI_RELU_BWD_BF16:
      GVK_REQUIRE(d->out0 && d->aux && d->ldaux >= d->N && d->ldaux % 8 == 0, "gemm RELU_BWD_BF16: out0/aux");
      return dispatch_tile<GVK_EPI_RELU_BWD_BF16>(a, d->tile, s);
    default:
      return set_error(-2, "gvk_gemm_nt_bf16: unknown epilogue %d", d->epilogue);
  }
}
