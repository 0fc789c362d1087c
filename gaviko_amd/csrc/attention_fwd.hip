// Flash-style multi-head self-attention forward for gfx950, head dim 64, bf16 operands / fp32 accumulate.
// Replaces vision_transformer.py:63-71 (q.k^T * scale -> softmax -> .v and both einops rearranges): the [B,h,T,T]
// score tensor is never materialised, only the per-row log-sum-exp is saved for the backward.
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 queries; two workgroups share a CU.
// S^T[key][query] = K.Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows from LDS, B = Q kept in registers), so a lane
// holds ONE query column: the online-softmax max is lane-local plus one exchange between the two 32-lane halves.
// The S^T accumulator is then reused in place as the B operand of O^T[d][query] += V^T.P^T (its k order is the
// accumulator row order; the V^T fragment is gathered in that same order with ds_read_b64_tr_b16 from the row-major
// [key][d] V tile), so nothing is transposed through memory.
//
// Round 3: the kernel was bound by the SIMD's VECTOR ISSUE, not by the matrix pipe (356 VALU instructions per 128-key tile
// and wave beside 32 MFMAs; profiles/r02_attention_fwd_stamps.txt), so the per-score VALU work moved onto the matrix pipe:
//   * the q block ARRIVES pre-scaled by scale*log2(e): the qkv projection's epilogue applies the factor in fp32 before its one rounding to
//     bf16 (gvk_gemm_desc.scale_cols; gvk_qkv_prescale_bf16 for callers that hold a raw q block), so a score needs no multiply and the
//     forward and both backward passes see bit-identical Q' operands (their P must agree: the backward subtracts the forward's lse);
//   * the running-max subtraction and the key mask are a FIFTH MFMA per 32-key block over an augmented contraction:
//     A_aug[key] = [1, 1, 1, key >= T, 0...], B_aug[query] = [-m_hi, -m_mid, -m_lo, -3e38, 0...] (m split into three bf16
//     pieces = exact to fp32), i.e. S' = K.Q'^T - m  arrives in the accumulator ready for v_exp_f32 -- no v_fma, no v_cndmask;
//   * the running maximum is only raised when a tile exceeds it by more than kThr (log2 units): the O / l rescale and the
//     rebuild of B_aug sit in a wave-uniform slow path that a few tiles per workgroup take (P <= 2^kThr stays harmless: bf16
//     keeps fp32's exponent range and the accumulators are fp32);
//   * row sums in fp32 from the unrounded probabilities (the matrix-pipe form -- a third "d block" of the PV product with an all-ones A
//     operand, VAR bit 0 -- is 0.8 % faster in the step but sums ROUNDED probabilities; it is kept in the diag build, see kAttnVarDefault);
//   * 96-key tiles when they pad the sequence less than 128-key tiles do (T = 1033: 11 x 96 = 1056 keys instead of 1152);
//   * O leaves through LDS as whole 128-byte rows (16 B per lane) instead of 8-byte pieces at a 1.5 KB row stride.
// LDS tile rows are 64 bf16 = 128 B; 16-B chunk c of row r sits at chunk c ^ attn_swz(r), attn_swz(r) = bit1(r)<<2 | bits3:2(r)
// -- conflict-free for both the ds_read_b128 row reads and the 4x16 transposed reads.
#include "attention_common.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

// tools/probe/probe_attn.hip compiles this file with GVK_STAMPS: shader-clock stamps of ONE steady-state key tile (kt == 4),
// plus kernel entry / loop start / loop end / kernel end (slots 6..9), written through dr.seed_ptr (unused without dropout) as uint64
// [workgroup][wave][12].  Compiled out of the library.
#ifdef GVK_STAMPS
#define GVK_ASTAMP(k) if (kt == 4) st_[k] = __builtin_amdgcn_s_memtime();
#define GVK_KSTAMP(k) st_[k] = __builtin_amdgcn_s_memtime();
#else
#define GVK_ASTAMP(k)
#define GVK_KSTAMP(k)
#endif

constexpr int kQB = 128;          // queries per workgroup
constexpr float kThr = 8.0f;      // the running max is raised only when a tile exceeds it by more than this (log2 units)

// VAR bit 0: row sums on the matrix pipe (ones operand);  bit 1: the next tile's LDS-DMA spread behind the S^T MFMA groups
template <int KB, bool DROP, int VAR>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, float* __restrict__ lse,
                                                       int T, int H, int ld_qkv, int ld_out, AttnDrop dr) {
  constexpr int NKB = KB / 32;                 // MFMA key blocks per staged tile
  constexpr int kTileBytes = KB * 128;
  constexpr bool ONES = (VAR & 1) != 0 && !DROP;
  constexpr bool SPREAD = (VAR & 2) != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K tile | V tile]
#ifdef GVK_STAMPS
  unsigned long long st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  GVK_KSTAMP(6)
  int bh, qb;
  xcd_group_block(blockIdx.x, (T + kQB - 1) / kQB, gridDim.x / ((T + kQB - 1) / kQB), bh, qb);   // all query blocks of a (batch, head) on one XCD
  const int b = bh / H, head = bh - b * H, q0 = qb * kQB;
  const int lane = lane_id(), wave = wave_id();
  const int r31 = lane & 31, hh = lane >> 5;
  const int inner = H * 64;
  const bf16* base = qkv + (size_t)b * T * ld_qkv + head * 64;
  // a wave whose 32 query rows all lie past the sequence (the last query block of T = 1033 has 9 real rows) only stages tiles
  const bool active = q0 + wave * 32 < T;

  // Q' fragments (q * scale * log2(e), see the header): B operand, col = query (lane&31), k = d = 16*ks + 8*hh + j
  const int qrow = min(q0 + wave * 32 + r31, T - 1);
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(base + (size_t)qrow * ld_qkv + 16 * ks + 8 * hh);

  // K / V staging by LDS-DMA through a buffer resource: the per-lane byte offsets of a tile's pieces never change (NKB VGPRs), the tile
  // advances through the instruction's SCALAR offset -- no vector address arithmetic inside the loop (the 64-bit per-lane pointers of the
  // previous form cost 28 VALU instructions per tile).  Only the last tile can reach past the sequence; its pieces use a second, clamped
  // offset set (rows past the end step back to row T-1: finite data that the key mask then ignores), so every access is in bounds.
  const int rsub = lane >> 3, slot = lane & 7;
  const int nkt = (T + KB - 1) / KB;
  const int nB = (int)gridDim.x / (((T + kQB - 1) / kQB) * H);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, nB * T * ld_qkv * 2, 0x00020000);
  int vo[NKB], vo_last[NKB];
#pragma unroll
  for (int r = 0; r < NKB; ++r) {
    const int row = r * 32 + wave * 8 + rsub;
    const int col = inner + head * 64 + ((slot ^ attn_swz(row)) << 3);
    vo[r] = ((b * T + row) * ld_qkv + col) * 2;
    const int over = max((nkt - 1) * KB + row - (T - 1), 0);
    vo_last[r] = ((b * T + row - over) * ld_qkv + col) * 2;
  }
  const int tile_step = KB * ld_qkv * 2;       // bytes
  // one 32-row slice (K and V) of tile kt into buffer buf
  auto stage_piece = [&](int buf, int kt, int r) {
    char* sK = smem + buf * 2 * kTileBytes;
    char* sV = sK + kTileBytes;
    const int v = (kt == nkt - 1) ? vo_last[r] : vo[r];
    const int so = kt * tile_step;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (GVK_LDS void*)(sK + (r * 32 + wave * 8) * 128), 16, v, so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (GVK_LDS void*)(sV + (r * 32 + wave * 8) * 128), 16, v, so + inner * 2, 0, 0);
  };

  f32x16 ot[2];
  ot[0] = f32x16{};
  ot[1] = f32x16{};
  [[maybe_unused]] f32x16 lt = f32x16{};       // ONES: every register of every lane holds the query's running row sum
  float m_run = 0.f;                           // log2 domain; set by the first tile
  [[maybe_unused]] float l_run = 0.f;
  bf16x8 qa = aug_const(0.f, true, 0.f, hh);     // constant side of the augmented MFMA: [-m pieces, -3e38 | 0...]
  [[maybe_unused]] unsigned int akey = 0u, qoff = 0u;
  if constexpr (DROP) {
    akey = attn_key(dr.seed + *dr.seed_ptr, b * H + head);
    qoff = (unsigned int)(q0 + wave * 32 + r31) * (unsigned int)T;
  }
  [[maybe_unused]] const bf16 one_ = (bf16)1.0f;
  [[maybe_unused]] const bf16x8 ones = {one_, one_, one_, one_, one_, one_, one_, one_};

#pragma unroll
  for (int r = 0; r < NKB; ++r) stage_piece(0, 0, r);
  __syncthreads();
  GVK_KSTAMP(7)
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    GVK_ASTAMP(0)
    if constexpr (!SPREAD) {
      if (kt + 1 < nkt) {
#pragma unroll
        for (int r = 0; r < NKB; ++r) stage_piece(buf ^ 1, kt + 1, r);
      }
    }
    GVK_ASTAMP(1)
    if (active) {
      const char* sK = smem + buf * 2 * kTileBytes;
      const char* sV = sK + kTileBytes;

      // ---- S' = K . Q'^T - m   (NKB key blocks of 32, five MFMAs each).  The K fragments of block kb+1 are read from LDS while the
      //      MFMAs of block kb run.
      f32x16 st[NKB];
      bf16x8 kfr[2][4];
      auto load_k = [&](int kb, bf16x8 (&dst)[4]) {
        const int row = kb * 32 + r31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dst[ks] = *(const bf16x8*)(sK + row * 128 + (((2 * ks + hh) ^ attn_swz(row)) << 4));
      };
      load_k(0, kfr[0]);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) load_k(kb + 1, kfr[(kb + 1) & 1]);
        const bf16x8 ka = aug_sel_first(kt * KB + kb * 32 + r31 >= T, hh);      // selector side: [1, 1, 1, key >= T, 0...]
        __builtin_amdgcn_sched_barrier(0);
        st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qa, f32x16{}, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb & 1][ks], qf[ks], st[kb], 0, 0, 0);
        if constexpr (SPREAD) {
          if (kt + 1 < nkt) stage_piece(buf ^ 1, kt + 1, kb);       // behind this block's MFMAs: the DMA issue overlaps their execution
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      GVK_ASTAMP(2)
      // ---- running maximum (log2 domain).  d = tile max of S' = (tile max of the scores) - m_run
      float d = st[0][0];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = (kb == 0 ? 1 : 0); r < 16; ++r) d = fmaxf(d, st[kb][r]);
      d = half_max(d);
      if (kt == 0 || __builtin_amdgcn_ballot_w64(d > kThr) != 0ull) {
        // slow path (wave-uniform): raise the maximum, bring this tile's S' and the accumulated O / l to the new scale.  Tile 0 takes
        // it with m_run = 0 as the subtracted value and O = l = 0.
        const float m_new = (kt == 0) ? d : m_run + fmaxf(d, 0.f);
        const float delta = m_new - m_run;
        const float alpha = (kt == 0) ? 0.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[kb][r] -= delta;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[db][r] *= alpha;
        if constexpr (ONES) {
#pragma unroll
          for (int r = 0; r < 16; ++r) lt[r] *= alpha;
        } else {
          l_run *= alpha;
        }
        m_run = m_new;
        qa = aug_const(m_run, true, 0.f, hh);
      }
      // ---- probabilities and O^T += V^T . P^T, software-pipelined over the 32-key blocks: the exp2 / conversion of block kb+1 is issued
      //      right BEHIND the MFMAs of block kb, so the VALU work runs while the matrix pipe executes them.  P^T goes straight from the
      //      accumulator registers into the B operand; V^T fragments of block kb+1 are gathered (ds_read_b64_tr_b16) before the MFMAs of
      //      block kb are issued.
      [[maybe_unused]] float psum = 0.f;
      auto exp_block = [&](int kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = __builtin_amdgcn_exp2f(st[kb][r]);
          if constexpr (!ONES) psum += p;                     // statistics of the undropped probabilities
          if constexpr (DROP) {
            const int key = kt * KB + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            p *= attn_drop_scale(akey, qoff + (unsigned int)key, dr.thresh, dr.inv_keep);
          }
          st[kb][r] = p;
        }
      };
      auto cvt_block = [&](int kb, bf16x8 (&pf)[2]) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[sb][j] = (bf16)st[kb][8 * sb + j];
      };
      GVK_ASTAMP(3)
      const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
      bf16x8 vfr[2][2][2], pfr[2][2];                         // [block parity][16-key step][d half], [block parity][16-key step]
      auto load_v = [&](int kb, bf16x8 (&dst)[2][2]) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          const int key0 = kb * 32 + 16 * sb + 4 * (g >> 1);   // lane half hh == g>>1
          const int ra = key0 + tq, rb = key0 + 8 + tq;
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const int chunk = db * 4 + 2 * (g & 1) + (tp >> 1);
            const bf16x4 va = lds_read_tr16(sV + ra * 128 + ((chunk ^ attn_swz(ra)) << 4) + (tp & 1) * 8);
            const bf16x4 vb = lds_read_tr16(sV + rb * 128 + ((chunk ^ attn_swz(rb)) << 4) + (tp & 1) * 8);
            dst[sb][db] = bf16x8{va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
          }
        }
      };
      load_v(0, vfr[0]);
      exp_block(0);
      cvt_block(0, pfr[0]);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) load_v(kb + 1, vfr[(kb + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
          for (int db = 0; db < 2; ++db) ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[kb & 1][sb][db], pfr[kb & 1][sb], ot[db], 0, 0, 0);
          if constexpr (ONES) lt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pfr[kb & 1][sb], lt, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < NKB) {                                   // VALU of the next block, behind the MFMAs just issued
          exp_block(kb + 1);
          cvt_block(kb + 1, pfr[(kb + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (!ONES) l_run += psum;
    } else if constexpr (SPREAD) {
      if (kt + 1 < nkt) {
#pragma unroll
        for (int r = 0; r < NKB; ++r) stage_piece(buf ^ 1, kt + 1, r);
      }
    }
    GVK_ASTAMP(4)
    __syncthreads();
    GVK_ASTAMP(5)
  }
  GVK_KSTAMP(8)

  // ---- epilogue: O[q][d] = O^T / l through a wave-private 4 KB LDS image (32 rows of 128 B, 16-B chunk c of row q at c ^ (q & 7)),
  //      then whole rows out: 8 lanes x 16 B per row, 8 rows per store instruction.  lse = ln(sum exp(s*scale))
  if (!active) return;                         // (the last barrier of the loop is behind every wave: the K / V buffers are free)
  float l_tot;
  if constexpr (ONES) l_tot = lt[0];
  else l_tot = half_sum(l_run);
  const int qw = q0 + wave * 32;
  store_rows_t(ot, 1.0f / l_tot, smem + wave * 4096, out + ((size_t)b * T + qw) * ld_out + head * 64, (size_t)ld_out, T - qw, lane);
  const int q = qw + r31;
  if (q < T && hh == 0 && lse != nullptr) lse[((size_t)b * H + head) * T + q] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.69314718055994530942f;
#ifdef GVK_STAMPS
  __builtin_amdgcn_s_waitcnt(0);                     // stores retired
  GVK_KSTAMP(9)
  if (dr.seed_ptr != nullptr && lane == 0) {
    unsigned long long* o = (unsigned long long*)dr.seed_ptr + ((size_t)blockIdx.x * 4 + wave) * 12;
#pragma unroll
    for (int k = 0; k < 12; ++k) o[k] = st_[k];
  }
#endif
}

}  // namespace gvk

namespace gvk {
// diagnostics (tools/bench_attn.py): kernel variant forced by the environment; the library default is what launch_attn_fwd picks
static int attn_var() { return diag_env("GAVIKO_HIP_ATTN_VAR") ? atoi(diag_env("GAVIKO_HIP_ATTN_VAR")) : -1; }   // read per launch: one process can A/B
static int attn_kb() { return getenv("GAVIKO_HIP_ATTN_KB") ? atoi(getenv("GAVIKO_HIP_ATTN_KB")) : 0; }

template <int KB, bool DROP, int VAR>
static int launch_attn_fwd_t(const void* qkv, void* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale, AttnDrop dr, hipStream_t stream) {
  constexpr int lds = 2 * 2 * KB * 128;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<KB, DROP, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(attn_fwd): %s", hipGetErrorString(e));
    attr = true;
  }
  const dim3 grid(((T + kQB - 1) / kQB) * H * B);
  GVK_LAUNCH((attn_fwd_kernel<KB, DROP, VAR>), grid, dim3(256), lds, stream, (const bf16*)qkv, (bf16*)out, lse, T, H, ld_qkv, ld_out, dr);
  return check_launch("attention_fwd_bf16");
}

// Shipped: VAR 0, row sums added in fp32 from the UNROUNDED probabilities -- lse then agrees with float64 to 1.3e-6.  Row sums on the matrix
// pipe (VAR bit 0) are 0.8 % faster in the step (724.1 / 725.7 / 725.5 against 719.9 / 718.7 / 719.9 volumes/s, same box, interleaved) but
// add up the bf16-ROUNDED probabilities: on a peaked row lse then carries the 2^-9 rounding of the dominating term (measured 2.2e-3 at
// amp 2.5, 2-6e-4 at amp 1; tests/test_kernels_gpu.py::test_attention_fwd), which the backward's P = exp2(s - lse) turns into a per-row
// scale error of the same size.  Parity first: that variant stays in the diag build.  The spread LDS-DMA (bit 1) is neutral in the step
// (719.4 / 719.6 / 720.1) although 3 % faster in isolation.
constexpr int kAttnVarDefault = 0;

template <bool DROP>
static int launch_attn_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale, AttnDrop dr, hipStream_t stream) {
  // key tile: the size that pads the sequence less (T = 1033: 11 x 96 = 1056 against 9 x 128 = 1152); ties go to the larger tile
  const int p96 = (T + 95) / 96 * 96, p128 = (T + 127) / 128 * 128;
  int kb = p96 < p128 ? 96 : 128;
  if (attn_kb() == 96 || attn_kb() == 128) kb = attn_kb();
  if constexpr (DROP) {
    if (kb == 96) return launch_attn_fwd_t<96, true, kAttnVarDefault>(qkv, out, lse, B, T, H, ld_qkv, ld_out, scale, dr, stream);
    return launch_attn_fwd_t<128, true, kAttnVarDefault>(qkv, out, lse, B, T, H, ld_qkv, ld_out, scale, dr, stream);
  } else {
    const int var = attn_var() >= 0 ? attn_var() : kAttnVarDefault;
#define GVK_ATTN_CASE(KB_, V_) if (kb == KB_ && var == V_) return launch_attn_fwd_t<KB_, false, V_>(qkv, out, lse, B, T, H, ld_qkv, ld_out, scale, dr, stream);
    GVK_ATTN_CASE(96, kAttnVarDefault) GVK_ATTN_CASE(128, kAttnVarDefault)
#ifdef GVK_DIAG                                          // the other variants (row sums on the matrix pipe, spread LDS-DMA): measurement build only
    GVK_ATTN_CASE(96, 0) GVK_ATTN_CASE(96, 1) GVK_ATTN_CASE(96, 2) GVK_ATTN_CASE(96, 3)
    GVK_ATTN_CASE(128, 0) GVK_ATTN_CASE(128, 1) GVK_ATTN_CASE(128, 2) GVK_ATTN_CASE(128, 3)
#endif
#undef GVK_ATTN_CASE
    return set_error(-2, "gvk_attention_fwd_bf16: no kernel variant %d for key tile %d", var, kb);
  }
}
}  // namespace gvk

namespace gvk {
// x[m][n] *= s for n < cols (in place, fp32 multiply, one rounding): what gvk_gemm_desc.scale_cols does inside the qkv projection, for callers
// that hold a raw q block.  8 elements per thread.
__global__ __launch_bounds__(256) void qkv_prescale_kernel(bf16* __restrict__ x, int rows, int cols, int ld, float s) {
  const int per = cols >> 3;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)rows * per) return;
  const int m = (int)(i / per), c = (int)(i - (long long)m * per) << 3;
  bf16x8 v = *(bf16x8*)(x + (size_t)m * ld + c);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] * s);
  *(bf16x8*)(x + (size_t)m * ld + c) = v;
}
}  // namespace gvk

extern "C" int gvk_qkv_prescale_bf16(void* qkv, int rows, int H, int ld_qkv, float scale, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(qkv && rows > 0 && H > 0 && ld_qkv >= 3 * H * 64 && ld_qkv % 8 == 0, "gvk_qkv_prescale_bf16: bad arguments");
  const long long n = (long long)rows * (H * 8);
  GVK_LAUNCH(qkv_prescale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16*)qkv, rows, H * 64, ld_qkv,
             scale * 1.44269504088896340736f);
  return check_launch("qkv_prescale_bf16");
}

extern "C" int gvk_attention_fwd_bf16_dropout(const void* qkv, void* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale,
                                              float drop_p, uint64_t seed, const void* seed_ptr, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(qkv && out, "gvk_attention_fwd_bf16: null pointer");
  GVK_REQUIRE(B > 0 && T > 0 && H > 0, "gvk_attention_fwd_bf16: empty shape");
  GVK_REQUIRE(ld_qkv >= 3 * H * 64 && ld_qkv % 8 == 0 && ld_out >= H * 64 && ld_out % 8 == 0,
              "gvk_attention_fwd_bf16: head dim is fixed at 64; ld_qkv=%d ld_out=%d inconsistent with H=%d (16-byte rows)", ld_qkv, ld_out, H);
  GVK_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed_ptr != nullptr), "gvk_attention_fwd_bf16: drop_p in [0,1) and a seed word");
  GVK_REQUIRE(drop_p == 0.f || (int64_t)T * T < (int64_t)1 << 32, "gvk_attention_fwd_bf16: the dropout mask index (query*T + key) is 32-bit");
  GVK_REQUIRE((int64_t)B * T * ld_qkv * 2 < (int64_t)1 << 31, "gvk_attention_fwd_bf16: the qkv tensor must stay below 2 GiB (32-bit buffer offsets)");
  const AttnDrop dr{seed, (const unsigned long long*)seed_ptr, drop_threshold_u32(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f};
  if (drop_p > 0.f) return launch_attn_fwd<true>(qkv, out, lse, B, T, H, ld_qkv, ld_out, scale, dr, (hipStream_t)stream);
  return launch_attn_fwd<false>(qkv, out, lse, B, T, H, ld_qkv, ld_out, scale, dr, (hipStream_t)stream);
}

extern "C" int gvk_attention_fwd_bf16(const void* qkv, void* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale,
                                      void* stream) {
  return gvk_attention_fwd_bf16_dropout(qkv, out, lse, B, T, H, ld_qkv, ld_out, scale, 0.f, 0, nullptr, stream);
}
