// Flash-style multi-head self-attention forward for gfx950, head dim 64, bf16 operands / fp32 accumulate.
// Replaces vision_transformer.py:63-71 (q.k^T * scale -> softmax -> .v and both einops rearranges): the [B,h,T,T]
// score tensor is never materialised, only the per-row log-sum-exp is saved for the backward.
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 queries.
// S^T[key][query] = K.Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows from LDS, B = Q kept in registers), so a lane
// holds ONE query column: the online-softmax max/sum are lane-local plus one exchange between the two 32-lane
// halves.  The S^T accumulator is then reused in place as the B operand of O^T[d][query] += V^T.P^T (its k order
// is the accumulator row order; the V^T fragment is gathered in that same order with ds_read_b64_tr_b16 from the
// row-major [key][d] V tile), so nothing is transposed through memory.
// LDS tile rows are 64 bf16 = 128 B; 16-B chunk c of row r sits at chunk c ^ swz(r), swz(r) = bit1(r)<<2 | bits3:2(r)
// -- conflict-free for both the ds_read_b128 row reads and the 4x16 transposed reads.
#include "common.hpp"
#include "dropout.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

// A/B switch (compile-time, tools/gpu experiments): raise the wave's priority around its MFMA clusters so that, of the two waves a SIMD
// hosts (two workgroups per CU), the one in a matrix phase issues first and the other fills the gaps with its softmax VALU work
// tools/probe/probe_attn.hip compiles this file with GVK_STAMPS: shader-clock stamps of ONE steady-state key tile (kt == 4) of the four-wave kernel,
// written through dr.seed_ptr (unused without dropout) as uint64 [workgroup][wave][8].  Compiled out of the library.
#ifdef GVK_STAMPS
#define GVK_ASTAMP(k) if (kt == 4) st_[k] = __builtin_amdgcn_s_memtime();
#else
#define GVK_ASTAMP(k)
#endif
#ifdef GVK_ATTN_PRIO
#define GVK_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define GVK_PRIO(x)
#endif

__device__ __forceinline__ int swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

constexpr int kQB = 128;   // queries per workgroup
constexpr int kKB = 128;   // keys per staged tile (4 MFMA key blocks of 32): one barrier pair per 128 keys
constexpr int kTileBytes = kKB * 128;

// attention-probability dropout (vision_transformer.py:68, live for the unfrozen-backbone methods): the softmax statistics are taken
// of the undropped scores, the dropped and rescaled P feeds the P.V product; mask element (b*H + head, query, key) -- dropout.hpp
struct AttnDrop { unsigned long long seed; const unsigned long long* seed_ptr; unsigned int thresh; float inv_keep; };

// RS: K / V tiles staged through registers (global_load_dwordx4 at the top of a tile, ds_write_b128 in front of its closing barrier) instead
// of by LDS-DMA: issuing the eight 1-KiB LDS-DMA instructions of a tile holds the wave's instruction stream for ~650 cycles of a ~3900-cycle
// tile (tools/probe/probe_attn.py), and this kernel is bound by the SIMD's vector issue, not by latency.
template <bool DROP, bool RS>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, float* __restrict__ lse,
                                                       int T, int H, int ld_qkv, int ld_out, float scale_log2e, AttnDrop dr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K tile | V tile]
  int bh, qb;
  xcd_group_block(blockIdx.x, (T + kQB - 1) / kQB, gridDim.x / ((T + kQB - 1) / kQB), bh, qb);   // all query blocks of a (batch, head) on one XCD
  const int b = bh / H, head = bh - b * H, q0 = qb * kQB;
  const int lane = lane_id(), wave = wave_id();
  const int r31 = lane & 31, hh = lane >> 5;
  const int inner = H * 64;
  const bf16* base = qkv + (size_t)b * T * ld_qkv + head * 64;
  const bf16* kbase = base + inner;
  const bf16* vbase = base + 2 * inner;

  // Q fragments: B operand, col = query (lane&31), k = d = 16*ks + 8*hh + j
  const int qrow = min(q0 + wave * 32 + r31, T - 1);
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(base + (size_t)qrow * ld_qkv + 16 * ks + 8 * hh);

  // K / V staging: lane-private source pointers advanced by one tile per call (the straightforward form recomputed a clamped 64-bit
  // address per 16-byte piece: ~100 VALU instructions per tile in a kernel whose softmax is VALU-bound).  Only the last tile can
  // reach past the sequence; it clamps its rows to T-1 (finite data that the key mask then ignores).
  const int rsub = lane >> 3, slot = lane & 7;
  const bf16* kp[kKB / 32];
  const bf16* vp[kKB / 32];
#pragma unroll
  for (int r = 0; r < kKB / 32; ++r) {
    const int row = r * 32 + wave * 8 + rsub;
    kp[r] = kbase + (size_t)row * ld_qkv + ((slot ^ swz(row)) << 3);
    vp[r] = kp[r] + inner;
  }
  const size_t tile_step = (size_t)kKB * ld_qkv;
  const int nkt = (T + kKB - 1) / kKB;
  auto stage = [&](int buf, int kt) {
    char* sK = smem + buf * 2 * kTileBytes;
    char* sV = sK + kTileBytes;
    if (kt == nkt - 1) {
#pragma unroll
      for (int r = 0; r < kKB / 32; ++r) {
        const int row = r * 32 + wave * 8 + rsub;
        const int over = max(kt * kKB + row - (T - 1), 0);            // rows past the end step back to row T-1
        glds16(kp[r] - (size_t)over * ld_qkv, sK + (r * 32 + wave * 8) * 128);
        glds16(vp[r] - (size_t)over * ld_qkv, sV + (r * 32 + wave * 8) * 128);
      }
    } else {
#pragma unroll
      for (int r = 0; r < kKB / 32; ++r) {
        glds16(kp[r], sK + (r * 32 + wave * 8) * 128);
        glds16(vp[r], sV + (r * 32 + wave * 8) * 128);
        kp[r] += tile_step;
        vp[r] += tile_step;
      }
    }
  };
  // register staging (RS): the same source addresses and the same LDS image, in two halves
  [[maybe_unused]] u32x4 kreg[kKB / 32], vreg[kKB / 32];
  [[maybe_unused]] auto fetch = [&](int kt) {
#pragma unroll
    for (int r = 0; r < kKB / 32; ++r) {
      const int row = r * 32 + wave * 8 + rsub;
      const int over = (kt == nkt - 1) ? max(kt * kKB + row - (T - 1), 0) : 0;
      kreg[r] = *(const u32x4*)(kp[r] - (size_t)over * ld_qkv);
      vreg[r] = *(const u32x4*)(vp[r] - (size_t)over * ld_qkv);
      kp[r] += tile_step;
      vp[r] += tile_step;
    }
  };
  [[maybe_unused]] auto commit = [&](int buf) {
    char* sK = smem + buf * 2 * kTileBytes;
    char* sV = sK + kTileBytes;
#pragma unroll
    for (int r = 0; r < kKB / 32; ++r) {
      *(u32x4*)(sK + (r * 32 + wave * 8) * 128 + lane * 16) = kreg[r];
      *(u32x4*)(sV + (r * 32 + wave * 8) * 128 + lane * 16) = vreg[r];
    }
  };

  f32x16 ot[2];
  ot[0] = f32x16{};
  ot[1] = f32x16{};
  float m_run = -INFINITY, l_run = 0.f;
  [[maybe_unused]] unsigned int akey = 0u, qoff = 0u;
  if constexpr (DROP) {
    akey = attn_key(dr.seed + *dr.seed_ptr, b * H + head);
    qoff = (unsigned int)(q0 + wave * 32 + r31) * (unsigned int)T;
  }

#ifdef GVK_STAMPS
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  if constexpr (RS) { fetch(0); commit(0); } else { stage(0, 0); }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    GVK_ASTAMP(0)
    if (kt + 1 < nkt) {
      if constexpr (RS) fetch(kt + 1); else stage(buf ^ 1, kt + 1);
    }
    GVK_ASTAMP(1)
    const char* sK = smem + buf * 2 * kTileBytes;
    const char* sV = sK + kTileBytes;

    // ---- S^T = K . Q^T   (kKB/32 key blocks of 32).  The K fragments of block kb+1 are read from LDS while the four MFMAs of block kb
    //      run: left to itself the compiler issued every ds_read right before its MFMA and waited lgkmcnt(0) in between (16 exposed
    //      LDS round trips per tile).
    constexpr int NKB = kKB / 32;
    f32x16 st[NKB];
    bf16x8 kfr[2][4];
    auto load_k = [&](int kb, bf16x8 (&dst)[4]) {
      const int row = kb * 32 + r31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dst[ks] = *(const bf16x8*)(sK + row * 128 + (((2 * ks + hh) ^ swz(row)) << 4));
    };
    load_k(0, kfr[0]);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb + 1 < NKB) load_k(kb + 1, kfr[(kb + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      st[kb] = f32x16{};
      GVK_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb & 1][ks], qf[ks], st[kb], 0, 0, 0);
      GVK_PRIO(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    GVK_ASTAMP(2)
    // ---- online softmax in the log2 domain.  VALU budget per score: max, fma, exp2, add (the scale is folded into the fma,
    //      the key mask is applied on the last tile only) -- this block, not the MFMAs, was the largest share of the kernel.
    if (kt == nkt - 1) {                                  // wave-uniform: only the last tile can contain keys >= T
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * kKB + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          st[kb][r] = (key < T) ? st[kb][r] : -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;   // scale > 0 commutes with max
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: exp2(-inf) = 0
    m_run = m_new;
    // Software pipeline over the four 32-key blocks: the exp2 / row-sum / bf16 conversion of block kb+1 is issued right BEHIND the four
    // PV MFMAs of block kb, so the VALU work runs while the matrix pipe executes them (written as "all exps, then all MFMAs" hipcc
    // sank every exp in front of the one MFMA that consumes it: 1936 cycles for this section against 512 of MFMA + ~1200 of VALU;
    // tools/probe/probe_attn.py).  Two scores per v_pk_fma_f32 / v_pk_add_f32.
    f32x2 psum2 = {0.f, 0.f};
    const f32x2 sc2 = {scale_log2e, scale_log2e}, nm2 = {-m_new, -m_new};
    auto exp_block = [&](int kb) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 a = __builtin_elementwise_fma(f32x2{st[kb][r], st[kb][r + 1]}, sc2, nm2);
        f32x2 p2 = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        psum2 += p2;                                        // statistics of the undropped probabilities
        if constexpr (DROP) {
          const int key = kt * kKB + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;        // r even: r + 1 is the next key
          p2 = p2 * f32x2{attn_drop_scale(akey, qoff + (unsigned int)key, dr.thresh, dr.inv_keep),
                          attn_drop_scale(akey, qoff + (unsigned int)key + 1u, dr.thresh, dr.inv_keep)};
        }
        st[kb][r] = p2[0];
        st[kb][r + 1] = p2[1];
      }
    };
    auto cvt_block = [&](int kb, bf16x8 (&pf)[2]) {
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[sb][j] = (bf16)st[kb][8 * sb + j];
    };
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[db][r] *= alpha;

    GVK_ASTAMP(3)
    // ---- O^T += V^T . P^T : P^T straight from the accumulator registers as the B operand; V^T fragments of block kb+1 are gathered
    //      (ds_read_b64_tr_b16) before the MFMAs of block kb are issued
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
          const bf16x4 va = lds_read_tr16(sV + ra * 128 + ((chunk ^ swz(ra)) << 4) + (tp & 1) * 8);
          const bf16x4 vb = lds_read_tr16(sV + rb * 128 + ((chunk ^ swz(rb)) << 4) + (tp & 1) * 8);
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
      GVK_PRIO(1);
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int db = 0; db < 2; ++db) ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[kb & 1][sb][db], pfr[kb & 1][sb], ot[db], 0, 0, 0);
      GVK_PRIO(0);
      __builtin_amdgcn_sched_barrier(0);
      if (kb + 1 < NKB) {                                   // VALU of the next block, behind the four MFMAs just issued
        exp_block(kb + 1);
        cvt_block(kb + 1, pfr[(kb + 1) & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    l_run = l_run * alpha + (psum2[0] + psum2[1]);
    GVK_ASTAMP(4)
    if constexpr (RS) { if (kt + 1 < nkt) commit(buf ^ 1); }   // the other buffer was last read one barrier ago
    __syncthreads();
    GVK_ASTAMP(5)
  }
#ifdef GVK_STAMPS
  if (dr.seed_ptr != nullptr && lane == 0) {
    unsigned long long* o = (unsigned long long*)dr.seed_ptr + ((size_t)blockIdx.x * 4 + wave) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = st_[k];
  }
#endif

  // ---- epilogue: O[q][d] = O^T / l ; lse = ln(sum exp(s*scale))
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + wave * 32 + r31;
  if (q < T) {
    bf16* orow = out + ((size_t)b * T + q) * ld_out + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 o = {(bf16)(ot[db][4 * g4 + 0] * inv), (bf16)(ot[db][4 * g4 + 1] * inv), (bf16)(ot[db][4 * g4 + 2] * inv),
                    (bf16)(ot[db][4 * g4 + 3] * inv)};
        *(bf16x4*)(orow + db * 32 + 8 * g4 + 4 * hh) = o;
      }
    if (hh == 0 && lse != nullptr) lse[((size_t)b * H + head) * T + q] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.69314718055994530942f;
  }
}


// ---- eight-wave form: two groups of four waves share every staged K / V tile (256 queries per workgroup, one workgroup per CU) and run the
// same program ONE BARRIER APART.  A tile is two phases,
//     P1 = S^T = K.Q^T (16 MFMAs) | running max, rescale factor, O *= alpha        P2 = exp2, row sums, bf16 P | O^T += V^T.P^T (16 MFMAs)
// so while one group is in the MFMA half of a phase the other is in a VALU half (the 4-wave kernel leaves that overlap to two unrelated
// workgroups that happen to share a CU and mostly move in lockstep: 30 us = 0.17 of the bf16 peak).  Waves w and w+4 share a SIMD.
// LDS: THREE K|V buffers; tile t+2 is requested at the start of P2(t) -- its buffer (tile t-1) was last read in the other group's P2(t-1),
// which ended at the barrier this phase began with -- and every wave waits vmcnt(0) at the end of P1(t+1), one barrier (group 0) or two
// (group 1) before any wave reads the tile.
constexpr int kQB8 = 256;

template <bool DROP>
__global__ __launch_bounds__(512) void attn_fwd8_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, float* __restrict__ lse,
                                                        int T, int H, int ld_qkv, int ld_out, float scale_log2e, AttnDrop dr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3 buffers][K tile | V tile]
  int bh, qb;
  xcd_group_block(blockIdx.x, (T + kQB8 - 1) / kQB8, gridDim.x / ((T + kQB8 - 1) / kQB8), bh, qb);
  const int b = bh / H, head = bh - b * H, q0 = qb * kQB8;
  const int lane = lane_id(), wave = wave_id();
  const int grp = wave >> 2, sw = wave & 3;              // group (phase offset) and the wave's 32-query slice inside the group's 128
  const int r31 = lane & 31, hh = lane >> 5;
  const int inner = H * 64;
  const bf16* base = qkv + (size_t)b * T * ld_qkv + head * 64;
  const bf16* kbase = base + inner;

  const int qme = q0 + grp * 128 + sw * 32 + r31;
  const int qrow = min(qme, T - 1);
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(base + (size_t)qrow * ld_qkv + 16 * ks + 8 * hh);

  // staging: 8 waves x 8 rows = 64 rows per pass, two passes per 128-key tile and operand
  const int rsub = lane >> 3, slot = lane & 7;
  const int nkt = (T + kKB - 1) / kKB;
  auto stage = [&](int kt) {
    char* sK = smem + (kt % 3) * 2 * kTileBytes;
    char* sV = sK + kTileBytes;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = r * 64 + wave * 8 + rsub;
      const int key = min(kt * kKB + row, T - 1);                       // rows past the end re-read row T-1 (masked below)
      const bf16* kp = kbase + (size_t)key * ld_qkv + ((slot ^ swz(row)) << 3);
      glds16(kp, sK + (r * 64 + wave * 8) * 128);
      glds16(kp + inner, sV + (r * 64 + wave * 8) * 128);
    }
  };

  f32x16 ot[2];
  ot[0] = f32x16{};
  ot[1] = f32x16{};
  float m_run = -INFINITY, l_run = 0.f;
  [[maybe_unused]] unsigned int akey = 0u, qoff = 0u;
  if constexpr (DROP) {
    akey = attn_key(dr.seed + *dr.seed_ptr, b * H + head);
    qoff = (unsigned int)qme * (unsigned int)T;
  }
  constexpr int NKB = kKB / 32;

  stage(0);
  if (nkt > 1) stage(1);
  __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();            // the stagger
  for (int kt = 0; kt < nkt; ++kt) {
    const char* sK = smem + (kt % 3) * 2 * kTileBytes;
    const char* sV = sK + kTileBytes;
    // ================= P1: S^T = K . Q^T, then the running max and the rescale of O
    f32x16 st[NKB];
    bf16x8 kfr[2][4];
    auto load_k = [&](int kb, bf16x8 (&dst)[4]) {
      const int row = kb * 32 + r31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dst[ks] = *(const bf16x8*)(sK + row * 128 + (((2 * ks + hh) ^ swz(row)) << 4));
    };
    load_k(0, kfr[0]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb + 1 < NKB) load_k(kb + 1, kfr[(kb + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      st[kb] = f32x16{};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb & 1][ks], qf[ks], st[kb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (kt == nkt - 1) {                                  // wave-uniform: only the last tile can contain keys >= T
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * kKB + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          st[kb][r] = (key < T) ? st[kb][r] : -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[db][r] *= alpha;
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this wave's share of tile kt+1 (requested one phase ago) has landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= P2: probabilities, row sums, O^T += V^T . P^T
    if (kt + 2 < nkt) stage(kt + 2);
    f32x2 psum2 = {0.f, 0.f};
    const f32x2 sc2 = {scale_log2e, scale_log2e}, nm2 = {-m_new, -m_new};
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 a = __builtin_elementwise_fma(f32x2{st[kb][r], st[kb][r + 1]}, sc2, nm2);
        const f32x2 p2 = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        st[kb][r] = p2[0];
        st[kb][r + 1] = p2[1];
        psum2 += p2;
      }
    l_run = l_run * alpha + (psum2[0] + psum2[1]);
    if constexpr (DROP) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * kKB + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          st[kb][r] *= attn_drop_scale(akey, qoff + (unsigned int)key, dr.thresh, dr.inv_keep);
        }
    }
    const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    bf16x8 vfr[2][2];
    auto load_v = [&](int step, bf16x8 (&dst)[2]) {
      const int kb = step >> 1, sb = step & 1;
      const int key0 = kb * 32 + 16 * sb + 4 * (g >> 1);
      const int ra = key0 + tq, rb = key0 + 8 + tq;
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int chunk = db * 4 + 2 * (g & 1) + (tp >> 1);
        const bf16x4 va = lds_read_tr16(sV + ra * 128 + ((chunk ^ swz(ra)) << 4) + (tp & 1) * 8);
        const bf16x4 vb = lds_read_tr16(sV + rb * 128 + ((chunk ^ swz(rb)) << 4) + (tp & 1) * 8);
        dst[db] = bf16x8{va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
      }
    };
    load_v(0, vfr[0]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int step = 0; step < 2 * NKB; ++step) {
      if (step + 1 < 2 * NKB) load_v(step + 1, vfr[(step + 1) & 1]);
      const int kb = step >> 1, sb = step & 1;
      bf16x8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (bf16)st[kb][8 * sb + j];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int db = 0; db < 2; ++db) ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[step & 1][db], pf, ot[db], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();            // balance the barrier count

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qme < T) {
    bf16* orow = out + ((size_t)b * T + qme) * ld_out + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 o = {(bf16)(ot[db][4 * g4 + 0] * inv), (bf16)(ot[db][4 * g4 + 1] * inv), (bf16)(ot[db][4 * g4 + 2] * inv),
                    (bf16)(ot[db][4 * g4 + 3] * inv)};
        *(bf16x4*)(orow + db * 32 + 8 * g4 + 4 * hh) = o;
      }
    if (hh == 0 && lse != nullptr) lse[((size_t)b * H + head) * T + qme] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.69314718055994530942f;
  }
}

}  // namespace gvk

namespace gvk {
template <bool DROP>
static int launch_attn_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale, AttnDrop dr, hipStream_t stream) {
  // opt-in (GAVIKO_HIP_ATTN8=1): parity-green but no faster -- 32.8 vs 30.5 us isolated, 679 vs 684 volumes/s in the step (profiles/r02_pmc_attention.json:
  // both forms spend ~0.3 of their wave cycles issuing VALU, ~0.2 in MFMA and the rest waiting; forcing the MFMA / VALU halves of the two waves
  // of a SIMD apart by a barrier did not change that)
  static const bool use8 = getenv("GAVIKO_HIP_ATTN8") != nullptr && getenv("GAVIKO_HIP_ATTN8")[0] == '1';
  if (use8 && T > 128) {
    const int lds8 = 3 * 2 * kTileBytes;
    static bool attr8 = false;
    if (!attr8) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd8_kernel<DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds8);
      if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(attn_fwd8): %s", hipGetErrorString(e));
      attr8 = true;
    }
    GVK_LAUNCH(attn_fwd8_kernel<DROP>, dim3(((T + kQB8 - 1) / kQB8) * H * B), dim3(512), lds8, stream, (const bf16*)qkv,
                       (bf16*)out, lse, T, H, ld_qkv, ld_out, scale * 1.44269504088896340736f, dr);
    return check_launch("attention_fwd8_bf16");
  }
  const int lds = 2 * 2 * kTileBytes;
  static const bool rs = getenv("GAVIKO_HIP_ATTN_RS") != nullptr && getenv("GAVIKO_HIP_ATTN_RS")[0] == '1';   // opt-in: measured 32.3 vs 31.0 us (no gain)
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<DROP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<DROP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(attn_fwd): %s", hipGetErrorString(e));
    attr = true;
  }
  const dim3 grid(((T + kQB - 1) / kQB) * H * B);
  if (rs) GVK_LAUNCH((attn_fwd_kernel<DROP, true>), grid, dim3(256), lds, stream, (const bf16*)qkv, (bf16*)out, lse, T, H, ld_qkv, ld_out, scale * 1.44269504088896340736f, dr);
  else GVK_LAUNCH((attn_fwd_kernel<DROP, false>), grid, dim3(256), lds, stream, (const bf16*)qkv, (bf16*)out, lse, T, H, ld_qkv, ld_out, scale * 1.44269504088896340736f, dr);
  return check_launch("attention_fwd_bf16");
}
}  // namespace gvk

extern "C" int gvk_attention_fwd_bf16_dropout(const void* qkv, void* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale,
                                              float drop_p, uint64_t seed, const void* seed_ptr, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(qkv && out, "gvk_attention_fwd_bf16: null pointer");
  GVK_REQUIRE(B > 0 && T > 0 && H > 0, "gvk_attention_fwd_bf16: empty shape");
  GVK_REQUIRE(ld_qkv >= 3 * H * 64 && ld_qkv % 8 == 0 && ld_out >= H * 64 && ld_out % 4 == 0,
              "gvk_attention_fwd_bf16: head dim is fixed at 64; ld_qkv=%d ld_out=%d inconsistent with H=%d", ld_qkv, ld_out, H);
  GVK_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed_ptr != nullptr), "gvk_attention_fwd_bf16: drop_p in [0,1) and a seed word");
  GVK_REQUIRE(drop_p == 0.f || (int64_t)T * T < (int64_t)1 << 32, "gvk_attention_fwd_bf16: the dropout mask index (query*T + key) is 32-bit");
  const AttnDrop dr{seed, (const unsigned long long*)seed_ptr, drop_threshold_u32(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f};
  if (drop_p > 0.f) return launch_attn_fwd<true>(qkv, out, lse, B, T, H, ld_qkv, ld_out, scale, dr, (hipStream_t)stream);
  return launch_attn_fwd<false>(qkv, out, lse, B, T, H, ld_qkv, ld_out, scale, dr, (hipStream_t)stream);
}

extern "C" int gvk_attention_fwd_bf16(const void* qkv, void* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale,
                                      void* stream) {
  return gvk_attention_fwd_bf16_dropout(qkv, out, lse, B, T, H, ld_qkv, ld_out, scale, 0.f, 0, nullptr, stream);
}
