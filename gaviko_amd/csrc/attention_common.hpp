// Helpers shared by the bf16 flash-attention forward and backward kernels (attention_fwd.hip, attention_bwd.hip).
#pragma once
#include "common.hpp"
#include "dropout.hpp"

namespace gvk {

// LDS tile rows are 64 bf16 = 128 B; 16-B chunk c of row r sits at chunk c ^ attn_swz(r): conflict-free for both the ds_read_b128 row
// reads and the 4x16 transposed reads (ds_read_b64_tr_b16)
__device__ __forceinline__ int attn_swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// LDS-DMA through inline asm (the one-pass backward only).  hipcc's wait-count pass cannot tell LDS buffers apart: behind a
// `buffer_load ... lds` it KNOWS of, every later LDS read first waits for that load, and its counted waits for loop-invariant register
// loads land inside the loop, where they drain whatever else is in flight.  In a loop that also carries global loads and stores of a
// hand-off that turns into a full s_waitcnt vmcnt(0) in front of every tile's first fragment reads.  Issued from asm the compiler does
// not see the LDS write; the kernel drains the queue itself (GVK_DMA_DRAIN) in front of the barrier that hands a tile to the other
// waves.  (Measured on the forward and the two-pass backward, which carry nothing else: no difference, they keep the builtin.)
// lds_wave_base: wave-uniform; lane l writes 16 (4) bytes at base + 16 l (4 l).
__device__ __forceinline__ unsigned lds_addr_u32(const void* p) { return (unsigned)(size_t)(GVK_LDS const char*)p; }
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, const void* lds_wave_base, int voff, int soff) {
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds_wave_base));
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(a), "v"(voff), "s"(r), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void lds_dma4(__amdgpu_buffer_rsrc_t r, const void* lds_wave_base, int voff, int soff) {
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds_wave_base));
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(a), "v"(voff), "s"(r), "s"(soff) : "memory", "m0");
}
#define GVK_DMA_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// In front of a loop that issues asm LDS-DMA: every load the compiler tracks (the prologue's fragment loads) is waited for HERE, with the
// builtin the compiler's scoreboard understands.  Otherwise it places the counted waits for those loop-invariant loads INSIDE the loop
// (s_waitcnt vmcnt(7) ... vmcnt(0) in front of their first uses), where -- blind to the DMA requests -- they drain the prefetch on
// every iteration.
#define GVK_LOADS_LANDED() __builtin_amdgcn_s_waitcnt(0x0F70)

// attention-probability dropout (vision_transformer.py:68, live for the unfrozen-backbone methods): the softmax statistics are taken
// of the undropped scores, the dropped and rescaled P feeds the P.V product; mask element (b*H + head, query, key) -- dropout.hpp
struct AttnDrop { unsigned long long seed; const unsigned long long* seed_ptr; unsigned int thresh; float inv_keep; };

// Row constants on the matrix pipe.  A score tile needs  S - c[row or column]  (running maximum / log-sum-exp) and a mask; as VALU work
// that is a v_fma and a v_cndmask per score.  Instead the constant rides an extra MFMA over an "augmented" contraction of 16 slots
// (one v_mfma_f32_32x32x16_bf16 whose operands hold, in the lanes of k = 0..7,
//      constant side:  [-c_hi, -c_mid, -c_lo, -3e38 | 0,  -e_hi, -e_mid, -e_lo, 0]      selector side:  [1, 1, 1, flag, 0, 0, 0, 0]  (first constant + mask)
//                                                                                                    or  [0, 0, 0, 0,    1, 1, 1, 0]  (second constant)
// and zeros in the lanes of k = 8..15): c split into three bf16 pieces is exact to fp32, the products with 1.0 are exact and the
// accumulator is fp32, so the tile arrives as S - c ready for v_exp_f32, with masked entries at -3e38.
__device__ __forceinline__ void split3(float r0, bf16& h0, bf16& h1, bf16& h2) {
  h0 = (bf16)r0;
  const float r1 = r0 - (float)h0;
  h1 = (bf16)r1;
  h2 = (bf16)(r1 - (float)h1);
}
// constant-side fragment: -c (and the mask magnitude -3e38 when `neg`), -e; only the lanes with hh == 0 (k = 0..7) carry it
__device__ __forceinline__ bf16x8 aug_const(float c, bool neg, float e, int hh) {
  bf16 c0, c1, c2, e0, e1, e2;
  split3(-c, c0, c1, c2);
  split3(-e, e0, e1, e2);
  const bf16 z = (bf16)0.f;
  const bf16x8 v = {c0, c1, c2, neg ? (bf16)(-3.0e38f) : z, e0, e1, e2, z};
  const bf16x8 zero = {z, z, z, z, z, z, z, z};
  return hh == 0 ? v : zero;
}
// selector-side fragments as packed words (bf16 1.0 = 0x3F80)
__device__ __forceinline__ bf16x8 aug_sel_first(bool flag, int hh) {
  const u32x4 w = {hh == 0 ? 0x3F803F80u : 0u, hh == 0 ? (flag ? 0x3F803F80u : 0x00003F80u) : 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ bf16x8 aug_sel_second(int hh) {
  const u32x4 w = {0u, 0u, hh == 0 ? 0x3F803F80u : 0u, hh == 0 ? 0x00003F80u : 0u};
  return __builtin_bit_cast(bf16x8, w);
}

__device__ __forceinline__ float half_max(float v) {      // max over the two 32-lane halves (lanes l and l + 32 hold the same row / column)
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned int)r[0]), __builtin_bit_cast(float, (unsigned int)r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned int)r[0]) + __builtin_bit_cast(float, (unsigned int)r[1]);
}

// Epilogue of a transposed 64 x 32 accumulator pair (acc[db][r]: d = db*32 + (r&3) + 8*(r>>2) + 4*hh, row = lane & 31) scaled by
// `mul[row]`: through a wave-private 4 KB LDS image (32 rows of 128 B, 16-B chunk c of row q at c ^ (q & 7)), then whole rows out --
// 8 lanes x 16 B per row, 8 rows per store instruction (per-lane 8-byte pieces at a 1.5 - 4.5 KB row stride touched 32 lines per
// instruction).  dst = address of row 0, column 0 of this wave's 32 rows; ld in elements; rows >= nrows are not stored.
__device__ __forceinline__ void store_rows_t(const f32x16 (&acc)[2], float mul, char* so, bf16* dst, size_t ld, int nrows, int lane) {
  const int r31 = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const bf16x4 o = {(bf16)(acc[db][4 * g4 + 0] * mul), (bf16)(acc[db][4 * g4 + 1] * mul), (bf16)(acc[db][4 * g4 + 2] * mul),
                        (bf16)(acc[db][4 * g4 + 3] * mul)};
      *(bf16x4*)(so + r31 * 128 + (((db * 4 + g4) ^ (r31 & 7)) << 4) + hh * 8) = o;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + (lane >> 3), ch = lane & 7;
    const u32x4 v = *(const u32x4*)(so + row * 128 + ((ch ^ (row & 7)) << 4));
    if (row < nrows) *(u32x4*)(dst + (size_t)row * ld + ch * 8) = v;
  }
}

}  // namespace gvk
