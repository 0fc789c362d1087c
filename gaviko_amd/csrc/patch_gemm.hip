// Patch embedding as an IMPLICIT GEMM (Conv3d with kernel = stride = (pd, ph, pw) over a (1, D, H, W) volume,
// vision_transformer.py:126-128,150-157): tokens[b*n + t][c] = sum_k patch(b, t)[k] * W[c][k] + bias[c] + pos[t][c], written straight into
// the token rows of the residual stream (and GAViKO's local stream), with NO im2col matrix in between.
//
// The im2col + GEMM pair it replaces wrote 24.6 MB of bf16 patch rows per 4 volumes and read them back (19 us + 35 us).  Here the A operand
// is gathered from the fp32 volume inside the GEMM: the K axis in Conv3d.weight.flatten(1) order is (kd, kh, kw), so one 64-wide k-tile of
// one token is FOUR runs of 16 contiguous voxels (kh .. kh+3 of one depth slice) -- four lanes per run, each thread fetching its quarter of
// one run of eight token rows (8 x global_load_dwordx4), converting to bf16 and writing half-chunks into the same swizzled LDS image the other GEMM kernels use
// (LDS-DMA cannot convert, so this operand goes through registers; the weight tile still arrives by LDS-DMA).  Three LDS stages, gathers
// and weight tiles requested two k-tiles ahead behind counted vmcnt waits.  128 x 128 x 64 tiles, four waves, v_mfma_f32_16x16x32_bf16, the shared PATCH
// epilogue (bias + position rows + scatter to token rows row_off.. of every sample, optional second copy).
#include "gemm_epilogue.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

struct PatchGeom {
  const float* img;                    // [B][1][D][H][W]
  int D, H, W, pd, ph, pw, gh, gw, ntok;   // grid (D/pd, gh = H/ph, gw = W/pw), ntok = tokens per volume
};

__global__ __launch_bounds__(256) void patch_gemm_kernel(GemmArgs p, PatchGeom g) {
  constexpr int BM = 128, BN = 128, BK = 64, ROWB = BK * 2, A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
  constexpr int MT = 4, NT = 4, WM = 64, WN = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware bijective remap + grouped rasterisation (as gemm_nt_kernel): the six column tiles of a row panel run on one XCD, close in
  // time, so the volume is fetched from HBM once and the other five gathers hit that XCD's L2
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

  const int lane = lane_id(), wave = wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  // ---- A gather.  Four lanes per 64-byte run (quarter q = tid & 3), the four runs of a token row on 16 consecutive lanes: a wave's load
  // instruction touches 16 cache lines (with two lanes per token row it touched 64 and the address path, not HBM, set the pace).  A thread
  // serves the SAME (run, quarter) of eight token rows 16 apart; rows past M re-read the last token (never stored).
  const int quarter = threadIdx.x & 3, run = (threadIdx.x >> 2) & 3, arow0 = threadIdx.x >> 4;       // rows arow0 + 16 i
  unsigned int aoff[8];                                  // element offsets into the volume (49 MB per 4 volumes: 32 bits are plenty)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = min(m0 + arow0 + 16 * i, p.M - 1);
    const int b = m / g.ntok, idx = m - b * g.ntok;
    const int td = idx / (g.gh * g.gw), r2 = idx - td * (g.gh * g.gw), th = r2 / g.gw, tw = r2 - th * g.gw;
    aoff[i] = (unsigned int)((((size_t)b * g.D + (size_t)td * g.pd) * g.H + (size_t)th * g.ph + run) * g.W + (size_t)tw * g.pw + 4 * quarter);
  }
  const int slice = g.ph * g.pw;                       // k values per depth slice of a patch (a multiple of 64: a k-tile never straddles two)
  // k-tiles are gathered in order, so the offsets just walk: +4 image rows per tile, and a jump to the next depth slice after every
  // slice / 64 tiles.  Two register sets: the voxels of tile t+2 are requested while tile t is multiplied and tile t+1 waits to be converted.
  f32x4 avA[8], avB[8];
  const int tps = slice / BK;
  const unsigned int step_tile = (unsigned int)((BK / g.pw) * g.W), step_slice = (unsigned int)((g.H - g.ph) * g.W);
  unsigned int walk = 0u;
  int in_slice = 0;
  auto gather = [&](int, f32x4 (&av)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) av[i] = *(const f32x4*)(g.img + (size_t)(aoff[i] + walk));
    walk += step_tile;
    if (++in_slice == tps) { in_slice = 0; walk += step_slice; }
  };
  auto commit = [&](int buf, const f32x4 (&av)[8]) {    // four k-values per row: half a swizzled 16-byte chunk
    char* sA = smem + buf * STAGE + arow0 * ROWB + (((run * 2 + (quarter >> 1)) ^ swz_a128(arow0)) << 4) + (quarter & 1) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bf16x4 h4 = {(bf16)av[i][0], (bf16)av[i][1], (bf16)av[i][2], (bf16)av[i][3]};
      *(bf16x4*)(sA + i * 16 * ROWB) = h4;               // swz_a128(row + 16) == swz_a128(row)
    }
  };
  // ---- W tile by LDS-DMA (gemm_nt_kernel's mapping: 8 rows per 1-KiB wave instruction, source-side XOR swizzle)
  const bf16* __restrict__ Wg = p.W + (size_t)n0 * p.ldw;
  auto stage_w = [&](int buf, int kt) {
    char* sW = smem + buf * STAGE + A_BYTES;
    const int rsub = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (r * 4 + wave) * 8 + rsub;
      glds16(Wg + (size_t)row * p.ldw + kt * BK + ((slot ^ swz_w(row)) << 3), sW + (r * 4 + wave) * 1024);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt = p.K / BK;
  // THREE LDS stages: the weight tile of k-tile t+2 is requested (LDS-DMA) and the voxels of t+2 are gathered while tile t is multiplied --
  // with one workgroup per CU nothing else hides the L2 round trip of a tile requested only one tile ahead (48 exposed round trips: 74 us).
  // A wave's vector-memory operations retire in order, so "tile t+1 has landed" is a COUNTED wait: the 8 gather loads and 4 LDS-DMA
  // instructions of tile t+2 may stay in flight (vmcnt(12)).
  gather(0, avA);
  stage_w(0, 0);
  if (nt > 1) { gather(1, avB); stage_w(1, 1); }
  commit(0, avA);
  if (nt > 1) __builtin_amdgcn_s_waitcnt(0x0070 | 12); else __builtin_amdgcn_s_waitcnt(0x0070);     // lgkmcnt(0) + vmcnt(12 | 0)
  __builtin_amdgcn_s_barrier();
  // one k-tile: `nxt` holds tile t+1 (requested one tile ago), `fill` is free for tile t+2
  auto ktile = [&](const int t, const int buf, f32x4 (&nxt)[8], f32x4 (&fill)[8]) {
    const int b1 = buf == 2 ? 0 : buf + 1, b2 = b1 == 2 ? 0 : b1 + 1;
    if (t + 2 < nt) {                                    // (buffer b2 was last read one barrier ago)
      gather(t + 2, fill);
      stage_w(b2, t + 2);
    }
    const char* sA = smem + buf * STAGE;
    const char* sW = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xa[MT], wb[NT];
      const int chunk = ks * 4 + lq;
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
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nt) commit(b1, nxt);
    if (t + 2 < nt) __builtin_amdgcn_s_waitcnt(0x0070 | 12); else __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
  };
  int buf = 0;
  for (int t = 0; t < nt; t += 2) {
    ktile(t, buf, avB, avA);
    buf = buf == 2 ? 0 : buf + 1;
    if (t + 1 < nt) {
      ktile(t + 1, buf, avA, avB);
      buf = buf == 2 ? 0 : buf + 1;
    }
  }
  gemm_epilogue<GVK_EPI_PATCH_F32, false, MT, NT>(p, acc, m0 + wm * WM, n0 + wn * WN, l15, lq);
}

}  // namespace gvk

extern "C" int gvk_patch_embed_bf16(const float* img, const void* w, const float* bias, const float* pos, float* out0, float* out1, int B, int D,
                                    int H, int W, int pd, int ph, int pw, int C, int rows_out, int row_off, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(img && w && pos && out0 && B > 0, "gvk_patch_embed_bf16: null pointer");
  GVK_REQUIRE(D % pd == 0 && H % ph == 0 && W % pw == 0, "gvk_patch_embed_bf16: volume %dx%dx%d not divisible by patch %dx%dx%d", D, H, W, pd, ph, pw);
  GVK_REQUIRE(pw == 16 && (ph * pw) % 64 == 0 && W % 4 == 0, "gvk_patch_embed_bf16: built for 16-voxel runs (pw = 16) and ph * pw a multiple of 64");
  GVK_REQUIRE(C % 128 == 0, "gvk_patch_embed_bf16: C=%d must be a multiple of 128 (use gvk_patchify_bf16 + gvk_gemm_nt_bf16)", C);
  const int ntok = (D / pd) * (H / ph) * (W / pw), K = pd * ph * pw;
  GVK_REQUIRE(rows_out >= ntok + row_off && row_off >= 0, "gvk_patch_embed_bf16: rows_out / row_off inconsistent");
  GemmArgs a{};
  a.W = (const bf16*)w; a.out0 = out0; a.out1 = out1; a.bias = bias; a.pos = pos;
  a.M = B * ntok; a.N = C; a.K = K; a.ldw = K; a.ldo = C; a.rows_in = ntok; a.rows_out = rows_out; a.row_off = row_off;
  a.nbm = (a.M + 127) / 128; a.nbn = C / 128; a.inv_keep = 1.f;
  PatchGeom g{img, D, H, W, pd, ph, pw, H / ph, W / pw, ntok};
  constexpr int lds = 3 * (128 + 128) * 64 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(patch_gemm): %s", hipGetErrorString(e));
    attr_set = true;
  }
  GVK_LAUNCH(patch_gemm_kernel, dim3(a.nbm * a.nbn), dim3(256), lds, (hipStream_t)stream, a, g);
  return check_launch("patch_embed_bf16");
}
