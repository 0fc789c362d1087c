// Stream-K form of the three-stage 128 x 128 bf16 GEMM tile (gemm_bf16.hip), for the shapes whose tiles do not fill the chip.
//
// M = 4132, N = 768 gives 33 x 6 = 198 tiles for 256 CUs: with one workgroup per tile 58 CUs idle while every busy CU runs a whole
// tile at the per-CU L2 -> LDS rate that bounds this loop.  Here the launch is ONE workgroup per CU and the work is the
// sequence of (tile, 64-wide k-step) units, tiles * K / 64 of them, cut into 256 equal consecutive ranges: a workgroup runs the tail
// of one tile and the head of the next (at most a few segments), every CU carries the same number of k-steps (198 * 48 / 256 = 37.1
// instead of 48 on 198 of them).
//
// A tile cut between workgroups is finished by the one that holds its FIRST k-step -- it reaches that segment LAST, at the end of its
// range, when the others' pieces (computed at the START of their ranges) have long been written: they leave their fp32 accumulators in
// a per-workgroup slot (64 KiB) behind a flag, the finisher adds the slots in workgroup order (deterministic) and runs the epilogue.
// The finisher waits on workgroups with HIGHER ids that are dispatched right behind it; a wait that is not answered within ~1 s gives
// up (the result is then wrong and the tests say so) instead of hanging the device.  Flags are cleared by their consumer, so a
// replayed launch finds them as the first one did.
#include "gemm_epilogue.hpp"
#include <map>
#include <mutex>

namespace gvk {
namespace {

constexpr int kBM = 128, kBN = 128, kBK = 64, kNS = 3;
constexpr int kSlotFloats = kBM * kBN;                // one workgroup's accumulators
constexpr int kMaxWgs = 256;

__device__ __forceinline__ long sk_first_unit(long total, int v, int nwg) { return total * v / nwg; }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_sk_kernel(GemmArgs p, float* __restrict__ part, int* __restrict__ flag) {
  constexpr int NW = 4, WM = 64, WN = 64, MT = 4, NT = 4;
  constexpr int ROWB = kBK * 2, RPI = 1024 / ROWB, CPR = ROWB / 16;
  constexpr int A_BYTES = kBM * ROWB, W_BYTES = kBN * ROWB, STAGE = A_BYTES + W_BYTES;
  constexpr int PER_TILE = kBM / (NW * RPI) + kBN / (NW * RPI);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_ok;

  const int nwg = gridDim.x, ntile = p.nbm * p.nbn, nk = p.K / kBK;
  int v;                                               // workgroups of one XCD (bid, bid + 8, ...) take consecutive ranges
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const long total = (long)ntile * nk;
  long u = sk_first_unit(total, v, nwg);
  const long u_end = sk_first_unit(total, v + 1, nwg);

  const int lane = lane_id(), wave = wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  const int tid = threadIdx.x;

  while (u < u_end) {
    const int tile = (int)(u / nk);
    const int kb = (int)(u - (long)tile * nk);
    const int ke = (int)min((long)nk, kb + (u_end - u));
    const int nt = ke - kb;
    // tile -> (row panel, column tile): the grouped rasterisation of gemm_bf16.hip (group_m row panels x all column tiles, m fastest)
    const int gsz = p.group_m * p.nbn;
    const int grp = tile / gsz, first_m = grp * p.group_m;
    const int gm = min(p.nbm - first_m, p.group_m);
    const int rem = tile - grp * gsz;
    const int tile_m = first_m + rem % gm, tile_n = rem / gm;
    const int m0 = tile_m * kBM, n0 = tile_n * kBN;
    const bf16* __restrict__ Ag = p.A + (size_t)m0 * p.lda;
    const bf16* __restrict__ Wg = p.W + (size_t)n0 * p.ldw;

    auto stage = [&](int buf, int kt) {
      char* sA = smem + buf * STAGE;
      char* sW = sA + A_BYTES;
      const int k0 = (kb + kt) * kBK;
      const int rsub = lane / CPR, slot = lane % CPR;
#pragma unroll
      for (int r = 0; r < kBM / (NW * RPI); ++r) {
        const int row = (r * NW + wave) * RPI + rsub;
        const int chunk = slot ^ swz_a128(row);
        glds16(Ag + (size_t)row * p.lda + k0 + chunk * 8, sA + (r * NW + wave) * 1024);
      }
#pragma unroll
      for (int r = 0; r < kBN / (NW * RPI); ++r) {
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

    // every wave has left the previous segment's LDS tiles, and nothing but LDS-DMA counts in vmcnt from here on
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
    __syncthreads();

#define GVK_LOAD_FRAGS(SA, SW, KS, XA, WB)                                                                  \
  {                                                                                                         \
    const int chunk_ = (KS) * 4 + lq;                                                                       \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                        \
      const int row = wm * WM + i * 16 + l15;                                                               \
      XA[i] = *(const bf16x8*)((SA) + row * ROWB + ((chunk_ ^ swz_a128(row)) << 4));                        \
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
    static_assert((kNS - 1) * PER_TILE <= 63, "vmcnt is a 6-bit counter");
#pragma unroll
    for (int sgi = 0; sgi < kNS; ++sgi)
      if (sgi < nt) stage(sgi, sgi);
    {
      const int ahead = min(nt, kNS) - 1;              // k-tiles that may still be in flight once the first one is in
      if (ahead == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * PER_TILE) & 0xF) | (((2 * PER_TILE) >> 4) << 14));
      else if (ahead == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (PER_TILE & 0xF) | ((PER_TILE >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    GVK_LOAD_FRAGS(smem, smem + A_BYTES, 0, xa0, wb0)
    __builtin_amdgcn_s_waitcnt(0xC07F);
#define GVK_TILE3(T, BUF, PREFETCH, AHEAD, LAST)                                                            \
    {                                                                                                       \
      const int buf = (BUF);                                                                                \
      const int nxt = buf == kNS - 1 ? 0 : buf + 1;                                                         \
      const char* sA = smem + buf * STAGE;                                                                  \
      const char* sW = sA + A_BYTES;                                                                        \
      GVK_LOAD_FRAGS(sA, sW, 1, xa1, wb1)                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      GVK_MMA(xa0, wb0)                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      __builtin_amdgcn_s_waitcnt(0x0070 | (((AHEAD) * PER_TILE) & 0xF) | ((((AHEAD) * PER_TILE) >> 4) << 14)); \
      __builtin_amdgcn_s_barrier();                                                                         \
      if (PREFETCH) stage(buf, (T) + kNS);                                                                  \
      if (!(LAST)) GVK_LOAD_FRAGS(smem + nxt * STAGE, smem + nxt * STAGE + A_BYTES, 0, xa0, wb0)            \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      GVK_MMA(xa1, wb1)                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                                                   \
    }
    int t = 0, b = 0;
    for (; t < nt - kNS; ++t) {
      GVK_TILE3(t, b, true, kNS - 2, false)
      b = b == kNS - 1 ? 0 : b + 1;
    }
    for (; t < nt; ++t) {                                // the last (up to) kNS k-tiles request nothing
      const int r = nt - 1 - t;
      if (r == 2) GVK_TILE3(t, b, false, 1, false)
      else if (r == 1) GVK_TILE3(t, b, false, 0, false)
      else GVK_TILE3(t, b, false, 0, true)
      b = b == kNS - 1 ? 0 : b + 1;
    }
#undef GVK_TILE3
#undef GVK_LOAD_FRAGS
#undef GVK_MMA

    if (kb > 0) {
      // a later piece of a tile another workgroup finishes: leave the accumulators in this workgroup's slot, then raise the flag
      float* dst = part + (size_t)v * kSlotFloats;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) *(f32x4*)(dst + ((i * NT + j) * 256 + tid) * 4) = acc[i][j];
      // The slots and flags live in UNCACHED device memory (coherent across the eight XCDs' private L2s without write-backs or invalidates:
      // an agent-scope fence here is an L2 write-back per wave, +70 us per launch when it was tried): a store is visible once it is
      // acknowledged, so "all stores of the workgroup retired, then the flag" is vmcnt(0) + the barrier
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flag + v, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (ke < nk) {
        // the head of a tile whose later k-steps belong to the next workgroup(s): add their pieces, in workgroup order
        for (int vv = v + 1; vv < nwg && sk_first_unit(total, vv, nwg) < (long)(tile + 1) * nk; ++vv) {
          if (tid == 0) {
            int ok = 0;
            for (int it = 0; it < (1 << 21); ++it) {     // ~1 s at most; in practice the flag was raised long ago
              if (__hip_atomic_load(flag + vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) { ok = 1; break; }
              __builtin_amdgcn_s_sleep(32);
            }
            s_ok = ok;
          }
          __syncthreads();                               // (orders the slot reads below behind the flag read; the slot is uncached)
          const float* src = part + (size_t)vv * kSlotFloats;
          if (s_ok) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j) {
                const f32x4 o = *(const f32x4*)(src + ((i * NT + j) * 256 + tid) * 4);
                acc[i][j] += o;
              }
          }
          __syncthreads();
          if (tid == 0) __hip_atomic_store(flag + vv, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed: ready for the next launch
        }
      }
      gemm_epilogue<EPI, false, MT, NT>(p, acc, m0 + wm * WM, n0 + wn * WN, l15, lq);
    }
    u += nt;
  }
}

struct SkWorkspace { float* part = nullptr; int* flag = nullptr; };
std::mutex g_sk_mu;
std::map<hipStream_t, SkWorkspace> g_sk_ws;

static int sk_workspace(hipStream_t stream, SkWorkspace& out) {
  std::lock_guard<std::mutex> lk(g_sk_mu);
  auto it = g_sk_ws.find(stream);
  if (it == g_sk_ws.end()) {
    SkWorkspace w;
    hipError_t e = hipExtMallocWithFlags((void**)&w.part, (size_t)kMaxWgs * kSlotFloats * sizeof(float), hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&w.flag, kMaxWgs * sizeof(int), hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipMemset(w.flag, 0, kMaxWgs * sizeof(int));
    if (e != hipSuccess) return set_error(-3, "gemm stream-K workspace: %s", hipGetErrorString(e));
    it = g_sk_ws.emplace(stream, w).first;
  }
  out = it->second;
  return 0;
}

template <int EPI>
static int launch_sk(const GemmArgs& a, hipStream_t stream) {
  constexpr int lds = kNS * (kBM + kBN) * kBK * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sk_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(-3, "hipFuncSetAttribute(gemm_sk): %s", hipGetErrorString(e));
    attr_set = true;
  }
  SkWorkspace w;
  if (int rc = sk_workspace(stream, w)) return rc;
  GemmArgs p = a;
  p.nbm = (a.M + kBM - 1) / kBM;
  p.nbn = a.N / kBN;
  p.a_rows = (a.M + 127) / 128 * 128;
  {
    const int per_xcd = (p.nbm * p.nbn) >> 3;
    int g = (per_xcd + p.nbn / 2) / p.nbn;
    p.group_m = g < 1 ? 1 : g > 8 ? 8 : g;
  }
  GVK_LAUNCH((gemm_sk_kernel<EPI>), dim3(kMaxWgs), dim3(256), lds, stream, p, w.part, w.flag);
  return check_launch("gemm_nt_bf16 (stream-K)");
}

}  // namespace

bool gemm_sk_supports(const GemmArgs& a, int epilogue) {
  if (!(epilogue == GVK_EPI_STORE_BF16 || epilogue == GVK_EPI_BIAS_RES_F32 || epilogue == GVK_EPI_BIAS_RES_F32_BF16 || epilogue == GVK_EPI_STORE_F32))
    return false;
  if (a.N % kBN != 0 || a.K % kBK != 0 || a.drop_thresh != 0u) return false;
  const long tiles = (long)((a.M + kBM - 1) / kBM) * (a.N / kBN);
  const long units = tiles * (a.K / kBK);
  // fewer tiles than CUs (else the plain launch already has a tile per CU and more), at least two k-steps per workgroup, and a tile cut
  // into at most four pieces (tiles >= 64)
  return tiles < kMaxWgs && tiles >= 64 && units >= 2 * kMaxWgs;
}

int launch_gemm_sk(const GemmArgs& a, int epilogue, hipStream_t stream) {
  if (!gemm_sk_supports(a, epilogue)) return set_error(-2, "gvk_gemm_nt_bf16: the stream-K tile does not cover this shape / epilogue");
  switch (epilogue) {
    case GVK_EPI_STORE_BF16: return launch_sk<GVK_EPI_STORE_BF16>(a, stream);
    case GVK_EPI_BIAS_RES_F32: return launch_sk<GVK_EPI_BIAS_RES_F32>(a, stream);
    case GVK_EPI_BIAS_RES_F32_BF16: return launch_sk<GVK_EPI_BIAS_RES_F32_BF16>(a, stream);
    default: return launch_sk<GVK_EPI_STORE_F32>(a, stream);
  }
}

}  // namespace gvk
