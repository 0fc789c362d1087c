// Every parameter gradient of one rank-L side-path module of one layer in ONE launch (round 4).
//
// Replaces, per layer and stream, two gvk_outer_reduce calls (partial + final each), one gvk_reduce_batch (partial + final) and, on the
// MWSA stream, gvk_ln_lowrank_affine: 12 launches per layer -> 2.  What the reference computes here is the autograd of
//   gaviko.py:155-156,187  (GPA: proj_down shared by the global and the local tokens, proj_up over every row)
//   gaviko.py:231-232,242  (MWSA: LayerNorm + proj_down, the qkv matrix, proj_up behind proj_drop)
// i.e. outer products  D[l][c] = sum_m narrow'[m][l] * wide'[m][c]  over M = 4..8 k token rows, plus a handful of tiny column sums /
// (J x L) products over the same rows.
//
// Structure.  The grid is a job list: [outer job 0 | outer job 1 | small jobs].  An outer-product workgroup (4 waves) owns one 64-column
// tile and FOUR row slabs (one per wave, <= 64 rows each: the whole slab's wide rows are requested before the first MFMA -- one HBM round
// trip per wave), multiplies on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products), sums its four waves' tiles through
// LDS and writes ONE partial tile [(L+1)][64] (row L = column sum of wide').  The partial tiles of a column tile are then summed by the
// workgroup that arrives LAST at the tile's ticket counter -- in slab order, so the result does not depend on who that is: deterministic,
// no second launch, no 5-us dependent hop.  The hand-off between workgroups is the agent-scope release / ticket / acquire form
// (cdna_hip_programming.md Guideline 16): plain partial stores, every wave drains its stores, workgroup barrier, ONE lane releases at
// agent scope and takes the ticket; the last arriver acquires at agent scope before any wave of it loads a partial.  The ticket words are
// zero at allocation and the last arriver writes its word back to zero, so a replayed plan needs no memset (launches that share ticket
// words are ordered by their stream).
#include "common.hpp"
#include "dropout.hpp"
#include "../../include/gaviko_hip.h"
#include <algorithm>

namespace gvk {

constexpr int kPgRows = 64;          // rows per wave (one slab)
constexpr int kPgMaxOuter = 2;
constexpr int kPgMaxSmall = 8;
constexpr int kPgMaxSg = 40;         // slab groups per outer job (M <= 40 * 256 rows)
constexpr int kPgRedSlabs = 32;      // row slabs of a small job

struct PgOuter {
  const float* narrow; const float* wide; const float* narrow2; const float* wide2;   // rows M1.. come from (narrow2, wide2)
  const float* lat_override;                                                        // narrow rows t < P of a sample come from here [B*P][L]
  const float* mean; const float* rstd;                                             // wide' = (wide - mean[m]) * rstd[m]
  float* out; float* colsum;
  const float* aff_w; const float* aff_g; const float* aff_b; float* aff_dgamma; float* aff_dbeta; float* aff_dbias;
  unsigned long long seed; unsigned int drop_thresh; float inv_keep;                // wide' *= dropout mask of element (m, c)
  int M, M1, T, P, transposed, accumulate;
  int wg0, nsg, tick0; long scr0;                                                    // first workgroup / slab groups / first ticket / scratch offset (floats)
};
struct PgSmall {
  const float* a; const float* b; const float* a2; float* out;                       // as gvk_reduce_job
  int M, M1, J, L, accumulate, wg0, nslab, tick0; long scr0;
};
struct PgArgs {
  PgOuter o[kPgMaxOuter];
  PgSmall s[kPgMaxSmall];
  float* scratch; int* tickets; const unsigned long long* seed_ptr;
  int nouter, nsmall, small_wg0, C, L;
};

// One lane publishes the workgroup's partial stores and takes the ticket; returns (to every thread) whether this workgroup arrived last.
__device__ __forceinline__ bool pg_arrive_last(int* ticket, int n, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its partial stores ...
  __syncthreads();                                           // ... before the one lane that signals for all of them
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // write the XCD L2's dirty lines back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (kept explicit: the compiler may drop the fence's own wait)
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == n - 1) ? 1 : 0;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // drop this CU's stale lines before any wave of it reads a partial
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this stream
    }
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}

template <int NT>
__global__ __launch_bounds__(256) void param_grads_kernel(PgArgs p) {
  constexpr int NW = NT * 16;                                // padded narrow width (L + 1 <= NW)
  __shared__ float lds[4 * kPgRows * 32];                    // narrow rows of the four slabs, later the four waves' accumulator tiles (32 KiB)
  __shared__ float st[4][kPgRows][2];
  __shared__ float sw[4][32];
  __shared__ int s_flag;
  const int C = p.C, L = p.L;
  const int lane = lane_id(), wave = wave_id();
  const int bid = blockIdx.x;

  if (bid >= p.small_wg0) {
    // ------------------------------------------------------------------ small jobs: column sums / (J x L) products over the rows
    int ji = 0;
#pragma unroll
    for (int k = 1; k < kPgMaxSmall; ++k)
      if (k < p.nsmall && bid >= p.s[k].wg0) ji = k;
    const PgSmall jb = p.s[ji];
    const int local = bid - jb.wg0;
    const int slab = local % jb.nslab, chunk = local / jb.nslab;
    const int nout = jb.b ? jb.J * jb.L : jb.J;
    const int o0 = chunk * 64;
    const int no = min(64, nout - o0);
    const int rows_per = (jb.M + jb.nslab - 1) / jb.nslab;
    const int r0 = slab * rows_per, r1 = min(jb.M, r0 + rows_per);
    const int oi = threadIdx.x & 63, sl = threadIdx.x >> 6;     // 4 row slices
    float acc = 0.f;
    if (oi < no) {
      const int o = o0 + oi;
      if (jb.b != nullptr) {
        const int j = o / jb.L, l = o - j * jb.L;
        for (int m = r0 + sl; m < r1; m += 32) {                // eight rows' loads in flight per thread and pass
          float av[8], bv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int mm = m + u * 4;
            av[u] = mm < r1 ? jb.a[(size_t)mm * jb.J + j] : 0.f;
            bv[u] = mm < r1 ? jb.b[(size_t)mm * jb.L + l] : 0.f;
          }
          acc += ((av[0] * bv[0] + av[1] * bv[1]) + (av[2] * bv[2] + av[3] * bv[3])) + ((av[4] * bv[4] + av[5] * bv[5]) + (av[6] * bv[6] + av[7] * bv[7]));
        }
      } else {
        for (int m = r0 + sl; m < r1; m += 32) {
          float av[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int mm = m + u * 4;
            av[u] = mm < r1 ? (mm < jb.M1 ? jb.a[(size_t)mm * jb.J + o] : jb.a2[(size_t)(mm - jb.M1) * jb.J + o]) : 0.f;
          }
          acc += ((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7]));
        }
      }
    }
    lds[threadIdx.x] = acc;
    __syncthreads();
    float* part = p.scratch + jb.scr0 + (size_t)chunk * jb.nslab * 64;
    if (sl == 0 && oi < no) part[slab * 64 + oi] = (lds[oi] + lds[64 + oi]) + (lds[128 + oi] + lds[192 + oi]);
    if (!pg_arrive_last(p.tickets + jb.tick0 + chunk, jb.nslab, &s_flag)) return;
    if (threadIdx.x < no) {
      float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < kPgRedSlabs; s += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a4[u] += (s + u < jb.nslab) ? part[(s + u) * 64 + threadIdx.x] : 0.f;
      }
      const float t = (a4[0] + a4[1]) + (a4[2] + a4[3]);
      float* dst = jb.out + o0 + threadIdx.x;
      *dst = jb.accumulate ? *dst + t : t;
    }
    return;
  }

  // -------------------------------------------------------------------- outer products
  const int k = (p.nouter > 1 && bid >= p.o[1].wg0) ? 1 : 0;
  const PgOuter& J = p.o[k];
  const int nct = C >> 6;
  const int local = bid - J.wg0;
  const int ct = local % nct, sg = local / nct;
  const int rows_per = (J.M + 4 * J.nsg - 1) / (4 * J.nsg);      // <= kPgRows by the host's choice of nsg
  const int r0 = min(J.M, (4 * sg + wave) * rows_per);
  const int nr = min(J.M, r0 + rows_per) - r0;
  const int j = lane & 15, kq = lane >> 4;
  const int c = ct * 64 + 4 * j;
  unsigned long long seed = J.seed;
  if (J.drop_thresh != 0u && p.seed_ptr != nullptr) seed += *p.seed_ptr;
  float (*nar)[NW] = (float (*)[NW])(lds + wave * kPgRows * 32);

  // the whole slab's wide rows: 16 float4 per lane requested before anything else waits
  auto wide_row = [&](int gm) -> const float* {
    return (J.narrow2 != nullptr && gm >= J.M1) ? J.wide2 + (size_t)(gm - J.M1) * C : J.wide + (size_t)gm * C;
  };
  f32x4 xr[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = 16 * i + 4 * u + kq;
      xr[i][u] = (r < nr) ? *(const f32x4*)(wide_row(r0 + r) + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  // this wave's narrow rows -> LDS (column L = 1: row L of the result is the column sum of wide'), four elements per lane and pass
  for (int i0 = lane; i0 < nr * NW; i0 += 4 * 64) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 64;
      const int r = i / NW, l = i - r * NW, m = r0 + r;
      v[u] = (l == L) ? 1.f : 0.f;
      if (i < nr * NW && l < L) {
        const float* src = (J.narrow2 != nullptr && m >= J.M1) ? J.narrow2 + (size_t)(m - J.M1) * L : J.narrow + (size_t)m * L;
        if (J.lat_override != nullptr) {
          const int s = m / J.T, t = m - s * J.T;
          if (t < J.P) src = J.lat_override + ((size_t)s * J.P + t) * L;
        }
        v[u] = src[l];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 64;
      if (i < nr * NW) nar[i / NW][i % NW] = v[u];
    }
  }
  if (J.mean != nullptr && lane < nr) {
    st[wave][lane][0] = J.mean[r0 + lane];
    st[wave][lane][1] = J.rstd[r0 + lane];
  }
  __syncthreads();
  if (J.aff_w != nullptr && lane < NW) {                       // S[l] = sum_m narrow[m][l] over this wave's rows (the affine epilogue needs it)
    float s = 0.f;
    for (int r = 0; r < nr; ++r) s += nar[r][lane];
    sw[wave][lane] = s;
  }
  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (16 * i >= nr) break;                                   // wave-uniform
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = 16 * i + 4 * u + kq;
      f32x4 xv = xr[i][u];
      if (r < nr) {
        if (J.drop_thresh != 0u) {
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[e] *= drop_scale(seed, (unsigned long long)(r0 + r) * C + c + e, J.drop_thresh, J.inv_keep);
        }
        if (J.mean != nullptr) {
          const float mu = st[wave][r][0], rs = st[wave][r][1];
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[e] = (xv[e] - mu) * rs;
        }
      }
      if (16 * i + 4 * u < nr) {                               // wave-uniform: skip k-steps wholly past the slab
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float a = (r < nr) ? nar[r][t * 16 + j] : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[e], acc[t][e], 0, 0, 0);
        }
      }
    }
  }
  // the four waves' tiles -> one: every wave leaves its 16 NT accumulator registers in LDS, wave w then sums its share in wave order
  __syncthreads();                                             // (every wave is done reading its narrow rows)
  float (*red)[16 * NT][64] = (float (*)[16 * NT][64])lds;     // [wave][register][lane], 16 KiB per tile row NT = 1, 32 KiB NT = 2
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[wave][(t * 4 + e) * 4 + q][lane] = acc[t][e][q];
  __syncthreads();
  float* slab = p.scratch + J.scr0 + ((size_t)ct * J.nsg + sg) * ((L + 2) * 64);
  // wave w finishes tile row t = w / (4 / NT) ... : the (t, q) pairs are dealt to the waves; a lane then holds the four e of one (l, 4 columns)
  for (int tq = wave; tq < NT * 4; tq += 4) {
    const int t = tq >> 2, q = tq & 3;
    const int l = t * 16 + 4 * kq + q;                         // D row = 4 * (lane >> 4) + register
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rg = (t * 4 + e) * 4 + q;
      o[e] = (red[0][rg][lane] + red[1][rg][lane]) + (red[2][rg][lane] + red[3][rg][lane]);
    }
    if (l <= L) *(f32x4*)(slab + l * 64 + 4 * j) = o;
  }
  if (J.aff_w != nullptr && wave == 0 && lane < NW)
    slab[(L + 1) * 64 + lane] = (sw[0][lane] + sw[1][lane]) + (sw[2][lane] + sw[3][lane]);

  if (!pg_arrive_last(p.tickets + J.tick0 + ct, J.nsg, &s_flag)) return;

  // ---- last arriver of this column tile: sum the slab groups' tiles in slab order, then the epilogue
  const float* tile0 = p.scratch + J.scr0 + (size_t)ct * J.nsg * ((L + 2) * 64);
  const int stride = (L + 2) * 64;
  float (*qt)[64] = (float (*)[64])lds;                       // affine epilogue: the summed tile [L + 1][64] and S behind it
  const int ngran = (L + 1) * 16 + ((J.aff_w != nullptr) ? NW / 4 : 0);     // float4 granules: the tile (+ the S row)
  for (int g = threadIdx.x; g < ngran; g += 256) {
    f32x4 a4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int s = 0; s < kPgMaxSg; s += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s + u < J.nsg) a4[u] += *(const f32x4*)(tile0 + (size_t)(s + u) * stride + 4 * g);
    }
    const f32x4 v = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    const int l = g >> 4, cc = ct * 64 + 4 * (g & 15);
    if (J.aff_w != nullptr) {
      *(f32x4*)(&qt[0][0] + 4 * g) = v;                        // rows 0..L-1 = Q, row L = column sum (unused), row L+1.. = S
      continue;
    }
    if (l < L) {
      if (J.out != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float* o = J.transposed ? J.out + (size_t)(cc + e) * L + l : J.out + (size_t)l * C + cc + e;
          *o = J.accumulate ? *o + v[e] : v[e];
        }
      }
    } else if (J.colsum != nullptr) {
      f32x4* o = (f32x4*)(J.colsum + cc);
      *o = J.accumulate ? *o + v : v;
    }
  }
  if (J.aff_w == nullptr) return;
  // Affine / weight gradients of  y = LN(x) . Wd^T  from Q[l][c] = sum_m dlat[m][l] xhat[m][c] and S[l] = sum_m dlat[m][l]:
  //   dWd[l][c] = g_c Q[l][c] + b_c S[l],  dgamma_c = sum_l Wd[l][c] Q[l][c],  dbeta_c = sum_l Wd[l][c] S[l],  dbias_l = S[l]
  __syncthreads();
  const float* S = &qt[L + 1][0];
  if (threadIdx.x < 64) {
    const int cc = ct * 64 + threadIdx.x;
    const float gc = J.aff_g[cc], bc = J.aff_b[cc];
    float dg = 0.f, db = 0.f;
    for (int l = 0; l < L; ++l) {
      const float q = qt[l][threadIdx.x], sl = S[l], w = J.aff_w[(size_t)l * C + cc];
      float* o = J.out + (size_t)l * C + cc;
      const float d = gc * q + bc * sl;
      *o = J.accumulate ? *o + d : d;
      dg += w * q;
      db += w * sl;
    }
    J.aff_dgamma[cc] = J.accumulate ? J.aff_dgamma[cc] + dg : dg;
    J.aff_dbeta[cc] = J.accumulate ? J.aff_dbeta[cc] + db : db;
  }
  if (ct == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + L && J.aff_dbias != nullptr) {
    const int l = threadIdx.x - 64;
    J.aff_dbias[l] = J.accumulate ? J.aff_dbias[l] + S[l] : S[l];
  }
}

static unsigned int pg_drop_threshold(float p) { return drop_threshold_u32(p); }

}  // namespace gvk

extern "C" int64_t gvk_param_grads_scratch(const gvk_pgrad_outer* outer, int n_outer, const gvk_reduce_job* small, int n_small, int C, int L) {
  using namespace gvk;
  int64_t n = 0;
  for (int k = 0; k < n_outer; ++k) {
    const int M = outer[k].M + (outer[k].narrow2 ? outer[k].M2 : 0);
    const int nsg = std::max(1, (M + 4 * kPgRows - 1) / (4 * kPgRows));
    n += (int64_t)(C / 64) * nsg * (L + 2) * 64;
  }
  for (int k = 0; k < n_small; ++k) {
    const int nout = small[k].b ? small[k].J * small[k].L : small[k].J;
    n += (int64_t)((nout + 63) / 64) * kPgRedSlabs * 64;
  }
  return n;
}

extern "C" int gvk_param_grads(const gvk_pgrad_outer* outer, int n_outer, const gvk_reduce_job* small, int n_small, float* scratch,
                               int64_t scratch_elems, int32_t* tickets, int n_tickets, const void* seed_ptr, int C, int L, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(scratch && tickets && n_outer >= 0 && n_outer <= kPgMaxOuter && n_small >= 0 && n_small <= kPgMaxSmall && n_outer + n_small > 0,
              "gvk_param_grads: up to %d outer and %d small jobs, scratch and tickets required", kPgMaxOuter, kPgMaxSmall);
  GVK_REQUIRE(C > 0 && C % 64 == 0 && L > 0 && L <= 31, "gvk_param_grads: C=%d must be a multiple of 64 and L=%d at most 31", C, L);
  GVK_REQUIRE(n_outer == 0 || outer != nullptr, "gvk_param_grads: null outer job list");
  GVK_REQUIRE(n_small == 0 || small != nullptr, "gvk_param_grads: null small job list");
  PgArgs a{};
  a.scratch = scratch; a.tickets = tickets; a.seed_ptr = (const unsigned long long*)seed_ptr; a.C = C; a.L = L;
  a.nouter = n_outer; a.nsmall = n_small;
  int wg = 0, tick = 0;
  long scr = 0;
  const int nct = C / 64;
  for (int k = 0; k < n_outer; ++k) {
    const gvk_pgrad_outer& d = outer[k];
    GVK_REQUIRE(d.narrow && d.wide && d.M > 0 && (d.out || d.colsum), "gvk_param_grads: outer job %d: null pointer / empty", k);
    GVK_REQUIRE((d.mean == nullptr) == (d.rstd == nullptr), "gvk_param_grads: outer job %d: mean / rstd must come together", k);
    GVK_REQUIRE(d.narrow2 == nullptr || (d.wide2 != nullptr && d.M2 > 0 && d.mean == nullptr && d.lat_override == nullptr && d.drop_p <= 0.f),
                "gvk_param_grads: outer job %d: the second source takes plain rows only", k);
    GVK_REQUIRE(d.lat_override == nullptr || (d.T > 0 && d.P > 0 && d.P <= d.T), "gvk_param_grads: outer job %d: override needs 0 < P <= T", k);
    GVK_REQUIRE(d.drop_p >= 0.f && d.drop_p < 1.f && (d.drop_p == 0.f || seed_ptr != nullptr), "gvk_param_grads: outer job %d: drop_p in [0,1) and a seed word", k);
    GVK_REQUIRE(d.aff_w == nullptr || (d.aff_gamma && d.aff_beta && d.aff_dgamma && d.aff_dbeta && d.out && !d.transposed && d.colsum == nullptr),
                "gvk_param_grads: outer job %d: the affine epilogue needs gamma / beta / dgamma / dbeta, out [L][C] and takes no colsum", k);
    PgOuter& o = a.o[k];
    o.narrow = d.narrow; o.wide = d.wide; o.narrow2 = d.narrow2; o.wide2 = d.wide2; o.lat_override = d.lat_override;
    o.mean = d.mean; o.rstd = d.rstd; o.out = d.out; o.colsum = d.colsum;
    o.aff_w = d.aff_w; o.aff_g = d.aff_gamma; o.aff_b = d.aff_beta; o.aff_dgamma = d.aff_dgamma; o.aff_dbeta = d.aff_dbeta; o.aff_dbias = d.aff_dbias;
    o.seed = d.seed; o.drop_thresh = pg_drop_threshold(d.drop_p); o.inv_keep = d.drop_p > 0.f ? 1.f / (1.f - d.drop_p) : 1.f;
    o.M1 = d.M; o.M = d.M + (d.narrow2 ? d.M2 : 0); o.T = d.T; o.P = d.P; o.transposed = d.transposed; o.accumulate = d.accumulate;
    o.nsg = std::max(1, (o.M + 4 * kPgRows - 1) / (4 * kPgRows));
    GVK_REQUIRE(o.nsg <= kPgMaxSg, "gvk_param_grads: outer job %d: %d rows exceed %d", k, o.M, kPgMaxSg * 4 * kPgRows);
    o.wg0 = wg; o.tick0 = tick; o.scr0 = scr;
    wg += nct * o.nsg; tick += nct; scr += (long)nct * o.nsg * (L + 2) * 64;
  }
  a.small_wg0 = wg;
  for (int k = 0; k < n_small; ++k) {
    const gvk_reduce_job& j = small[k];
    GVK_REQUIRE(j.a && j.out && j.M > 0 && j.J > 0 && (j.b == nullptr || j.L > 0), "gvk_param_grads: small job %d malformed", k);
    GVK_REQUIRE(j.a2 == nullptr || (j.b == nullptr && j.M2 > 0), "gvk_param_grads: small job %d: a second source (a2, M2) goes with a column sum only", k);
    const int nout = j.b ? j.J * j.L : j.J;
    const int rows = j.M + (j.a2 ? j.M2 : 0);
    PgSmall& s = a.s[k];
    s.a = j.a; s.b = j.b; s.a2 = j.a2; s.out = j.out; s.M = rows; s.M1 = j.a2 ? j.M : 0x7fffffff; s.J = j.J; s.L = j.L; s.accumulate = j.accumulate;
    s.nslab = std::min(kPgRedSlabs, std::max(1, (rows + 127) / 128));
    const int nchunk = (nout + 63) / 64;
    s.wg0 = wg; s.tick0 = tick; s.scr0 = scr;
    wg += nchunk * s.nslab; tick += nchunk; scr += (long)nchunk * kPgRedSlabs * 64;
  }
  GVK_REQUIRE(scr <= scratch_elems, "gvk_param_grads: scratch holds %lld floats, %ld needed", (long long)scratch_elems, scr);
  GVK_REQUIRE(tick <= n_tickets, "gvk_param_grads: %d ticket words given, %d needed", n_tickets, tick);
  hipStream_t st = (hipStream_t)stream;
  if (L + 1 <= 16) GVK_LAUNCH(param_grads_kernel<1>, dim3(wg), dim3(256), 0, st, a);
  else GVK_LAUNCH(param_grads_kernel<2>, dim3(wg), dim3(256), 0, st, a);
  return check_launch("param_grads");
}
