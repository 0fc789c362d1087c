// Cross attention of one query vector against a set of rank-L token latents read straight from global memory (GPA: gpa.hip,
// DVPT: dvpt.hip).  One wave per (query, token set): lanes over tokens, four tokens per lane in flight, online softmax per lane,
// merged across lanes at the end.
#pragma once
#include "common.hpp"

namespace gvk {

template <int L>
__device__ __forceinline__ void load_tok(const float* __restrict__ src, int i, int n, float (&t)[L]) {
  const f32x4* r = (const f32x4*)(src + (size_t)min(i, n - 1) * L);
#pragma unroll
  for (int v = 0; v < L / 4; ++v) {
    const f32x4 x = r[v];
    t[4 * v] = x[0]; t[4 * v + 1] = x[1]; t[4 * v + 2] = x[2]; t[4 * v + 3] = x[3];
  }
}

// softmax(q . tok^T) . tok over n tokens; returns ctx (all lanes) and lse.  U tokens per lane in flight.
template <int L, int U = 4>
__device__ __forceinline__ void cross_one(const float (&q)[L], const float* __restrict__ src, int n, int lane, float (&ctx)[L], float& lse) {
  float m = -INFINITY, s = 0.f, c[L];
#pragma unroll
  for (int l = 0; l < L; ++l) c[l] = 0.f;
  for (int i0 = lane; i0 < n; i0 += 64 * U) {
    float t[U][L], d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) load_tok<L>(src, i0 + 64 * u, n, t[u]);
    float mb = -INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float a = 0.f;
#pragma unroll
      for (int l = 0; l < L; ++l) a = __builtin_fmaf(q[l], t[u][l], a);
      d[u] = (i0 + 64 * u < n) ? a : -INFINITY;
      mb = fmaxf(mb, d[u]);
    }
    const float mn = fmaxf(m, mb);                       // finite: token i0 itself is valid
    const float sc = __expf(m - mn);                     // first batch: exp(-inf) = 0
    s *= sc;
#pragma unroll
    for (int l = 0; l < L; ++l) c[l] *= sc;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float e = __expf(d[u] - mn);                 // invalid token: exp(-inf) = 0
      s += e;
#pragma unroll
      for (int l = 0; l < L; ++l) c[l] = __builtin_fmaf(e, t[u][l], c[l]);
    }
    m = mn;
  }
  const float mw = wave_max(m);
  const float f = (m == -INFINITY) ? 0.f : __expf(m - mw);          // lanes without tokens contribute nothing
  const float st = wave_sum(s * f);
  lse = mw + __logf(st);
  const float inv = 1.f / st;
#pragma unroll
  for (int l = 0; l < L; ++l) ctx[l] = wave_sum(c[l] * f) * inv;
}

// dq (already-scaled query space) of softmax cross attention: dq[l] = sum_n A_n (dA_n - delta) tok_n[l].  U tokens per lane in flight
// (4 = one round trip for ~250 tokens per wave; 2 keeps the wave under 128 registers at L = 20)
template <int L, int U = 4>
__device__ __forceinline__ void cross_dq(const float (&q)[L], const float (&dc)[L], const float* __restrict__ src, int n, int lane, float lse,
                                         float delta, float (&dq)[L]) {
  float a[L];
#pragma unroll
  for (int l = 0; l < L; ++l) a[l] = 0.f;
  for (int i0 = lane; i0 < n; i0 += 64 * U) {
    float t[U][L];
#pragma unroll
    for (int u = 0; u < U; ++u) load_tok<L>(src, i0 + 64 * u, n, t[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float d = 0.f, da = 0.f;
#pragma unroll
      for (int l = 0; l < L; ++l) { d = __builtin_fmaf(q[l], t[u][l], d); da = __builtin_fmaf(dc[l], t[u][l], da); }
      const float ds = (i0 + 64 * u < n) ? __expf(d - lse) * (da - delta) : 0.f;
#pragma unroll
      for (int l = 0; l < L; ++l) a[l] = __builtin_fmaf(ds, t[u][l], a[l]);
    }
  }
#pragma unroll
  for (int l = 0; l < L; ++l) dq[l] = wave_sum(a[l]);
}

}  // namespace gvk
