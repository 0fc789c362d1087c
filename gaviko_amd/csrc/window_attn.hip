// Masked-window local self-attention (MWSA) core for GAViKO, fp32, single head over an L-dim latent.
// Replaces gaviko.py:235-241: q@k^T * dim^-0.5 + dense 0/-inf [N,N] mask -> softmax -> attn_drop -> @v.
// The reference materialises N x N scores and re-uploads a 4 MB mask every call; here the window is index
// arithmetic on the (D,H,W) patch grid: query (d,h,w) sees keys with  d - dk/2 <= d' < d - dk/2 + dk  (same for h,w),
// clipped to the grid (27..216 keys at local_k = 6,6,6), so only the live keys are ever touched.
// One 64-lane wave per token row, lanes spread over the keys of the window; scores are recomputed in the second pass
// instead of stored (20-MAC dot products against an L2-resident 240 KB/sample q|k|v matrix).
// Backward is split in a query-side pass (dq, delta) and a key-side pass over the REVERSE window (dk, dv): no atomics.
#include "common.hpp"
#include "window_args.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

template <int L>
__global__ __launch_bounds__(256) void win_fwd_kernel(WinArgs p) {
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  const int N = p.D * p.H * p.W;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.B * N) return;
  const int lane = lane_id();
  const int b = row / N, i = row - b * N;
  const int qd = i / (p.H * p.W), qh = (i / p.W) % p.H, qw = i % p.W;
  Win w;
  axis_fwd(qd, p.kd, p.D, w.d0, w.nd);
  axis_fwd(qh, p.kh, p.H, w.h0, w.nh);
  axis_fwd(qw, p.kw, p.W, w.w0, w.nw);
  const int nk = w.count();
  const float* base = p.qkv + (size_t)b * N * 3 * L;
  float q[L];
#pragma unroll
  for (int l = 0; l < L; ++l) q[l] = base[(size_t)i * 3 * L + l] * p.scale;
  // pass 1: online max / sum over this lane's keys
  float m = -INFINITY, s = 0.f;
  for (int kk = lane; kk < nk; kk += 64) {
    const float* kr = base + (size_t)w.index(kk, p.H, p.W) * 3 * L + L;
    float d = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) d += q[l] * kr[l];
    const float mn = fmaxf(m, d);
    s = s * __expf(m - mn) + __expf(d - mn);
    m = mn;
  }
  const float mw = wave_max(m);
  s = wave_sum(s * __expf(m - mw));   // lanes without keys: m = -inf -> exp(-inf) = 0
  const float lse = mw + __logf(s);
  // pass 2: probabilities (recomputed), dropout, context
  float c[L];
#pragma unroll
  for (int l = 0; l < L; ++l) c[l] = 0.f;
  for (int kk = lane; kk < nk; kk += 64) {
    const int j = w.index(kk, p.H, p.W);
    const float* kr = base + (size_t)j * 3 * L + L;
    float d = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) d += q[l] * kr[l];
    float pr = __expf(d - lse);
    if (p.drop_thresh != 0u)
      pr *= (hash_u32_w(p.seed, (unsigned long long)row * N + j) >= p.drop_thresh) ? p.inv_keep : 0.f;
    const float* vr = kr + L;
#pragma unroll
    for (int l = 0; l < L; ++l) c[l] += pr * vr[l];
  }
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float t = wave_sum(c[l]);
    if (lane == l) p.ctx[(size_t)row * L + l] = t;
  }
  if (lane == 0 && p.lse) p.lse[row] = lse;
}

// query side: delta_i = dctx_i . ctx_i ; dq_i = scale * sum_j ds_ij k_j
template <int L>
__global__ __launch_bounds__(256) void win_bwd_q_kernel(WinArgs p) {
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  const int N = p.D * p.H * p.W;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.B * N) return;
  const int lane = lane_id();
  const int b = row / N, i = row - b * N;
  const int qd = i / (p.H * p.W), qh = (i / p.W) % p.H, qw = i % p.W;
  Win w;
  axis_fwd(qd, p.kd, p.D, w.d0, w.nd);
  axis_fwd(qh, p.kh, p.H, w.h0, w.nh);
  axis_fwd(qw, p.kw, p.W, w.w0, w.nw);
  const int nk = w.count();
  const float* base = p.qkv + (size_t)b * N * 3 * L;
  float q[L], dc[L];
  float delta = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    q[l] = base[(size_t)i * 3 * L + l] * p.scale;
    dc[l] = p.dctx[(size_t)row * L + l];
    delta += dc[l] * p.ctx[(size_t)row * L + l];
  }
  const float lse = p.lse[row];
  float dq[L];
#pragma unroll
  for (int l = 0; l < L; ++l) dq[l] = 0.f;
  for (int kk = lane; kk < nk; kk += 64) {
    const int j = w.index(kk, p.H, p.W);
    const float* kr = base + (size_t)j * 3 * L + L;
    const float* vr = kr + L;
    float d = 0.f, dp = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) { d += q[l] * kr[l]; dp += dc[l] * vr[l]; }
    const float pr = __expf(d - lse);
    if (p.drop_thresh != 0u)
      dp *= (hash_u32_w(p.seed, (unsigned long long)row * N + j) >= p.drop_thresh) ? p.inv_keep : 0.f;
    const float ds = pr * (dp - delta) * p.scale;
#pragma unroll
    for (int l = 0; l < L; ++l) dq[l] += ds * kr[l];
  }
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float t = wave_sum(dq[l]);
    if (lane == l) p.dqkv[(size_t)row * 3 * L + l] = t;
  }
  if (lane == 0) p.delta[row] = delta;
}

// key side over the reverse window: dk_j = scale * sum_i ds_ij q_i ; dv_j = sum_i p~_ij dctx_i
template <int L>
__global__ __launch_bounds__(256) void win_bwd_kv_kernel(WinArgs p) {
  if (p.drop_thresh != 0u && p.seed_ptr != nullptr) p.seed += *p.seed_ptr;
  const int N = p.D * p.H * p.W;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.B * N) return;
  const int lane = lane_id();
  const int b = row / N, j = row - b * N;
  const int kd_ = j / (p.H * p.W), kh_ = (j / p.W) % p.H, kw_ = j % p.W;
  Win w;
  axis_rev(kd_, p.kd, p.D, w.d0, w.nd);
  axis_rev(kh_, p.kh, p.H, w.h0, w.nh);
  axis_rev(kw_, p.kw, p.W, w.w0, w.nw);
  const int nq = w.count();
  const float* base = p.qkv + (size_t)b * N * 3 * L;
  float k[L], v[L], dk[L], dv[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    k[l] = base[(size_t)j * 3 * L + L + l];
    v[l] = base[(size_t)j * 3 * L + 2 * L + l];
    dk[l] = 0.f; dv[l] = 0.f;
  }
  for (int qq = lane; qq < nq; qq += 64) {
    const int i = w.index(qq, p.H, p.W);
    const size_t gi = (size_t)b * N + i;
    const float* qr = base + (size_t)i * 3 * L;
    const float* dcr = p.dctx + gi * L;
    float d = 0.f, dp = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) { d += qr[l] * k[l]; dp += dcr[l] * v[l]; }
    float pr = __expf(d * p.scale - p.lse[gi]);
    float msk = 1.f;
    if (p.drop_thresh != 0u) msk = (hash_u32_w(p.seed, (unsigned long long)gi * N + j) >= p.drop_thresh) ? p.inv_keep : 0.f;
    const float ds = pr * (dp * msk - p.delta[gi]) * p.scale;
    pr *= msk;
#pragma unroll
    for (int l = 0; l < L; ++l) { dk[l] += ds * qr[l]; dv[l] += pr * dcr[l]; }
  }
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float a = wave_sum(dk[l]), c = wave_sum(dv[l]);
    if (lane == l) {
      p.dqkv[(size_t)row * 3 * L + L + l] = a;
      p.dqkv[(size_t)row * 3 * L + 2 * L + l] = c;
    }
  }
}

static int fill(WinArgs& a, const gvk_window_attn_desc* d) {
  a.qkv = d->qkv; a.ctx = d->ctx; a.lse = d->lse; a.dctx = d->dctx; a.delta = d->delta; a.dqkv = d->dqkv;
  a.B = d->B; a.D = d->D; a.H = d->H; a.W = d->W; a.kd = d->kd; a.kh = d->kh; a.kw = d->kw; a.scale = d->scale;
  a.seed = d->seed; a.seed_ptr = (const unsigned long long*)d->seed_ptr;
  a.drop_thresh = 0u; a.inv_keep = 1.f;
  if (d->drop_p > 0.f) {
    double t = (double)d->drop_p * 4294967296.0;
    a.drop_thresh = (unsigned int)(t > 4294967295.0 ? 4294967295.0 : t);
    a.inv_keep = 1.f / (1.f - d->drop_p);
  }
  return 0;
}

}  // namespace gvk

#define GVK_WIN_DISPATCH(KERNEL, what)                                                                             \
  switch (d->L) {                                                                                                  \
    case 4: GVK_LAUNCH((KERNEL<4>), dim3(grid), dim3(256), 0, s, a); break;                               \
    case 8: GVK_LAUNCH((KERNEL<8>), dim3(grid), dim3(256), 0, s, a); break;                               \
    case 16: GVK_LAUNCH((KERNEL<16>), dim3(grid), dim3(256), 0, s, a); break;                             \
    case 20: GVK_LAUNCH((KERNEL<20>), dim3(grid), dim3(256), 0, s, a); break;                             \
    case 32: GVK_LAUNCH((KERNEL<32>), dim3(grid), dim3(256), 0, s, a); break;                             \
    default: return set_error(-2, what ": L=%d unsupported (4, 8, 16, 20, 32)", d->L);                            \
  }

extern "C" int gvk_window_attn_fwd(const gvk_window_attn_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->qkv && d->ctx, "gvk_window_attn_fwd: null pointer");
  GVK_REQUIRE(d->B > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->kd > 0 && d->kh > 0 && d->kw > 0, "gvk_window_attn_fwd: bad grid/window");
  WinArgs a{};
  fill(a, d);
  hipStream_t s = (hipStream_t)stream;
  {
    const int rc = launch_win_mfma_fwd(a, d->L, s);
    if (rc != 1) return rc;
  }
  const int grid = (d->B * d->D * d->H * d->W + 3) / 4;
  GVK_WIN_DISPATCH(win_fwd_kernel, "gvk_window_attn_fwd");
  return check_launch("window_attn_fwd");
}

extern "C" int gvk_window_attn_bwd(const gvk_window_attn_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->qkv && d->ctx && d->lse && d->dctx && d->delta && d->dqkv, "gvk_window_attn_bwd: null pointer");
  GVK_REQUIRE(d->B > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->kd > 0 && d->kh > 0 && d->kw > 0, "gvk_window_attn_bwd: bad grid/window");
  WinArgs a{};
  fill(a, d);
  hipStream_t s = (hipStream_t)stream;
  {
    const int rc = launch_win_mfma_bwd(a, d->L, s);
    if (rc != 1) return rc;
  }
  const int grid = (d->B * d->D * d->H * d->W + 3) / 4;
  GVK_WIN_DISPATCH(win_bwd_q_kernel, "gvk_window_attn_bwd");
  int rc = check_launch("window_attn_bwd/q");
  if (rc) return rc;
  GVK_WIN_DISPATCH(win_bwd_kv_kernel, "gvk_window_attn_bwd");
  return check_launch("window_attn_bwd/kv");
}
