// DVPT (model/dvpt.py, `--method dvpt`): the latent-space core of share_MLP (dvpt.py:37-47) and its backward.
//   z = proj_d(QuickGELU(x))            -> gvk_skinny_down (act_in = 1)
//   prompts attend to the patch latents  -> dvpt_cross_fwd      (this file; scale = d_model^-1/2, queries = the prompt latents)
//   proj_u([attended | cls | patches]) * prompt_gate, added to the MLP block's output -> gvk_skinny_up (lat_override, alpha_ptr)
// Backward, given dcomb = dy . W_u (unscaled by the gate):
//   dvpt_gate_grad   dgate = <dcomb, lat'> + <b_u, colsum(dy)>     (lat' = attended latents on the prompt rows, z elsewhere)
//   dvpt_cross_bwd_p per prompt: delta, dq -> dz of the prompt rows (they are the queries)
//   dvpt_bwd_tok     per cls / patch row: gate*dcomb + the gather over the prompts of the attention backward
//   gvk_scale_dev    x *= *alpha (dW_u, db_u are accumulated without the gate first)
// Cross attention itself is cross.hpp (shared with GPA).
#include "common.hpp"
#include "cross.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

struct DvptArgs {
  const float* z; const float* enh; const float* lse; const float* dcomb; const float* gate; const float* bu; const float* cs;
  float* enh_o; float* lse_o; float* delta; float* dz; float* dgate;
  int B, T, P, N, C;
  float scale;
};

template <int L>
__global__ __launch_bounds__(64) void dvpt_cross_fwd_kernel(DvptArgs p) {
  const int b = blockIdx.y, pi = blockIdx.x, lane = lane_id();
  const int ll_ = lane < L ? lane : 0;
  const float q_l = p.z[((size_t)b * p.T + pi) * L + ll_] * p.scale;
  float q[L], c[L], lse;
#pragma unroll
  for (int l = 0; l < L; ++l) q[l] = __shfl(q_l, l, 64);
  cross_one<L>(q, p.z + ((size_t)b * p.T + p.P + 1) * L, p.N, lane, c, lse);
  float c_l = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l) c_l = (lane == l) ? c[l] : c_l;
  if (lane < L) p.enh_o[((size_t)b * p.P + pi) * L + lane] = c_l;
  if (lane == 0) p.lse_o[b * p.P + pi] = lse;
}

template <int L>
__global__ __launch_bounds__(64) void dvpt_cross_bwd_p_kernel(DvptArgs p) {
  const int b = blockIdx.y, pi = blockIdx.x, lane = lane_id();
  const int ll_ = lane < L ? lane : 0;
  const bool in = lane < L;
  const float gate = p.gate[0];
  const size_t row = (size_t)b * p.T + pi, o = ((size_t)b * p.P + pi) * L;
  const float denh_l = gate * p.dcomb[row * L + ll_];
  const float del = wave_sum(in ? denh_l * p.enh[o + ll_] : 0.f);
  const float q_l = p.z[row * L + ll_] * p.scale;
  float dc[L], q[L], dq[L];
#pragma unroll
  for (int l = 0; l < L; ++l) { dc[l] = __shfl(denh_l, l, 64); q[l] = __shfl(q_l, l, 64); }
  cross_dq<L>(q, dc, p.z + ((size_t)b * p.T + p.P + 1) * L, p.N, lane, p.lse[b * p.P + pi], del, dq);
  float dq_l = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l) dq_l = (lane == l) ? dq[l] : dq_l;
  if (in) p.dz[row * L + lane] = dq_l * p.scale;        // the prompt latent only acts as the (scaled) query
  if (lane == 0) p.delta[b * p.P + pi] = del;
}

// cls / patch rows: dz = gate*dcomb (+ for patches the attention backward gathered over the P prompts, staged in LDS)
template <int L>
__global__ __launch_bounds__(256) void dvpt_bwd_tok_kernel(DvptArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.y, P = p.P;
  float* q_s = (float*)smem;              // [P][L] scaled queries
  float* dc_s = q_s + P * L;              // [P][L] gate * dcomb of the prompt rows
  float* ls_s = dc_s + P * L;             // [P] lse
  float* de_s = ls_s + P;                 // [P] delta
  const float gate = p.gate[0];
  for (int i = threadIdx.x; i < P * L; i += 256) {
    const size_t src = ((size_t)b * p.T) * L + i;
    q_s[i] = p.z[src] * p.scale;
    dc_s[i] = gate * p.dcomb[src];
  }
  for (int i = threadIdx.x; i < P; i += 256) { ls_s[i] = p.lse[b * P + i]; de_s[i] = p.delta[b * P + i]; }
  __syncthreads();
  const int t = P + blockIdx.x * 256 + threadIdx.x;         // rows P .. T-1
  if (t >= p.T) return;
  const size_t row = (size_t)b * p.T + t;
  float tok[L], g[L];
#pragma unroll
  for (int l = 0; l < L; ++l) { tok[l] = p.z[row * L + l]; g[l] = gate * p.dcomb[row * L + l]; }
  if (t > P) {
    for (int q = 0; q < P; ++q) {
      float d = 0.f, da = 0.f;
#pragma unroll
      for (int l = 0; l < L; ++l) { d = __builtin_fmaf(q_s[q * L + l], tok[l], d); da = __builtin_fmaf(dc_s[q * L + l], tok[l], da); }
      const float a = __expf(d - ls_s[q]);
      const float ds = a * (da - de_s[q]);
#pragma unroll
      for (int l = 0; l < L; ++l) g[l] += a * dc_s[q * L + l] + ds * q_s[q * L + l];
    }
  }
#pragma unroll
  for (int l = 0; l < L; ++l) p.dz[row * L + l] = g[l];
}

// dgate = sum_{m,l} dcomb[m][l] * lat'[m][l] + sum_c bu[c] * cs[c]   (one workgroup; deterministic)
template <int L>
__global__ __launch_bounds__(1024) void dvpt_gate_grad_kernel(DvptArgs p) {
  __shared__ float red[16];
  const long n = (long)p.B * p.T * L;
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) {
    const long m = i / L;
    const int l = (int)(i - m * L);
    const int b = (int)(m / p.T), t = (int)(m - (long)b * p.T);
    const float lat = t < p.P ? p.enh[((size_t)b * p.P + t) * L + l] : p.z[i];
    s = __builtin_fmaf(p.dcomb[i], lat, s);
  }
  for (int c = threadIdx.x; c < p.C; c += 1024) s = __builtin_fmaf(p.bu[c], p.cs[c], s);
  s = wave_sum(s);
  if (lane_id() == 0) red[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tsum = 0.f;
    for (int w = 0; w < 16; ++w) tsum += red[w];
    p.dgate[0] = tsum;
  }
}

__global__ __launch_bounds__(256) void scale_dev_kernel(float* x, const float* alpha, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] *= alpha[0];
}

}  // namespace gvk

#define GVK_DVPT_LAUNCH(KERNEL, grid, block, lds)                                                   \
  switch (d->L) {                                                                                   \
    case 20: GVK_LAUNCH((KERNEL<20>), grid, block, lds, s, a); break;                               \
    default: return set_error(-2, "gvk_dvpt: latent width %d unsupported (share_MLP fixes it at 20, dvpt.py:27)", d->L); \
  }

static void fill_dvpt(gvk::DvptArgs& a, const gvk_dvpt_desc* d) {
  a.z = d->z; a.enh = d->enh; a.lse = d->lse; a.dcomb = d->dcomb; a.gate = d->gate; a.bu = d->bu; a.cs = d->colsum_dy;
  a.enh_o = d->enh; a.lse_o = d->lse; a.delta = d->delta; a.dz = d->dz; a.dgate = d->dgate;
  a.B = d->B; a.T = d->T; a.P = d->P; a.N = d->T - d->P - 1; a.C = d->C; a.scale = d->scale;
}

extern "C" int gvk_dvpt_fwd(const gvk_dvpt_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->z && d->enh && d->lse, "gvk_dvpt_fwd: null pointer");
  GVK_REQUIRE(d->B > 0 && d->P > 0 && d->T > d->P + 1, "gvk_dvpt_fwd: need T > P + 1 (prompts | cls | patches)");
  DvptArgs a{};
  fill_dvpt(a, d);
  hipStream_t s = (hipStream_t)stream;
  GVK_DVPT_LAUNCH(dvpt_cross_fwd_kernel, dim3(d->P, d->B), dim3(64), 0);
  return check_launch("dvpt_cross_fwd");
}

extern "C" int gvk_dvpt_bwd(const gvk_dvpt_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->z && d->enh && d->lse && d->dcomb && d->gate && d->bu && d->colsum_dy && d->delta && d->dz && d->dgate,
              "gvk_dvpt_bwd: null pointer");
  GVK_REQUIRE(d->B > 0 && d->P > 0 && d->T > d->P + 1 && d->C > 0, "gvk_dvpt_bwd: bad shape");
  GVK_REQUIRE((2 * d->P * d->L + 2 * d->P) * 4 <= 64 * 1024, "gvk_dvpt_bwd: too many prompts for the LDS stage");
  DvptArgs a{};
  fill_dvpt(a, d);
  hipStream_t s = (hipStream_t)stream;
  GVK_DVPT_LAUNCH(dvpt_gate_grad_kernel, dim3(1), dim3(1024), 0);
  int rc = check_launch("dvpt_gate_grad");
  if (rc) return rc;
  GVK_DVPT_LAUNCH(dvpt_cross_bwd_p_kernel, dim3(d->P, d->B), dim3(64), 0);
  rc = check_launch("dvpt_cross_bwd_p");
  if (rc) return rc;
  const int lds = (2 * d->P * d->L + 2 * d->P) * 4;
  GVK_DVPT_LAUNCH(dvpt_bwd_tok_kernel, dim3((d->T - d->P + 255) / 256, d->B), dim3(256), lds);
  return check_launch("dvpt_bwd_tok");
}

extern "C" int gvk_scale_dev(float* x, const float* alpha, int64_t n, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x && alpha && n >= 0, "gvk_scale_dev: bad arguments");
  if (n == 0) return 0;
  GVK_LAUNCH(scale_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, alpha, (long)n);
  return check_launch("scale_dev");
}
