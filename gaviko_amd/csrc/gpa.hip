// Gated prompt awakening (GPA) core for GAViKO in the L-dim latent space, fp32.
// Replaces gaviko.py:159-185 (Awakening_Prompt.forward between proj_down and proj_up):
//   PRE gate   imp[b][p] = sigmoid(W3 . GELU(W1 . LN(cls) + b1) + b3)                       gaviko.py:23-29,164
//   PCF weight gw[b]     = sigmoid(wg . LN'(cls) + bg)                                       gaviko.py:51-55,167
//   GXA / LXA  ctx = softmax(q . tok^T * L^-1/2) . tok, q = Wq . prompt + bq                 gaviko.py:84-94
//              global tokens = rows 2P+2 .. T-1 of the global latent (the reference slices [P+1:] twice: 106-107,170)
//              local tokens  = all N local latents                                            gaviko.py:118-119,172
//   enh[b][p]  = (gw * ctx_g + (1 - gw) * ctx_l) * imp[b][p]                                 gaviko.py:175,178
// and the whole backward of the above.  ~25 tiny ATen launches per layer become ONE forward and ONE backward kernel.
// One wave per (sample, prompt) for the cross-attention (lanes over the ~1000 tokens); the token-side gradient is a
// gather over the P prompts (no atomics).
#include "common.hpp"
#include "cross.hpp"
#include "../../include/gaviko_hip.h"

#ifndef GVK_GPA_FWD_U
#define GVK_GPA_FWD_U 4
#endif
#ifndef GVK_GPA_BWD_U
#define GVK_GPA_BWD_U 4
#endif

namespace gvk {

struct GpaArgs {
  // forward inputs
  const float* xl; const float* ll;          // activated latents: global [B*T][L], local [B*N][L]
  const float* ca0_g; const float* ca0_b; const float* ca1_w; const float* ca1_b; const float* ca3_w; const float* ca3_b;
  const float* gl0_g; const float* gl0_b; const float* gl1_w; const float* gl1_b;
  const float* wgq; const float* bgq; const float* wlq; const float* blq;
  // forward outputs / saved
  float* imp; float* gw;                     // [B][P], [B]
  float* enh;                                // [B][P][L]
  bf16* enh16; int ld16, col16;              // optional split-bf16 copy of enh into row (b*T + p), columns col16.. of a GEMM operand (stride ld16)
  float* prm; float* qg; float* ql; float* cg; float* cl; float* lse_g; float* lse_l;   // [B][P][L] x5, [B][P] x2
  // backward
  const float* dcomb;                        // [B*T][L]  gradient wrt the combined latent (proj_up input)
  const float* zx; const float* zl;          // pre-activations of proj_down (QuickGELU'), [B*T][L], [B*N][L]
  float* dimp; float* dgw_part;              // [B][P]
  float* dqg; float* dql; float* dcg; float* dcl; float* delta_g; float* delta_l; float* dprm;
  float* dcls;                               // [B][L] gradient wrt the CLS latent from both gates
  float* gate_partials;                      // [B][n_gate]
  float* dzx; float* dzl;                    // outputs: gradient wrt proj_down pre-activations
  int B, T, N, P;
  float scale;
};

template <int L>
__device__ __forceinline__ void ln_small(const float* x, const float* g, const float* b, float* out, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l) s += x[l];
  mean = s / L;
  float q = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l) { const float d = x[l] - mean; q += d * d; }
  rstd = rsqrtf(q / L + 1e-5f);
#pragma unroll
  for (int l = 0; l < L; ++l) out[l] = (x[l] - mean) * rstd * g[l] + b[l];
}
// dx += LN backward of (dy) given x-hat = (x-mean)*rstd
template <int L>
__device__ __forceinline__ void ln_small_bwd(const float* x, float mean, float rstd, const float* g, const float* dy, float* dx_acc) {
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l) { const float xh = (x[l] - mean) * rstd, dh = dy[l] * g[l]; m1 += dh; m2 += dh * xh; }
  m1 /= L; m2 /= L;
#pragma unroll
  for (int l = 0; l < L; ++l) { const float xh = (x[l] - mean) * rstd; dx_acc[l] += rstd * (dy[l] * g[l] - m1 - xh * m2); }
}

// ---- forward: ONE launch, workgroup = one prompt of one sample, nine waves.
// Waves 0-3 each take a quarter of the sample's global image tokens (gaviko.py:172-176, the reference's double slice: tokens 2P+2..),
// waves 4-7 a quarter of its local tokens (:177-181): one round of four 80-byte rows per lane each, read straight from global memory
// (the per-sample sets are 80 KB and L2-resident), merged through LDS in a fixed order.  Wave 8 evaluates the two gates of the sample
// (cls_analyzer -> importance of THIS prompt, gl_balancer -> global/local weight; :160-170) meanwhile; wave 0 then fuses (:183-185).
// The GPA result gates the MLP's second GEMM of the layer (the up-projection rides it as extra K columns), so the chain
// LayerNorm -> gates -> cross-attention -> fuse is one dependent launch instead of two and a quarter as long.
// No LDS staging of the tokens: 32 fat workgroups would have to wait for the backbone's GEMM workgroups to retire; these fit beside them.
template <int L>
__global__ __launch_bounds__(576) void gpa_fwd_kernel(GpaArgs p) {
  __shared__ float part_c[8][L], part_lse[8], gate_s[2];
  const int b = blockIdx.y, pi = blockIdx.x, wave = wave_id(), lane = lane_id();
  const size_t o = ((size_t)b * p.P + pi) * L;
  const int ll_ = lane < L ? lane : 0;
  if (wave == 8) {                                       // gates
    __shared__ float dummy;
    (void)dummy;
    float cls[L], hn[L], gn[L];
#pragma unroll
    for (int l = 0; l < L; ++l) cls[l] = p.xl[((size_t)b * p.T + p.P) * L + l];
    float mean, rstd;
    ln_small<L>(cls, p.ca0_g, p.ca0_b, hn, mean, rstd);
    float a = p.ca1_b[lane];
#pragma unroll
    for (int l = 0; l < L; ++l) a += p.ca1_w[lane * L + l] * hn[l];
    const float t3 = p.ca3_b[pi] + wave_sum(p.ca3_w[pi * 64 + lane] * gelu_erf(a));
    ln_small<L>(cls, p.gl0_g, p.gl0_b, gn, mean, rstd);
    float t = p.gl1_b[0];
#pragma unroll
    for (int l = 0; l < L; ++l) t += p.gl1_w[l] * gn[l];
    if (lane == 0) {
      const float im = sigmoidf_(t3), gw = sigmoidf_(t);
      gate_s[0] = im; gate_s[1] = gw;
      p.imp[b * p.P + pi] = im;
      if (pi == 0) p.gw[b] = gw;
    }
    __syncthreads();
    return;
  }
  const int side = wave >> 2, quarter = wave & 3;
  // lane l owns element l of the per-prompt vectors while they are formed; they are then spread with lane shuffles
  const float pr_l = p.xl[((size_t)b * p.T + pi) * L + ll_];
  const float* wq = side == 0 ? p.wgq : p.wlq;
  float q_l = (side == 0 ? p.bgq : p.blq)[ll_];
#pragma unroll
  for (int l = 0; l < L; ++l) q_l = __builtin_fmaf(wq[ll_ * L + l], __shfl(pr_l, l, 64), q_l);
  q_l *= p.scale;                                        // scale folded into the query
  float q[L], c[L], lse;
#pragma unroll
  for (int l = 0; l < L; ++l) q[l] = __shfl(q_l, l, 64);
  const float* base = side == 0 ? p.xl + ((size_t)b * p.T + 2 * p.P + 2) * L : p.ll + (size_t)b * p.N * L;
  const int n = side == 0 ? p.T - (2 * p.P + 2) : p.N;
  const int per = (n + 3) >> 2, lo = quarter * per, cnt = min(per, n - lo);
  if (cnt > 0) {
    cross_one<L, GVK_GPA_FWD_U>(q, base + (size_t)lo * L, cnt, lane, c, lse);
  } else {
    lse = -INFINITY;
#pragma unroll
    for (int l = 0; l < L; ++l) c[l] = 0.f;
  }
  float c_l = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l) c_l = (lane == l) ? c[l] : c_l;
  if (lane < L) part_c[wave][lane] = c_l;
  if (lane == 0) part_lse[wave] = lse;
  if (wave == 4 && lane < L) p.ql[o + lane] = q_l;
  __syncthreads();
  if (wave != 0) return;
  float cs[2], ls[2];
#pragma unroll
  for (int sd = 0; sd < 2; ++sd) {                       // fixed-order merge of the four quarters (quarter 0 is never empty)
    float m = part_lse[4 * sd];
#pragma unroll
    for (int k = 1; k < 4; ++k) m = fmaxf(m, part_lse[4 * sd + k]);
    float st = 0.f, acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float w = __expf(part_lse[4 * sd + k] - m);  // empty quarter: exp(-inf) = 0
      st += w;
      acc = __builtin_fmaf(w, part_c[4 * sd + k][ll_], acc);
    }
    cs[sd] = acc / st;
    ls[sd] = m + __logf(st);
  }
  const float im = gate_s[0], gw = gate_s[1];
  if (lane < L) {
    const float e = (gw * cs[0] + (1.f - gw) * cs[1]) * im;
    p.enh[o + lane] = e;
    if (p.enh16 != nullptr) {                          // split-bf16 form [hi | lo | hi] (elementwise.hip: pack_split_bf16_kernel)
      bf16* d16 = p.enh16 + ((size_t)b * p.T + pi) * p.ld16 + p.col16;
      const bf16 hi = (bf16)e;
      d16[lane] = hi; d16[L + lane] = (bf16)(e - (float)hi); d16[2 * L + lane] = hi;
    }
    p.prm[o + lane] = pr_l; p.qg[o + lane] = q_l; p.cg[o + lane] = cs[0]; p.cl[o + lane] = cs[1];
  }
  if (lane == 0) { p.lse_g[b * p.P + pi] = ls[0]; p.lse_l[b * p.P + pi] = ls[1]; }
}

// ---- backward (round 4): ONE launch.  The grid of a sample is [token-side workgroups | the gates' workgroup | one workgroup per prompt].
// Nothing in it waits for another workgroup: what the token side and the gates need of the prompt side (dctx, delta, d importance, d balance)
// are a few hundred flops per prompt from tensors the FORWARD saved (cg, cl, imp, gw) and the incoming gradient's prompt rows, so they
// recompute them instead of reading them from a kernel that had to finish first; the only product of the prompt side's token loop, dq,
// goes to the prompt's own latent row (which the prompt workgroup writes itself) and to the query projections' parameter gradients.
// Three dependent kernels (prompt side -> gates -> token side, ~75 us of latency per layer in round 2, two kernels / ~50 us in round 3)
// are one kernel of max(...) instead of sum(...).
//
// prompt side: eight waves, waves 0-3 a quarter each of the global tokens, waves 4-7 of the local tokens (dq is a plain sum over tokens:
// the quarters' partial sums are added through LDS in a fixed order); waves 0 and 4 finish.
template <int L>
__device__ __forceinline__ void gpa_bwd_prompt_body(const GpaArgs& p, const int b, const int pi, float (*dq_s)[L], float* dpr_s) {
  const int wave = wave_id(), lane = lane_id();
  const int side = wave >> 2, quarter = wave & 3;
  const size_t o = ((size_t)b * p.P + pi) * L;
  const float gw = p.gw[b], im = p.imp[b * p.P + pi];
  const int ll_ = lane < L ? lane : 0;
  const bool in = lane < L;
  const size_t prow = (size_t)b * p.T + pi;
  const float denh_l = p.dcomb[prow * L + ll_];
  const float cg_l = p.cg[o + ll_], cl_l = p.cl[o + ll_];
  const float zx_l = p.zx[prow * L + ll_];
  const float fused_l = gw * cg_l + (1.f - gw) * cl_l;
  const float df_l = denh_l * im;
  const float dcg_l = gw * df_l, dcl_l = (1.f - gw) * df_l;
  const float dc_l = side == 0 ? dcg_l : dcl_l;
  const float del = wave_sum(in ? dc_l * (side == 0 ? cg_l : cl_l) : 0.f);
  const float q_l = (side == 0 ? p.qg : p.ql)[o + ll_];
  float dc[L], q[L], dq[L];
#pragma unroll
  for (int l = 0; l < L; ++l) { dc[l] = __shfl(dc_l, l, 64); q[l] = __shfl(q_l, l, 64); }
  const float* base = side == 0 ? p.xl + ((size_t)b * p.T + 2 * p.P + 2) * L : p.ll + (size_t)b * p.N * L;
  const int n = side == 0 ? p.T - (2 * p.P + 2) : p.N;
  const int per = (n + 3) >> 2, lo = quarter * per, cnt = max(0, min(per, n - lo));
  cross_dq<L, GVK_GPA_BWD_U>(q, dc, base + (size_t)lo * L, cnt, lane, (side == 0 ? p.lse_g : p.lse_l)[b * p.P + pi], del, dq);   // cnt = 0: zeros
  float dq_l = 0.f;
#pragma unroll
  for (int j = 0; j < L; ++j) dq_l = (lane == j) ? dq[j] : dq_l;
  if (in) dq_s[wave][lane] = dq_l;
  __syncthreads();
  if (quarter != 0) return;                              // waves 0 (global) and 4 (local) carry on with the summed dq of their side
  dq_l = in ? ((dq_s[wave][ll_] + dq_s[wave + 1][ll_]) + (dq_s[wave + 2][ll_] + dq_s[wave + 3][ll_])) * p.scale : 0.f;
  // unscaled-query gradients (q_scaled = scale * (W prompt + b)), and this query path's share of the prompt latent gradient
  const float* wq = side == 0 ? p.wgq : p.wlq;
  float dpr_l = 0.f;
#pragma unroll
  for (int j = 0; j < L; ++j) dpr_l = __builtin_fmaf(wq[j * L + ll_], __shfl(dq_l, j, 64), dpr_l);
  if (side == 1) {
    if (in) { dpr_s[lane] = dpr_l; p.dql[o + lane] = dq_l; p.dcl[o + lane] = dcl_l; }
    if (lane == 0) p.delta_l[b * p.P + pi] = del;
  }
  __syncthreads();                                       // (waves 0 and 4 only: the others have left)
  if (side == 0) {
    if (in) {
      const float dpr = dpr_l + dpr_s[lane];
      p.dqg[o + lane] = dq_l; p.dcg[o + lane] = dcg_l; p.dprm[o + lane] = dpr;
      p.dzx[prow * L + lane] = dpr * quick_gelu_grad(zx_l);       // prompt rows only feed the queries (gaviko.py:159,84-94)
    }
    const float dimp = wave_sum(in ? denh_l * fused_l : 0.f);
    const float dgw = wave_sum(in ? df_l * (cg_l - cl_l) : 0.f);
    if (lane == 0) { p.dimp[b * p.P + pi] = dimp; p.dgw_part[b * p.P + pi] = dgw; p.delta_g[b * p.P + pi] = del; }
  }
}

// ---- gates backward: one wave per sample; recomputes the tiny forward.  Per-sample parameter-gradient partials are
// written to gate_partials[b][:] in the order [ca0_g L | ca0_b L | ca1_w 64L | ca1_b 64 | ca3_w 64P | ca3_b P | gl0_g L | gl0_b L | gl1_w L | gl1_b 1].
// (device body: runs on ONE wave -- the extra workgroup per sample of gpa_bwd_tok_kernel, whose other waves have left, so the barriers below
//  only order this wave's own LDS traffic)
template <int L>
__device__ __forceinline__ void gpa_gates_bwd_body(const GpaArgs& p, const int b, const int lane, float (&dcls)[L]) {
  __shared__ float a1_s[64], da1_s[64], dpre3_s[64], dhn_s[64][L + 1];
  __shared__ float w3_s[64][65];                           // ca3_w rows (q < P), +1 pad
  const int P = p.P;
  const int n_gate = 4 * L + 64 * L + 64 + 64 * P + P + L + 1;
  float* out = p.gate_partials + (size_t)b * n_gate;
  float* o_ca0g = out; float* o_ca0b = out + L; float* o_ca1w = out + 2 * L; float* o_ca1b = o_ca1w + 64 * L;
  float* o_ca3w = o_ca1b + 64; float* o_ca3b = o_ca3w + 64 * P; float* o_gl0g = o_ca3b + P; float* o_gl0b = o_gl0g + L;
  float* o_gl1w = o_gl0b + L; float* o_gl1b = o_gl1w + L;
  // One wave per sample on a busy chip: every dependent round trip to memory costs microseconds, so EVERYTHING this kernel reads is
  // requested here, before the first use (the loop form interleaved loads with stores to `out` and paid ~8 round trips: 24 us).
  float cls[L], hn[L], gn[L];
#pragma unroll
  for (int l = 0; l < L; ++l) { cls[l] = p.xl[((size_t)b * p.T + P) * L + l]; dcls[l] = 0.f; }
  for (int q = 0; q < P && q < 64; ++q) w3_s[q][lane] = p.ca3_w[q * 64 + lane];
  const float b1 = p.ca1_b[lane];
  // d importance / d balance of prompt q = lane, from what the forward saved and the incoming gradient's prompt rows (gaviko.py:175,178):
  //   dimp_q = sum_l denh[q][l] fused[q][l],  dgw_q = sum_l denh[q][l] imp_q (cg - cl)[q][l],  fused = gw cg + (1 - gw) cl
  const float gwv = p.gw[b];
  float d3 = 0.f, dgw = 0.f;
  if (lane < P) {
    const float im = p.imp[b * P + lane];
    const size_t o = ((size_t)b * P + lane) * L;
    float dimp = 0.f;
#pragma unroll 4
    for (int l = 0; l < L; ++l) {
      const float de = p.dcomb[((size_t)b * p.T + lane) * L + l], cgv = p.cg[o + l], clv = p.cl[o + l];
      dimp = __builtin_fmaf(de, gwv * cgv + (1.f - gwv) * clv, dimp);
      dgw = __builtin_fmaf(de * im, cgv - clv, dgw);
    }
    d3 = dimp * im * (1.f - im);
  }
  float mean_a, rstd_a, mean_g, rstd_g;
  ln_small<L>(cls, p.ca0_g, p.ca0_b, hn, mean_a, rstd_a);
  float pre1 = b1;
#pragma unroll
  for (int l = 0; l < L; ++l) pre1 += p.ca1_w[lane * L + l] * hn[l];
  a1_s[lane] = gelu_erf(pre1);
  // layer 3 (P outputs): dpre3[q] = dimp * imp * (1 - imp)
  if (lane < P) o_ca3b[lane] = d3;
  dpre3_s[lane] = d3;
  __syncthreads();
  // d ca3_w[q][u] = dpre3[q] * a1[u];  da1[u] = sum_q dpre3[q] * ca3_w[q][u]   (lane = u)
  {
    float da = 0.f;
    const float a1 = a1_s[lane];
    for (int q = 0; q < P && q < 64; ++q) {
      const float dq3 = dpre3_s[q];
      o_ca3w[q * 64 + lane] = dq3 * a1;
      da += dq3 * w3_s[q][lane];
    }
    da1_s[lane] = da;
  }
  // layer 1 (lane = u): dpre1 = da1 * GELU'(pre1)
  const float dpre1 = da1_s[lane] * gelu_erf_grad(pre1);
  o_ca1b[lane] = dpre1;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    o_ca1w[lane * L + l] = dpre1 * hn[l];
    dhn_s[lane][l] = dpre1 * p.ca1_w[lane * L + l];      // (re-read, L2-resident: holding the row across the kernel spilled registers)
  }
  __syncthreads();
  float dhn[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    float a = 0.f;
    for (int u = 0; u < 64; ++u) a += dhn_s[u][l];
    dhn[l] = a;
  }
  if (lane == 0) {
#pragma unroll
    for (int l = 0; l < L; ++l) { o_ca0g[l] = dhn[l] * (cls[l] - mean_a) * rstd_a; o_ca0b[l] = dhn[l]; }
  }
  ln_small_bwd<L>(cls, mean_a, rstd_a, p.ca0_g, dhn, dcls);
  // PCF balance gate
  __builtin_amdgcn_sched_barrier(0);                     // (keeps the second gate's operand loads from being hoisted over the first: spills at 128 registers)
  ln_small<L>(cls, p.gl0_g, p.gl0_b, gn, mean_g, rstd_g);
  dgw = wave_sum(dgw);
  const float dpre = dgw * gwv * (1.f - gwv);
  float dgn[L];
#pragma unroll
  for (int l = 0; l < L; ++l) dgn[l] = dpre * p.gl1_w[l];
  if (lane == 0) {
    o_gl1b[0] = dpre;
#pragma unroll
    for (int l = 0; l < L; ++l) { o_gl1w[l] = dpre * gn[l]; o_gl0g[l] = dgn[l] * (cls[l] - mean_g) * rstd_g; o_gl0b[l] = dgn[l]; }
  }
  ln_small_bwd<L>(cls, mean_g, rstd_g, p.gl0_g, dgn, dcls);
  if (lane == 0) {
#pragma unroll
    for (int l = 0; l < L; ++l) p.dcls[b * L + l] = dcls[l];
  }
}

// ---- token side: four lanes per latent row (global rows first, then local rows), 128 rows per workgroup.
// Gathers over the P prompts (staged in LDS), adds the proj_up gradient, applies QuickGELU'.
template <int L>
__global__ __launch_bounds__(512) void gpa_bwd_kernel(GpaArgs p, int ntok) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float dq_s[8][L], dpr_s[L];
  const int b = blockIdx.y, P = p.P, bx = blockIdx.x;
  if (bx > ntok) {                                           // one workgroup per prompt
    gpa_bwd_prompt_body<L>(p, b, bx - ntok - 1, dq_s, dpr_s);
    return;
  }
  if (bx == ntok) {
    // The gates' workgroup of the sample (one wave; gaviko.py:160-170) and, with it, the CLS row -- the only token row the gates' gradient
    // reaches, and one that attends to nothing (t = P < 2P + 2), so dz = (dcomb + dcls) o QuickGELU'(z) is all of it.
    if (threadIdx.x >= 64) return;
    float dcls[L];
    gpa_gates_bwd_body<L>(p, b, (int)threadIdx.x, dcls);
    const size_t row = (size_t)b * p.T + P;
    if (threadIdx.x == 0) {
#pragma unroll
      for (int l = 0; l < L; ++l) p.dzx[row * L + l] = (p.dcomb[row * L + l] + dcls[l]) * quick_gelu_grad(p.zx[row * L + l]);
    }
    return;
  }
  float* q_s = (float*)smem;              // [2][P][L]  scaled queries (global, local)
  float* dc_s = q_s + 2 * P * L;          // [2][P][L]  dctx
  float* ls_s = dc_s + 2 * P * L;         // [2][P]     lse
  float* de_s = ls_s + 2 * P;             // [2][P]     delta
  const float gwv = p.gw[b];
  for (int i = threadIdx.x; i < P * L; i += 512) {
    const size_t o = (size_t)b * P * L + i;
    const int pi = i / L, l = i - pi * L;
    const float df = p.dcomb[((size_t)b * p.T + pi) * L + l] * p.imp[b * P + pi];     // d fused = d enh * importance
    q_s[i] = p.qg[o]; q_s[P * L + i] = p.ql[o];
    dc_s[i] = gwv * df; dc_s[P * L + i] = (1.f - gwv) * df;
  }
  for (int i = threadIdx.x; i < P; i += 512) { ls_s[i] = p.lse_g[b * P + i]; ls_s[P + i] = p.lse_l[b * P + i]; }
  __syncthreads();
  if ((int)threadIdx.x < 2 * P) {                            // delta = dctx . ctx per (side, prompt)
    const int side = (int)threadIdx.x / P, pi = (int)threadIdx.x - side * P;
    const float* cv = (side == 0 ? p.cg : p.cl) + ((size_t)b * P + pi) * L;
    float d = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) d = __builtin_fmaf(dc_s[(side * P + pi) * L + l], cv[l], d);
    de_s[side * P + pi] = d;
  }
  __syncthreads();
  const int part = threadIdx.x & 3;
  const int r = bx * 128 + (threadIdx.x >> 2);
  if (r >= p.T + p.N) return;                              // (whole quads leave together)
  const bool is_local = r >= p.T;
  const int t = is_local ? r - p.T : r;
  if (!is_local && t <= P) return;                         // prompt rows: their own workgroups; the CLS row: the gates' workgroup
  const size_t row = is_local ? (size_t)b * p.N + t : (size_t)b * p.T + t;
  const float* lat = (is_local ? p.ll : p.xl) + row * L;
  float tok[L], g[L];
#pragma unroll
  for (int l = 0; l < L; ++l) { tok[l] = lat[l]; g[l] = 0.f; }
  const bool attends = is_local || t >= 2 * P + 2;
  if (attends) {
    const int side = is_local ? 1 : 0;
    const float* qs = q_s + side * P * L;
    const float* dcs = dc_s + side * P * L;
    for (int q = part; q < P; q += 4) {
      float d = 0.f, da = 0.f;
#pragma unroll
      for (int l = 0; l < L; ++l) { d += qs[q * L + l] * tok[l]; da += dcs[q * L + l] * tok[l]; }
      const float a = __expf(d - ls_s[side * P + q]);
      const float ds = a * (da - de_s[side * P + q]);
#pragma unroll
      for (int l = 0; l < L; ++l) g[l] += a * dcs[q * L + l] + ds * qs[q * L + l];
    }
  }
#pragma unroll
  for (int l = 0; l < L; ++l) {                            // sum of the quad's four partial gathers
    g[l] += __shfl_xor(g[l], 1, 64);
    g[l] += __shfl_xor(g[l], 2, 64);
  }
  if (part != 0) return;
  if (!is_local) {
#pragma unroll
    for (int l = 0; l < L; ++l) g[l] += p.dcomb[row * L + l];                    // image rows pass through proj_up
  }
  const float* z = (is_local ? p.zl : p.zx) + row * L;
  float* dz = (is_local ? p.dzl : p.dzx) + row * L;
#pragma unroll
  for (int l = 0; l < L; ++l) dz[l] = g[l] * quick_gelu_grad(z[l]);
}

static void fill_gpa(GpaArgs& a, const gvk_gpa_desc* d) {
  a.xl = d->xl; a.ll = d->ll;
  a.ca0_g = d->ca0_g; a.ca0_b = d->ca0_b; a.ca1_w = d->ca1_w; a.ca1_b = d->ca1_b; a.ca3_w = d->ca3_w; a.ca3_b = d->ca3_b;
  a.gl0_g = d->gl0_g; a.gl0_b = d->gl0_b; a.gl1_w = d->gl1_w; a.gl1_b = d->gl1_b;
  a.wgq = d->wgq; a.bgq = d->bgq; a.wlq = d->wlq; a.blq = d->blq;
  a.imp = d->imp; a.gw = d->gw; a.enh = d->enh; a.prm = d->prm; a.qg = d->qg; a.ql = d->ql; a.cg = d->cg; a.cl = d->cl;
  a.lse_g = d->lse_g; a.lse_l = d->lse_l; a.dcomb = d->dcomb; a.zx = d->zx; a.zl = d->zl; a.dimp = d->dimp; a.dgw_part = d->dgw_part;
  a.dqg = d->dqg; a.dql = d->dql; a.dcg = d->dcg; a.dcl = d->dcl; a.delta_g = d->delta_g; a.delta_l = d->delta_l; a.dprm = d->dprm;
  a.dcls = d->dcls; a.gate_partials = d->gate_partials; a.dzx = d->dzx; a.dzl = d->dzl;
  a.B = d->B; a.T = d->T; a.N = d->N; a.P = d->P; a.scale = d->scale;
  a.enh16 = (bf16*)d->enh16; a.ld16 = d->ld16; a.col16 = d->col16;
}

}  // namespace gvk

#define GVK_GPA_LAUNCH(KERNEL, grid, block, lds)                                                        \
  switch (d->L) {                                                                                       \
    case 4: GVK_LAUNCH((KERNEL<4>), grid, block, lds, s, a); break;                            \
    case 8: GVK_LAUNCH((KERNEL<8>), grid, block, lds, s, a); break;                            \
    case 16: GVK_LAUNCH((KERNEL<16>), grid, block, lds, s, a); break;                          \
    case 20: GVK_LAUNCH((KERNEL<20>), grid, block, lds, s, a); break;                          \
    case 32: GVK_LAUNCH((KERNEL<32>), grid, block, lds, s, a); break;                          \
    default: return set_error(-2, "gvk_gpa: L=%d unsupported (4, 8, 16, 20, 32)", d->L);               \
  }

static int gpa_check(const gvk_gpa_desc* d, const char* what) {
  using namespace gvk;
  GVK_REQUIRE(d && d->xl && d->ll, "%s: null latents", what);
  GVK_REQUIRE(d->B > 0 && d->P > 0 && d->P <= 64 && d->N > 0, "%s: need 0 < P <= 64 (P=%d)", what, d->P);
  GVK_REQUIRE(d->T - (2 * d->P + 2) > 0, "%s: T=%d leaves no global image tokens after the double slice (P=%d)", what, d->T, d->P);
  return 0;
}

extern "C" int gvk_gpa_fwd(const gvk_gpa_desc* d, void* stream) {
  using namespace gvk;
  int rc = gpa_check(d, "gvk_gpa_fwd");
  if (rc) return rc;
  GVK_REQUIRE(d->imp && d->gw && d->enh && d->prm && d->qg && d->ql && d->cg && d->cl && d->lse_g && d->lse_l, "gvk_gpa_fwd: null output");
  GpaArgs a{};
  fill_gpa(a, d);
  hipStream_t s = (hipStream_t)stream;
  GVK_REQUIRE(d->enh16 == nullptr || (d->ld16 >= d->col16 + 3 * d->L && d->col16 >= 0), "gvk_gpa_fwd: enh16 slot out of range");
  GVK_REQUIRE(d->P <= 64, "gvk_gpa_fwd: P=%d > 64", d->P);
  GVK_GPA_LAUNCH(gpa_fwd_kernel, dim3(d->P, d->B), dim3(576), 0);
  return check_launch("gpa_fwd");
}

extern "C" int gvk_gpa_bwd(const gvk_gpa_desc* d, void* stream) {
  using namespace gvk;
  int rc = gpa_check(d, "gvk_gpa_bwd");
  if (rc) return rc;
  GVK_REQUIRE(d->dcomb && d->zx && d->zl && d->dimp && d->dgw_part && d->dqg && d->dql && d->dcg && d->dcl && d->delta_g && d->delta_l &&
                  d->dprm && d->dcls && d->gate_partials && d->dzx && d->dzl,
              "gvk_gpa_bwd: null pointer");
  GpaArgs a{};
  fill_gpa(a, d);
  hipStream_t s = (hipStream_t)stream;
  const int ntok = (d->T + d->N + 127) / 128;
  const int lds = (4 * d->P * d->L + 4 * d->P) * 4;
  switch (d->L) {
    case 4: GVK_LAUNCH((gpa_bwd_kernel<4>), dim3(ntok + 1 + d->P, d->B), dim3(512), lds, s, a, ntok); break;
    case 8: GVK_LAUNCH((gpa_bwd_kernel<8>), dim3(ntok + 1 + d->P, d->B), dim3(512), lds, s, a, ntok); break;
    case 16: GVK_LAUNCH((gpa_bwd_kernel<16>), dim3(ntok + 1 + d->P, d->B), dim3(512), lds, s, a, ntok); break;
    case 20: GVK_LAUNCH((gpa_bwd_kernel<20>), dim3(ntok + 1 + d->P, d->B), dim3(512), lds, s, a, ntok); break;
    case 32: GVK_LAUNCH((gpa_bwd_kernel<32>), dim3(ntok + 1 + d->P, d->B), dim3(512), lds, s, a, ntok); break;
    default: return set_error(-2, "gvk_gpa_bwd: L=%d unsupported (4, 8, 16, 20, 32)", d->L);
  }
  return check_launch("gpa_bwd");
}

extern "C" int gvk_gpa_gate_param_count(int L, int P) { return 4 * L + 64 * L + 64 + 64 * P + P + L + 1; }
