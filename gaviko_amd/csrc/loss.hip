// Loss seed of the training step, forward and backward in ONE launch: train.py:176-179 (FocalLoss(gamma=1.2) or
// CrossEntropyLoss), 283/306 (loss = criterion(outputs, labels)) and the two per-step host reads of train.py:327-328
// (running loss, number of correct argmax predictions), which become device-side accumulators.
//
// FocalLoss is reproduced as it EXECUTES (losses/focal_loss.py:84-115): the live _process_preds clamps to [eps, 1-eps] and
// then softmaxes, and forward calls it twice, so
//   p1 = softmax(clamp(x)),  p2 = softmax(clamp(p1)),  pt = p2[target],  l = w_t (1-pt)^gamma * -log(eps + pt)
//   loss = sum l / sum (not ignored) w_t
// and the gradient flows back through both softmaxes and both clamps (zero for every logit outside [eps, 1-eps]).
// The logits are [B, K] with K a handful of classes: one workgroup, one thread per sample, every vector recomputed per pass
// instead of kept (no arrays, no scratch).
#include "common.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

struct LossArgs {
  const float* logits; const long long* target; const float* weights;
  float* loss; float* dlogits; float* meter;
  int B, K, kind, reduction;
  float gamma, eps, hi;
  long long ignore_index;
};

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

struct Soft1 { float m, s; };
// softmax statistics of clamp(x)
__device__ __forceinline__ Soft1 soft1(const float* x, int K, float lo, float hi) {
  float m = -INFINITY;
  for (int k = 0; k < K; ++k) m = fmaxf(m, clampf(x[k], lo, hi));
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += expf(clampf(x[k], lo, hi) - m);
  return {m, s};
}
__device__ __forceinline__ float p1_of(const float* x, int k, Soft1 a, float lo, float hi) { return expf(clampf(x[k], lo, hi) - a.m) / a.s; }

__global__ __launch_bounds__(256) void loss_kernel(LossArgs p) {
  __shared__ float red[3][256];
  const int tid = threadIdx.x, K = p.K;
  const float lo = p.eps, hi = p.hi;
  float lsum = 0.f, wsum = 0.f, correct = 0.f;
  for (int b = tid; b < p.B; b += 256) {
    const float* x = p.logits + (size_t)b * K;
    float* dx = p.dlogits + (size_t)b * K;
    const long long tg = p.target[b];
    const bool ign = tg == p.ignore_index;
    const int t = ign ? 0 : (int)tg;
    const float w = p.weights != nullptr ? p.weights[t] : 1.f;
    int am = 0;                                           // torch.argmax: first maximum
    for (int k = 1; k < K; ++k)
      if (x[k] > x[am]) am = k;
    if (!ign && am == t) correct += 1.f;
    if (ign) {
      for (int k = 0; k < K; ++k) dx[k] = 0.f;
      if (p.reduction == 2) p.loss[b] = 0.f;             // 'none': an ignored row's loss is masked to zero (focal_loss.py:108)
      continue;
    }
    wsum += w;
    if (p.kind == 0) {                                    // cross entropy: -log softmax(x)[t]
      float m = -INFINITY;
      for (int k = 0; k < K; ++k) m = fmaxf(m, x[k]);
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += expf(x[k] - m);
      lsum += w * (logf(s) + m - x[t]);
      if (p.reduction == 2) p.loss[b] = w * (logf(s) + m - x[t]);
      for (int k = 0; k < K; ++k) dx[k] = w * (expf(x[k] - m) / s - (k == t ? 1.f : 0.f));
      continue;
    }
    const Soft1 a = soft1(x, K, lo, hi);
    float m2 = -INFINITY;
    for (int k = 0; k < K; ++k) m2 = fmaxf(m2, clampf(p1_of(x, k, a, lo, hi), lo, hi));
    float s2 = 0.f;
    for (int k = 0; k < K; ++k) s2 += expf(clampf(p1_of(x, k, a, lo, hi), lo, hi) - m2);
    const float pt = expf(clampf(p1_of(x, t, a, lo, hi), lo, hi) - m2) / s2;
    const float om = 1.f - pt, nll = -logf(p.eps + pt);
    const float fg = powf(om, p.gamma);
    lsum += w * fg * nll;
    if (p.reduction == 2) p.loss[b] = w * fg * nll;       // 'none' (focal_loss.py:118): the per-sample vector, dlogits = its rows' own gradients
    // dl/dpt = -gamma (1-pt)^(gamma-1) nll - (1-pt)^gamma / (eps + pt)
    const float dpt = w * (-p.gamma * powf(om, p.gamma - 1.f) * nll - fg / (p.eps + pt));
    // through softmax 2 and clamp 2: g1_j = dpt * pt * (delta_tj - p2_j) * [lo <= p1_j <= hi];  dot = sum_j g1_j p1_j
    float dot = 0.f;
    for (int j = 0; j < K; ++j) {
      const float p1 = p1_of(x, j, a, lo, hi);
      const float p2 = expf(clampf(p1, lo, hi) - m2) / s2;
      const float g1 = (p1 >= lo && p1 <= hi) ? dpt * pt * ((j == t ? 1.f : 0.f) - p2) : 0.f;
      dot += g1 * p1;
    }
    for (int k = 0; k < K; ++k) {
      const float p1 = p1_of(x, k, a, lo, hi);
      const float p2 = expf(clampf(p1, lo, hi) - m2) / s2;
      const float g1 = (p1 >= lo && p1 <= hi) ? dpt * pt * ((k == t ? 1.f : 0.f) - p2) : 0.f;
      dx[k] = (x[k] >= lo && x[k] <= hi) ? p1 * (g1 - dot) : 0.f;
    }
  }
  red[0][tid] = lsum; red[1][tid] = wsum; red[2][tid] = correct;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; red[2][tid] += red[2][tid + s];
    }
    __syncthreads();
  }
  const float den = p.reduction == 0 ? red[1][0] : 1.f;   // 'mean': sum over not-ignored samples of their weight
  const float loss = red[0][0] / den;
  for (int b = tid; b < p.B; b += 256) {
    float* dx = p.dlogits + (size_t)b * K;
    for (int k = 0; k < K; ++k) dx[k] /= den;
  }
  if (tid == 0) {
    if (p.reduction != 2) p.loss[0] = loss;
    if (p.meter != nullptr) {                              // train.py:327-328: running_loss += loss * B; num_acc += correct
      p.meter[0] += loss * (float)p.B;
      p.meter[1] += red[2][0];
      p.meter[2] += (float)p.B;
    }
  }
}

}  // namespace gvk

extern "C" int gvk_loss_fwd_bwd(const gvk_loss_desc* d, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(d && d->logits && d->target && d->loss && d->dlogits, "gvk_loss_fwd_bwd: null pointer");
  GVK_REQUIRE(d->B > 0 && d->K > 1 && d->K <= 4096, "gvk_loss_fwd_bwd: B=%d K=%d (K = 1, the sigmoid branch, is not built)", d->B, d->K);
  GVK_REQUIRE(d->kind == GVK_LOSS_CE || d->kind == GVK_LOSS_FOCAL, "gvk_loss_fwd_bwd: unknown kind %d", d->kind);
  GVK_REQUIRE(d->reduction >= 0 && d->reduction <= 2, "gvk_loss_fwd_bwd: reduction must be 0 (mean), 1 (sum) or 2 (none: loss receives B values)");
  LossArgs a{d->logits, (const long long*)d->target, d->weights, d->loss, d->dlogits, d->meter, d->B, d->K, d->kind, d->reduction,
             d->gamma, d->eps, 1.f - d->eps, (long long)d->ignore_index};
  GVK_LAUNCH(loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("loss_fwd_bwd");
}
