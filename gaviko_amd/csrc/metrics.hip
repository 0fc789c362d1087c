// Evaluation metrics of eval.py:103-122 on the device: softmax probabilities, argmax predictions, the K x K confusion matrix
// (accuracy and the quadratic-weighted Cohen kappa follow from it on the host, in float64, over K*K integers) and the
// one-vs-rest ROC AUC of every class as exact integer pair counts: AUC_c = (2 #{p_i > p_j} + #{p_i == p_j}) / (2 n_pos n_neg) over
// positives i and negatives j -- the Mann-Whitney form of the trapezoidal area sklearn.metrics.roc_auc_score integrates.
#include "common.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

__global__ __launch_bounds__(256) void eval_rows_kernel(const float* __restrict__ logits, const long long* __restrict__ target, float* __restrict__ proba,
                                                        int* __restrict__ pred, unsigned long long* __restrict__ confusion, int N, int K) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float* x = logits + (size_t)i * K;
  int am = 0;
  float m = x[0];
  for (int k = 1; k < K; ++k)
    if (x[k] > m) { m = x[k]; am = k; }                    // torch.argmax: first maximum
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += expf(x[k] - m);
  for (int k = 0; k < K; ++k) proba[(size_t)i * K + k] = expf(x[k] - m) / s;
  pred[i] = am;
  const long long t = target[i];
  if (t >= 0 && t < K) atomicAdd(&confusion[(size_t)t * K + am], 1ull);
}

// counts[c] = {2 * greater + ties, n_pos, n_neg}; one thread per (sample i, class c) with y_i == c, columns staged through LDS
__global__ __launch_bounds__(256) void ovr_auc_kernel(const float* __restrict__ proba, const long long* __restrict__ target,
                                                      unsigned long long* __restrict__ counts, int N, int K) {
  __shared__ float sc[1024];
  __shared__ int neg[1024];
  const int c = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const bool pos = i < N && target[i] == c;
  const float pi = i < N ? proba[(size_t)i * K + c] : 0.f;
  unsigned long long w2 = 0ull;
  for (int j0 = 0; j0 < N; j0 += 1024) {
    __syncthreads();
    for (int j = threadIdx.x; j < 1024; j += 256) {
      const int jj = j0 + j;
      sc[j] = jj < N ? proba[(size_t)jj * K + c] : 0.f;
      neg[j] = jj < N && target[jj] != c;
    }
    __syncthreads();
    if (pos) {
      const int lim = min(1024, N - j0);
      for (int j = 0; j < lim; ++j)
        if (neg[j]) w2 += pi > sc[j] ? 2ull : (pi == sc[j] ? 1ull : 0ull);
    }
  }
  if (pos) {
    atomicAdd(&counts[(size_t)c * 3], w2);
    atomicAdd(&counts[(size_t)c * 3 + 1], 1ull);
  } else if (i < N) {
    atomicAdd(&counts[(size_t)c * 3 + 2], 1ull);
  }
}

}  // namespace gvk

extern "C" int gvk_eval_rows(const float* logits, const void* target, float* proba, int32_t* pred, void* confusion, int N, int K, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(logits && target && proba && pred && confusion && N > 0 && K > 0, "gvk_eval_rows: bad arguments");
  GVK_LAUNCH(eval_rows_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, (const long long*)target, proba, (int*)pred,
             (unsigned long long*)confusion, N, K);
  return check_launch("eval_rows");
}

extern "C" int gvk_ovr_auc_counts(const float* proba, const void* target, void* counts, int N, int K, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(proba && target && counts && N > 0 && K > 0 && K <= 65535, "gvk_ovr_auc_counts: bad arguments");
  GVK_LAUNCH(ovr_auc_kernel, dim3((N + 255) / 256, K), dim3(256), 0, (hipStream_t)stream, proba, (const long long*)target, (unsigned long long*)counts, N, K);
  return check_launch("ovr_auc_counts");
}
