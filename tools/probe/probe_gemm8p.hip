// Probe build of the eight-phase GEMM: in-kernel stamps (GVK_STAMPS) and main-loop ablations (GVK_ABLATE).  Not part of the library.
//   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -ffp-contract=off -DGVK_ABLATE=<bits> -shared \
//         tools/probe/probe_gemm8p.hip -Lgaviko_amd -lgaviko_hip -Wl,-rpath,'$ORIGIN/../../gaviko_amd' -o tools/probe/libprobe_gemm8p_<bits>.so
// Driver: tools/probe/probe_gemm8p.py (GPU box).
#define GVK_STAMPS 1
#include "../../gaviko_amd/csrc/gemm8p_bf16.hip"

// STORE_BF16 epilogue; `stamps` = uint64 [workgroups][8][32] or NULL
extern "C" int probe_gemm8p(const void* a, const void* w, void* out, int M, int N, int K, int variant, void* stamps, void* stream) {
  gvk::GemmArgs g{};
  g.A = (const gvk::bf16*)a; g.W = (const gvk::bf16*)w; g.out0 = out;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldo = N;
  g.aux = (const gvk::bf16*)stamps;
  return gvk::launch_gemm8p(g, GVK_EPI_STORE_BF16, variant, (hipStream_t)stream);
}
