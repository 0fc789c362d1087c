// Probe build of the attention forward: shader-clock stamps of one steady-state key tile and of the kernel's phases.  Not part of the library.
//   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -ffp-contract=off -shared tools/probe/probe_attn.hip \
//         -Lgaviko_amd -lgaviko_hip -Wl,-rpath,'$ORIGIN/../../gaviko_amd' -o tools/probe/libprobe_attn.so
#define GVK_STAMPS 1
#define gvk_attention_fwd_bf16_dropout probe_unused_fwd_dropout
#define gvk_attention_fwd_bf16 probe_unused_fwd
#define gvk_qkv_prescale_bf16 probe_unused_prescale
#include "../../gaviko_amd/csrc/attention_fwd.hip"

template <int KB, int VAR>
static int run(const void* qkv, void* out, float* lse, int B, int T, int H, void* stamps, void* stream) {
  using namespace gvk;
  const AttnDrop dr{0, (const unsigned long long*)stamps, 0u, 1.f};
  const int lds = 2 * 2 * KB * 128;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<KB, false, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((attn_fwd_kernel<KB, false, VAR>), dim3(((T + kQB - 1) / kQB) * H * B), dim3(256), lds, (hipStream_t)stream, (const bf16*)qkv, (bf16*)out, lse,
                     T, H, 3 * H * 64, H * 64, dr);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int probe_attn_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, int kb, int var, void* stamps, void* stream) {
  if (kb == 96) return var == 1 ? run<96, 1>(qkv, out, lse, B, T, H, stamps, stream) : run<96, 0>(qkv, out, lse, B, T, H, stamps, stream);
  return var == 1 ? run<128, 1>(qkv, out, lse, B, T, H, stamps, stream) : run<128, 0>(qkv, out, lse, B, T, H, stamps, stream);
}
