// Probe build of the four-wave attention forward: shader-clock stamps of one steady-state key tile.  Not part of the library.
//   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -ffp-contract=off -shared tools/probe/probe_attn.hip \
//         -Lgaviko_amd -lgaviko_hip -Wl,-rpath,'$ORIGIN/../../gaviko_amd' -o tools/probe/libprobe_attn.so
#define GVK_STAMPS 1
#ifndef PROBE_RS
#define PROBE_RS false
#endif
#define gvk_attention_fwd_bf16_dropout probe_unused_fwd_dropout
#define gvk_attention_fwd_bf16 probe_unused_fwd
#include "../../gaviko_amd/csrc/attention_fwd.hip"

extern "C" int probe_attn_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, void* stamps, void* stream) {
  using namespace gvk;
  const AttnDrop dr{0, (const unsigned long long*)stamps, 0u, 1.f};
  const int lds = 2 * 2 * kTileBytes;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<false, PROBE_RS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((attn_fwd_kernel<false, PROBE_RS>), dim3(((T + kQB - 1) / kQB) * H * B), dim3(256), lds, (hipStream_t)stream, (const bf16*)qkv, (bf16*)out, lse,
                     T, H, 3 * H * 64, H * 64, 0.125f * 1.44269504088896340736f, dr);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
