/* Diagnostics and experiment kernels of libgaviko_hip_diag.so (`python -m gaviko_amd.build --diag`: the product sources compiled with
 * -DGVK_DIAG plus gemm_k2_bf16.hip, gemm_k4_bf16.hip, patch_gemm.hip).  NOT part of the product ABI: the product library
 * (libgaviko_hip.so, include/gaviko_hip.h) exports none of these, ignores every GAVIKO_HIP_* A/B switch named in the kernel sources
 * (gvk::diag_env) and knows no tile codes 9128128 / 4128128 (eight waves splitting every k-tile) / 5128128 (stream-K: gemm_sk_bf16.hip).  tools/ loads the diag library when GAVIKO_HIP_DIAG=1. */
#ifndef GAVIKO_HIP_DIAG_H
#define GAVIKO_HIP_DIAG_H
#include "gaviko_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* launches recorded on `stream` from now on are replaced by an empty kernel (contention studies; results are garbage); _clear undoes it */
int gvk_plan_nop_stream(void* stream);
int gvk_plan_nop_clear(void);

/* The same Conv3d (kernel = stride = patch; vision_transformer.py:126-128,150-157) as ONE implicit GEMM: the A operand is gathered from the
 * fp32 volume inside the kernel (no im2col matrix), w bf16 [C][pd*ph*pw] = Conv3d.weight.flatten(1), epilogue = GVK_EPI_PATCH_F32:
 * out0 f32 [B*rows_out][C] rows b*rows_out + row_off + t = patch(b, t) . w^T + bias + pos[t] (pos f32 [n_tokens][C]); out1 (optional)
 * f32 [B*n_tokens][C] receives the same rows densely (GAViKO's local stream, gaviko.py:532-548).  pw = 16, ph*pw % 64 == 0, C % 128 == 0. */
int gvk_patch_embed_bf16(const float* img, const void* w, const float* bias, const float* pos, float* out0, float* out1, int B, int D, int H,
                         int W, int pd, int ph, int pw, int C, int rows_out, int row_off, void* stream);
/* built, bit-identical to gvk_patchify_bf16 + gvk_gemm_nt_bf16(PATCH_F32), 73-79 us against 54 us for the pair */

#ifdef __cplusplus
}
#endif
#endif
