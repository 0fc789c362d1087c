/* gaviko_hip.h -- C-ABI of libgaviko_hip.so: the MI355X (gfx950) kernels behind GAViKO's 3D-ViT hot path.
 *
 * Boundary contract
 *   - plain C: device pointers + sizes + a hipStream_t passed as void*; no torch / C++ types.
 *   - every entry point enqueues on `stream` and returns immediately: 0 = ok, <0 = error
 *     (gvk_last_error() holds the message).  No allocation, no synchronisation, graph-capture safe.
 *   - "bf16" = raw uint16 storage of bfloat16.  Row-major everywhere.
 *   - activation matrices [M][ld] must be allocated with at least round_up(M,128) rows (the MFMA tiles read
 *     whole 128-row panels; rows >= M are computed but never stored).
 *
 * What each entry point replaces in the reference (implicit ATen dispatches):
 *   the reference has no native code -- citations are to the Python call sites, /root/reference/src/.
 */
#ifndef GAVIKO_HIP_H
#define GAVIKO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* gvk_last_error(void);
/* returns 950 when the code object loaded on the current device is gfx950, else <0 */
int gvk_device_check(void);
int gvk_abi_version(void);

/* ------------------------------------------------------------------ bf16 MFMA GEMM  Y = A . W^T (+ epilogue)
 * A [M][lda] bf16 (K contiguous), W [N][ldw] bf16 (K contiguous, i.e. nn.Linear.weight layout), fp32 accumulate.
 * Replaces every nn.Linear / Conv3d-as-GEMM on the path:
 *   model/vision_transformer.py:62 (to_qkv), :72 (to_out), :31,34 (MLP fc1/fc2), :150 (conv_proj) and their
 *   autograd dgrads (dX = dY . W, expressed as NT with the pre-transposed frozen weight). */
enum gvk_epilogue {
  GVK_EPI_STORE_BF16 = 0,     /* out0 bf16 [M][ldo] = acc (+bias)                                                */
  GVK_EPI_BIAS_RES_F32 = 1,   /* out0 f32  = acc + bias[n] + res[m][n]  (residual add; res may alias out0)        */
  GVK_EPI_BIAS_GELU_BF16 = 2, /* out0 bf16 = acc + bias (pre-activation, may be NULL); out1 bf16 = erf-GELU(out0) */
  GVK_EPI_PATCH_F32 = 3,      /* out0 f32 [(m/rows_in)*rows_out + row_off + m%rows_in][n] = acc + bias + pos[m%rows_in][n];
                                 out1 f32 [m][n] (may be NULL) = same value (GAViKO local stream)                 */
  GVK_EPI_GELU_BWD_BF16 = 4,  /* out0 bf16 = acc * GELU'(aux bf16 [m][n])   (fc2 dgrad fused with GELU backward)   */
  GVK_EPI_STORE_F32 = 5,      /* out0 f32  = acc (+bias)                                                          */
  GVK_EPI_BIAS_RES_F32_BF16 = 6 /* as 1, and out1 bf16 [M][ldo] = same value rounded                              */
};

typedef struct gvk_gemm_desc {
  const void* a;      /* bf16 [>=round_up(M,128)][lda] */
  const void* w;      /* bf16 [N][ldw] */
  void* out0;
  void* out1;
  const float* bias;  /* [N] or NULL */
  const float* res;   /* f32 [M][ldres] or NULL */
  const void* aux;    /* bf16 [M][ldaux] or NULL */
  const float* pos;   /* f32 [rows_in][N] or NULL */
  int32_t M, N, K;
  int32_t lda, ldw, ldo, ldres, ldaux;
  int32_t epilogue;
  int32_t rows_in, rows_out, row_off; /* GVK_EPI_PATCH_F32 only */
  int32_t tile;       /* 0 = auto, else BM*1000+BN (128128, 128064, 64064, 64128) */
} gvk_gemm_desc;
int gvk_gemm_nt_bf16(const gvk_gemm_desc* d, void* stream);

/* ------------------------------------------------------------------ casts / layout
 * fp32 -> bf16 copy of a [rows][cols] matrix (weights -> MFMA operand form), optionally transposed
 * (out [cols][rows]) for the dgrad operand of frozen weights. */
int gvk_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream);
int gvk_transpose_cast_f32_bf16(const float* in, void* out, int rows, int cols, void* stream);
/* im2col of non-overlapping 3-D patches, fp32 volume -> bf16 rows [B*n_patches][pd*ph*pw]
 * (K order (kd,kh,kw) == Conv3d weight.flatten(1); vision_transformer.py:126-128,150-151). */
int gvk_patchify_bf16(const float* img, void* out, int B, int D, int H, int W, int pd, int ph, int pw, void* stream);

/* ------------------------------------------------------------------ LayerNorm (eps 1e-5; vision_transformer.py:30,49,77)
 * fwd: x f32 [M][C] -> y bf16 [M][C] (MFMA operand) and/or y32 f32; saves mean/rstd f32 [M] (either may be NULL). */
int gvk_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                      float* mean, float* rstd, int M, int C, float eps, void* stream);
/* bwd (input gradient only -- frozen affine): dx = dres + LN'(dy); dres may be NULL; dx_bf16 (optional) = bf16 copy. */
int gvk_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                      const float* dres, float* dx, void* dx_bf16, int M, int C, void* stream);
/* affine gradients of a trainable LayerNorm: dgamma[c] += sum_m dy*xhat, dbeta[c] += sum_m dy (accumulate=0 overwrites).
 * scratch: f32 [2*64*C]. */
int gvk_layernorm_bwd_affine(const float* dy, const float* x, const float* mean, const float* rstd, float* dgamma,
                             float* dbeta, float* scratch, int M, int C, int accumulate, void* stream);

/* ------------------------------------------------------------------ multi-head self-attention, head dim 64
 * qkv bf16 [B*T (padded)][ld_qkv]: columns [q | k | v], each (head, 64) -- the to_qkv output as is
 * (vision_transformer.py:62-63).  out bf16 [B*T][ld_out] in 'b n (h d)' order (vision_transformer.py:71), lse f32 [B][H][T]
 * = log sum_j exp(scale * q.k_j) (natural log), saved for the backward.  scale = dim_head^-0.5 (vision_transformer.py:47,65). */
int gvk_attention_fwd_bf16(const void* qkv, void* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif
