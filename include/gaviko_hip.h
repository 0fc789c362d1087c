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
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* gvk_last_error(void);
/* returns 950 when the code object loaded on the current device is gfx950, else <0 */
int gvk_device_check(void);
int gvk_abi_version(void);   /* 12 */

/* ------------------------------------------------------------------ launch plans
 * The reference drives its step from the Python interpreter (train.py:296-319: one autograd node per op).  Here one
 * training step is ~670 launches on three streams; a plan records them once (kernel, grid, by-value arguments, stream,
 * and the event record/wait edges between streams) and gvk_plan_replay re-issues them from a single C loop.
 * Recording is per host thread: between gvk_plan_begin and gvk_plan_end every gvk_* launch made by that thread both
 * executes and is appended to the plan.  Pointers are baked in: replay is valid while the buffers passed at record
 * time stay allocated (the engine's static workspace).  Events used for cross-stream edges inside a plan must come
 * from gvk_plan_event_record.  gvk_plan_end returns the plan id (>= 0).                                              */
int gvk_plan_begin(void);
int gvk_plan_end(void);
int gvk_plan_abort(void);
int gvk_plan_size(int plan);                       /* number of recorded nodes */
int gvk_plan_replay(int plan);
int gvk_plan_free(int plan);
int gvk_plan_event_record(void* stream);           /* -> event id within the plan being recorded */
int gvk_plan_event_wait(void* stream, int event);
/* gvk_plan_event_record with the system-scope fence kept (events that peer devices' reads are ordered behind: all-reduce buckets) */
int gvk_plan_event_record_fenced(void* stream);
/* issued immediately (not recorded): `stream` waits for event `event` of plan `plan` as recorded by its most recent replay */
int gvk_plan_event_stream_wait(int plan, int event, void* stream);
/* measurement: milliseconds between two events of a replayed plan.  Events carry timestamps only in plans recorded after
 * gvk_plan_set_timing(1) (or with GAVIKO_HIP_PLAN_TIMING set in the environment); bench.py brackets the GEMM launches
 * of an instrumented copy of the step this way, so the kernels are timed inside the real three-stream schedule. */
int gvk_plan_set_timing(int on);
int gvk_plan_event_elapsed(int plan, int e0, int e1, float* ms);
/* small stream-ordered utilities the step needs between kernels (recorded into a plan like any launch) */
int gvk_memset_async(void* ptr, int value, size_t bytes, void* stream);
int gvk_seed_advance(void* seed_u64, uint64_t inc, void* stream);   /* device-side dropout epoch += inc */
int gvk_scale_f32(float* x, float alpha, long n, void* stream);

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
  GVK_EPI_BIAS_RES_F32_BF16 = 6, /* as 1, and out1 bf16 [M][ldo] = same value rounded                             */
  GVK_EPI_BIAS_RELU_BF16 = 7, /* out0 bf16 = max(acc + bias, 0)          (AdaptFormer down-projection, adaptformer.py:63-64)   */
  GVK_EPI_RELU_BWD_BF16 = 8   /* out0 bf16 = acc * (aux bf16 [m][n] > 0) (dgrad through that ReLU)                              */
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
  const void* seed_ptr; /* device uint64 dropout epoch (drop_p > 0) */
  int32_t M, N, K;
  int32_t lda, ldw, ldo, ldres, ldaux;
  int32_t epilogue;
  int32_t rows_in, rows_out, row_off; /* GVK_EPI_PATCH_F32 only */
  int32_t tile;       /* 0 = auto, else BM*1000+BN (128128, 128064, 64064, 64128); 3128128 / 3064128 = 128x128 / 64x128 with three LDS stages; 3096128 = 96x128 with three stages;;
                         256256 = eight waves on a 256x256 tile (STORE_BF16, BIAS_GELU_BF16, GELU_BWD_BF16 only);
                         8256256 / 7256256 = the eight-phase 256x256 kernel (gemm8p_bf16.hip: two wave groups one barrier apart, counted
                         vmcnt), LDS-DMA issued in the load sections / inside the MFMA clusters; epilogues 0, 1, 2, 4, 5, N % 256 == 0, K >= 128 */
  float drop_p;       /* nn.Dropout behind the Linear (vision_transformer.py:32-34,54), epilogues 1, 2 (on out1), 4: the value at
                         (m, n) is multiplied by mask(seed + *seed_ptr, m*N + n) / (1 - drop_p); 0 = off */
  uint64_t seed;
  int32_t scale_cols; /* GVK_EPI_STORE_BF16 (bf16 entry point) only: columns n < scale_cols (a multiple of 8) are multiplied by col_scale in fp32 */
  float col_scale;    /* before the ONE rounding to bf16 -- the q block of a qkv projection leaves the GEMM as q * scale * log2(e), the form
                         the attention kernels take (gvk_attention_*_bf16); 0 columns = off */
  /* LayerNorm folded into its consumer (GVK_EPI_STORE_BF16, bf16 entry point): a = the RAW rows x as bf16, w = gamma o W; with ln_mean /
     ln_rstd f32 [M] and ln_c1 f32 [N] = row sums of w (of its bf16 values) the epilogue stores rstd[m]*(acc - mean[m]*c1[n]) + bias[n],
     bias[n] = sum_c beta[c] W[n][c] -- LayerNorm(x) . W^T (vision_transformer.py:49,61-62) without the LayerNorm launch.  NULL = off */
  const float* ln_mean; const float* ln_rstd; const float* ln_c1;
  /* GVK_EPI_BIAS_RES_F32_BF16: also write per-row partial (sum, sum of squares) of the fp32 output, f32 [N / 64][M][2] (64-column groups
     of the 128-wide tiles); gvk_prompt_up_fix_stats turns them into the mean / rstd the folded consumer reads.  NULL = off.
     The partials are taken of (x - stat_pivot[m]), stat_pivot f32 [M] = any per-row estimate of the row mean (the engine passes the
     mean of the residual row, which its LayerNorm has just computed): shifted sums do not cancel when |mean| >> std, which the plain
     E[x^2] - mean^2 form does.  stat_pivot NULL = pivot 0 */
  float* stat_part;
  const float* stat_pivot;
  /* Row PANELS at a stride (bf16 entry point; 0 = off): the launch covers m_panels row tiles whose first rows are 0, m_stride, 2 m_stride, ...
     (the first rows of every sample of a [B][T][.] token tensor: m_stride = T) instead of the contiguous rows 0 .. M-1; M stays the
     number of rows the operands have (rows >= M are not stored).  Only rows of the panels are read and written.  What it is for: a
     product whose other rows are never consumed -- the last layer's MLP when the head pools the first rows of every sample
     (gaviko.py:316), the first layer's qkv dgrad when only the prompt rows of the input carry a trainable tensor.  Tile: one of the
     4-wave kernels (tile = 0 picks 64 x 128 with three stages), m_stride >= the tile's rows, (m_panels - 1) * m_stride + tile rows <= the padded M */
  int32_t m_panels, m_stride;
  /* with m_panels: cut every tile's K loop into pieces over otherwise idle CUs (a few 64-row tiles run at the latency of one workgroup's
     loop); the pieces are summed in a fixed order by the workgroup that arrives last (no float atomics, bitwise reproducible).
     splitk_ws: 256-byte aligned device memory, >= 1 KiB + tiles * 8 * 32 KiB, its first KiB ZERO at allocation (ticket words, left zero);
     launches that share it must be ordered by their stream.  NULL = one workgroup per tile */
  void* splitk_ws; uint64_t splitk_ws_bytes;
  /* pieces per tile (2..8) when splitk_ws is given; 0 = as many as keep the launch within one round of the chip.  Also: tile = 2128128 runs a
     FULL launch (no panels) of <= 256 tiles of 128 x 128 this way -- two pieces of a K >= 2304 loop on two workgroups that share a CU
     (measured: no faster than one three-stage workgroup per tile; the engine does not use it) */
  int32_t ksplit;
  /* (bf16 entry point, no dropout) the GELU derivative is computed ONCE, where its exponential is computed anyway: with aux_is_grad = 1
     GVK_EPI_BIAS_GELU_BF16 stores out0 = bf16 GELU'(acc + bias) instead of the pre-activation, and GVK_EPI_GELU_BWD_BF16 reads aux as that
     derivative (out0 = acc * aux).  A caller sets it on both GEMMs of an MLP (vision_transformer.py:31-34 and its autograd) or on neither */
  int32_t aux_is_grad;
} gvk_gemm_desc;
int gvk_gemm_nt_bf16(const gvk_gemm_desc* d, void* stream);
/* number of 64-column groups gvk_gemm_desc.stat_part is indexed by for an N-column output */
int gvk_gemm_stat_parts(int N);

/* ------------------------------------------------------------------ fp32 compute path
 * BASELINE cfg4 (adaptformer / melo) runs the reference in fp32 with a 1e-5 tolerance, which bf16 MFMA operands cannot
 * meet.  These entry points mirror their bf16 namesakes argument for argument -- same descriptor, same epilogue table,
 * same layouts -- with every 16-bit slot (A, W, bf16 outputs, aux) carrying float.  GELU is the exact erf form.
 * gvk_gemm_nt_f32: v_mfma_f32_16x16x4_f32 (exact fp32 products), N % 64 == 0, K % 16 == 0.
 * gvk_attention_*_f32: flash-style fp32 VALU kernels; delta f32 [B][H][T] is scratch written by the backward. */
int gvk_gemm_nt_f32(const gvk_gemm_desc* d, void* stream);
int gvk_attention_fwd_f32(const float* qkv, float* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale,
                          void* stream);
int gvk_attention_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv,
                          int B, int T, int H, int ld_qkv, int ld_out, float scale, void* stream);
/* fp32 counterparts of gvk_attention_fwd/bwd_bf16_dropout (same mask function, so the two precisions drop the same elements) */
int gvk_attention_fwd_f32_dropout(const float* qkv, float* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale,
                                  float drop_p, uint64_t seed, const void* seed_ptr, void* stream);
int gvk_attention_bwd_f32_dropout(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv,
                                  int B, int T, int H, int ld_qkv, int ld_out, float scale, float drop_p, uint64_t seed, const void* seed_ptr,
                                  void* stream);
int gvk_patchify_f32(const float* img, float* out, int B, int D, int H, int W, int pd, int ph, int pw, void* stream);
int gvk_transpose_f32(const float* in, float* out, int rows, int cols, void* stream);
int gvk_transpose_bf16(const void* in, void* out, int rows, int cols, void* stream);   /* operand transposes of the unfrozen-backbone wgrad GEMMs */
int gvk_copy_f32_strided(const float* in, float* out, int M, int C, int ld_in, void* stream);   /* out[M][C] = in[M][ld_in] cols 0..C */
/* stream-ordered device-to-device copy (recorded into a launch plan like any launch) */
int gvk_copy_async(void* dst, const void* src, size_t bytes, void* stream);

/* ------------------------------------------------------------------ casts / layout
 * fp32 -> bf16 copy of a [rows][cols] matrix (weights -> MFMA operand form), optionally transposed
 * (out [cols][rows]) for the dgrad operand of frozen weights. */
int gvk_cast_f32_bf16(const float* in, void* out, int64_t n, void* stream);
int gvk_transpose_cast_f32_bf16(const float* in, void* out, int rows, int cols, void* stream);
/* Split-bf16 packing of a narrow fp32 operand into 3 ca + 2 spare K columns (from col0) of a bf16 GEMM operand dst [rows][ld_dst]:
 * x = hi + lo (hi = bf16(x), lo = bf16(x - hi)); activation side (weight_side 0) writes [a_hi | a_lo | a_hi], weight side writes
 * [w_hi | w_hi | w_lo | b_hi | b_lo] (b f32 [rows] or NULL; it meets two constant-1 columns of the activation operand).  The GPA
 * up-projection (gaviko.py:187, x + proj_up(.)) then rides the MLP's second Linear (vision_transformer.py:34) as K-concatenation,
 * A' = [act | lat_hi | lat_lo | lat_hi | 1 | 1], W' = [W_fc2 | Wup_hi | Wup_hi | Wup_lo | b_hi | b_lo], at ~2^-16 relative accuracy. */
int gvk_pack_split_bf16(const float* a, int ca, const float* b, void* dst, int ld_dst, int col0, int rows, int weight_side, void* stream);
/* im2col of non-overlapping 3-D patches, fp32 volume -> bf16 rows [B*n_patches][pd*ph*pw]
 * (K order (kd,kh,kw) == Conv3d weight.flatten(1); vision_transformer.py:126-128,150-151). */
int gvk_patchify_bf16(const float* img, void* out, int B, int D, int H, int W, int pd, int ph, int pw, void* stream);


/* ------------------------------------------------------------------ LayerNorm (eps 1e-5; vision_transformer.py:30,49,77)
 * fwd: x f32 [M][C] -> y bf16 [M][C] (MFMA operand) and/or y32 f32; saves mean/rstd f32 [M] (either may be NULL). */
int gvk_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                      float* mean, float* rstd, int M, int C, float eps, void* stream);
/* gvk_layernorm_fwd (bf16 output) with the GPA prompt fix of the previous layer applied on the way in: rows with (m % T) < P first receive
 * x[m] += (enh[m / T][m % T] - lat[m]) . wup^T (wup f32 [C][L]; gaviko.py:183-187 when the plain-latent part rode the MLP GEMM) and are
 * written back; then every row is normalised.  Replaces gvk_prompt_up_fix + gvk_layernorm_fwd at a layer boundary. */
int gvk_layernorm_fwd_fix(float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd, int M, int C, float eps,
                          const float* enh, const float* lat, const float* wup, int T, int P, int L, void* stream);
/* bwd (input gradient only -- frozen affine): dx = dres + LN'(dy); dres may be NULL; dx_bf16 (optional) = bf16 copy. */
int gvk_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                      const float* dres, float* dx, void* dx_bf16, int M, int C, void* stream);
/* the same for a row SUBSET: rows g * group_stride + r, r < rows_per_group, g < groups, of every operand (the first rows of every sample);
 * the other rows are neither read nor written */
int gvk_layernorm_bwd_rows(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* dres, float* dx, void* dx_bf16, int groups, int rows_per_group, int group_stride, int C, void* stream);
/* affine gradients of a trainable LayerNorm: dgamma[c] += sum_m dy*xhat, dbeta[c] += sum_m dy (accumulate=0 overwrites).
 * scratch: f32 [2*64*C]. */
int gvk_layernorm_bwd_affine(const float* dy, const float* x, const float* mean, const float* rstd, float* dgamma,
                             float* dbeta, float* scratch, int M, int C, int accumulate, void* stream);

/* LayerNorm with a fused rank-L projection of the rows it already holds (L = 20 = configs/gaviko.yaml prompt_latent_dim):
 *   fwd_proj:  y_bf16 = LN(x) as gvk_layernorm_fwd, and   proj->y = act(x . W^T + bias)        of the RAW input rows
 *              (gaviko.py:155-156: GPA proj_down + QuickGELU of the post-attention stream; proj->z = pre-activation);
 *   bwd_proj:  dx as gvk_layernorm_bwd, and               proj->y = dx . W                      of the OUTPUT rows
 *              (autograd of gaviko.py:187 proj_up: the next-lower layer's dcomb = dG . W_up).
 * w_layout 0: w [L][C]; 1: w [C][L].  Covers L in {4, 8, 16, 20} and 128 <= C <= 1024; otherwise use gvk_skinny_down. */
typedef struct gvk_rowproj_desc {
  const float* w; const float* bias;       /* bias [L] or NULL */
  float* y; float* z;                      /* y [M][L]; z [M][L] pre-activation or NULL */
  void* y_split;                           /* optional: split-bf16 copy [hi | lo | hi] of y (gvk_pack_split_bf16, activation side) into columns
                                              col_split .. col_split + 3L - 1 of a bf16 [M][ld_split] GEMM operand */
  int32_t L, w_layout, act;                /* act: 0 none, 1 QuickGELU */
  int32_t ld_split, col_split;
} gvk_rowproj_desc;
int gvk_layernorm_fwd_proj(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd,
                           int M, int C, float eps, const gvk_rowproj_desc* proj, void* stream);
int gvk_layernorm_bwd_proj(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* dres, float* dx, void* dx_bf16, int M, int C, const gvk_rowproj_desc* proj, void* stream);
/* gvk_layernorm_bwd / _rows / _bwd_proj with the output gradient dy in bf16 [M][C] -- the form a dgrad GEMM with the STORE_BF16 epilogue
 * leaves it in (fc1 / qkv dgrad of a frozen backbone: half the bytes written by the GEMM and read here).  rows_per_group = 0: all M rows;
 * otherwise rows g * group_stride + r, r < rows_per_group, g < groups (then proj must be NULL).  proj != NULL: as gvk_layernorm_bwd_proj. */
typedef struct gvk_ln_bwd_dy16_desc {
  const void* dy_bf16; const float* x; const float* mean; const float* rstd; const float* gamma; const float* dres;
  float* dx; void* dx_bf16;
  const gvk_rowproj_desc* proj;
  int32_t M, C, groups, rows_per_group, group_stride;
} gvk_ln_bwd_dy16_desc;
int gvk_layernorm_bwd_dy16(const gvk_ln_bwd_dy16_desc* d, void* stream);
/* LayerNorm backward fused with a rank-L update of the same rows:  dx = dres + LN'(dy; x, mean, rstd, gamma) + lat . W^T  (+ bf16 copy).
 * Replaces gvk_layernorm_bwd followed by gvk_skinny_up(accumulate) on the backbone stream: the autograd of gaviko.py:304 (the MLP block's
 * LayerNorm) and of gaviko.py:155 (GPA's proj_down of the global tokens: dG1 += dzx . W_d) in one pass.  lat f32 [M][L]; w_layout 0: W [C][L],
 * 1: W [L][C].  Covers L = 20, C in {192, 768, 1024}. */
int gvk_layernorm_bwd_up(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                         float* dx, void* dx_bf16, const float* lat, const float* w, int w_layout, int M, int C, int L, void* stream);

/* ------------------------------------------------------------------ multi-head self-attention, head dim 64
 * qkv bf16 [B*T (padded)][ld_qkv]: columns [q' | k | v], each (head, 64) -- the to_qkv output (vision_transformer.py:62-63) with the q
 * block PRE-SCALED: q' = q * scale * log2(e), multiplied in fp32 before the one rounding to bf16.  gvk_gemm_nt_bf16 does it in the
 * projection's epilogue (gvk_gemm_desc.scale_cols = H*64, col_scale = scale * log2(e)); gvk_qkv_prescale_bf16 converts a raw qkv buffer
 * in place.  The forward and both backward passes then read bit-identical score operands (P of the backward is recomputed against the
 * forward's lse) and no kernel spends a multiply per score.  out bf16 [B*T][ld_out] in 'b n (h d)' order (vision_transformer.py:71), lse f32 [B][H][T]
 * = log sum_j exp(scale * q.k_j) (natural log), saved for the backward.  scale = dim_head^-0.5 (vision_transformer.py:47,65). */
int gvk_attention_fwd_bf16(const void* qkv, void* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale,
                           void* stream);
/* q block of a raw to_qkv output -> q * scale * log2(e), in place (rows = B*T) */
int gvk_qkv_prescale_bf16(void* qkv, int rows, int H, int ld_qkv, float scale, void* stream);

/* backward: dqkv bf16 [B*T][ld_qkv] = [dq | dk | dv] in the qkv layout -- gradients of the UNSCALED q, k, v (what the to_qkv dgrad
 * consumes) -- from qkv (q block pre-scaled as above), out (forward output, for delta = rowsum(dout*out)), dout and lse.  delta f32 [B][H][T] is scratch.  Deterministic (no atomics). */
int gvk_attention_bwd_bf16(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                           int B, int T, int H, int ld_qkv, int ld_out, float scale, void* stream);
/* ONE-PASS form of the same backward (round 5; ABI 10): five MFMA products per (query, key) block instead of seven and one exponential
 * per score instead of two -- each workgroup owns dK / dV of its 128 keys and also forms dQ's share of those keys; the shares of a
 * (batch, head)'s key blocks are summed by an ordered hand-off between the workgroups (fixed order per query tile: bitwise reproducible,
 * no float atomics).  ws: at least gvk_attention_bwd_ws_bytes(B, T, H) bytes, 256-byte aligned, ZERO at allocation (the kernel leaves its
 * progress words zero); its layout depends on ws_bytes only, so ONE workspace sized for the longest sequence serves shorter ones too.
 * The int32 at byte gvk_attention_bwd_status_offset(ws_bytes) counts hand-off waits that ran into their bound and stays 0.  Calls that
 * share a workspace must be ordered by their stream.  Same arguments and results as gvk_attention_bwd_bf16 otherwise (delta f32
 * [B][H][T] is written by a small kernel in front). */
size_t gvk_attention_bwd_ws_bytes(int B, int T, int H);
size_t gvk_attention_bwd_status_offset(size_t ws_bytes);
int gvk_attention_bwd_bf16_fused(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, void* ws,
                                 size_t ws_bytes, int B, int T, int H, int ld_qkv, int ld_out, float scale, void* stream);
/* gvk_attention_bwd_bf16 when only the FIRST need_rows tokens of every sample carry a consumer (the bottom layer of a frozen backbone: of its
 * input only the prompt rows hold a trainable tensor, gaviko.py:540-548): dq, dk, dv of tokens < need_rows (rounded up to 128) are written --
 * the very bits the full call writes there -- the other rows of dqkv are left untouched; delta is complete. */
int gvk_attention_bwd_bf16_rows(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                                int B, int T, int H, int ld_qkv, int ld_out, float scale, int need_rows, void* stream);
/* the same with nn.Dropout(drop_p) on the attention probabilities (vision_transformer.py:52,68 -- live in training for the methods
 * that do not freeze the backbone): softmax statistics of the undropped scores, out = (P * mask / (1 - drop_p)) . V; the backward
 * regenerates mask(seed + *seed_ptr; b*H + head, query, key).  drop_p = 0 is the plain call. */
int gvk_attention_fwd_bf16_dropout(const void* qkv, void* out, float* lse, int B, int T, int H, int ld_qkv, int ld_out, float scale,
                                   float drop_p, uint64_t seed, const void* seed_ptr, void* stream);
int gvk_attention_bwd_bf16_dropout(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                                   int B, int T, int H, int ld_qkv, int ld_out, float scale, float drop_p, uint64_t seed, const void* seed_ptr,
                                   void* stream);

/* ------------------------------------------------------------------ rank-L ("skinny") fp32 projections of the trainable
 * side paths, L in {4, 8, 16, 20, 32}.  GAViKO: gaviko.py:231-232,242 (MWSA norm/proj_down/qkv/proj_up) and
 * gaviko.py:155-156,187 (GPA proj_down+QuickGELU / proj_up), plus their autograd dgrad / wgrad.
 *
 * down:  y[m][0:L] = act( LN?(drop?(x[m][:])) . W^T + bias );  optional  y2[m][0:L2] = y . W2^T
 *        act: 0 none, 1 QuickGELU (x*sigmoid(1.702x)); w_layout 0: W [L][C], 1: W [C][L];
 *        ln_gamma/ln_beta non-NULL: LayerNorm(eps) of the row first (mean/rstd saved when non-NULL);
 *        drop_p > 0: the input row is multiplied by the counter-based dropout mask(seed, m*C + c) / (1-p). */
typedef struct gvk_skinny_down_desc {
  const float* x; const float* w; const float* bias;
  const float* ln_gamma; const float* ln_beta; float* mean; float* rstd;
  float* z; float* y;               /* pre-activation / activated output [M][L], either may be NULL */
  const float* w2; float* y2;       /* second stage, W2 [L2][L] */
  const uint64_t* seed_ptr;          /* optional device word added to `seed` at run time (HIP-graph-safe dropout) */
  int32_t M, C, L, L2, act, w_layout;
  int32_t act_in;                    /* 1: QuickGELU applied to the input rows before the projection (DVPT share_MLP, dvpt.py:38) */
  float eps, drop_p;
  uint64_t seed;
} gvk_skinny_down_desc;
int gvk_skinny_down(const gvk_skinny_down_desc* d, void* stream);

/* up:    v = drop?( lat[m][0:L] . W^T + bias );  out = accumulate ? out + v : (res ? res + v : v)
 *        w_layout 0: W [C][L], 1: W [L][C].  lat_override (optional, [B][P][L]): rows with (m % T) < P take their latent
 *        from lat_override[(m / T) * P + m % T] (GPA: prompt rows replaced by the fused context, gaviko.py:181-185). */
typedef struct gvk_skinny_up_desc {
  const float* lat; const float* w; const float* bias; const float* res; float* out; const float* lat_override;
  const uint64_t* seed_ptr;
  /* optional LayerNorm-backward epilogue: out = base + LN'(v) with v = lat . W^T as the LN output gradient (no bias/dropout) */
  const float* ln_x; const float* ln_mean; const float* ln_rstd; const float* ln_gamma;
  void* out_bf16;                   /* optional bf16 copy of out (plain epilogue only) */
  const float* alpha_ptr;           /* optional device scalar: v = alpha * (lat . W^T + bias)          (DVPT prompt_gate, dvpt.py:46) */
  const float* gg_x;                /* optional f32 [M][C]: v *= QuickGELU'(gg_x[m][c])                 (dgrad through DVPT's input GELU) */
  /* optional second projection of the rows just written (plain epilogue only): z2 = out[m] . W2^T + bias2, y2 = act2(z2); W2 [L2][C]
     (GPA's proj_down of the new local tokens, gaviko.py:156, rides on the MWSA up-projection that produces them, gaviko.py:242) */
  const float* w2; const float* bias2; float* z2; float* y2;
  /* optional, with the LayerNorm-backward epilogue, w_layout 1 and w2 (L = 20 tile kernels only): a second rank-L term behind it,
     out = base + LN'(lat . W^T) + lat_b . W_b^T (W_b [L][C]), and the second projection reads `out` through the dropout mask (drop2_p, seed2):
     z2 = (out o mask) . W2^T with W2 in w2_layout (0: [L2][C], 1: [C][L2]).  One pass for three steps of the MWSA backward across a layer
     boundary: dL_in = dL_out + LN'(dlat . Wd) of layer i+1 (gaviko.py:231), dL += dzl . Wd_gpa of layer i (:156), dctx = proj_drop'(dL) . Wup
     of layer i (:242-243) */
  const float* lat_b; const float* w_b;
  /* optional, plain epilogue (L = 20 tile kernels only): the NEXT layer's MWSA entry on the rows just written (gaviko.py:231-232 of layer
     i+1 behind :242 of layer i), so the local stream is read once per layer instead of twice:
     nx_lat = LayerNorm(out; nx_ln_gamma, nx_ln_beta, nx_eps) . nx_w^T + nx_bias (nx_w [L][C]; nx_mean / nx_rstd [M] saved),
     nx_y2 = nx_lat . nx_w2^T (nx_w2 [nx_L2][L], nx_L2 <= 64: the qkv projection) */
  const float* nx_w; const float* nx_bias; const float* nx_ln_gamma; const float* nx_ln_beta; float* nx_mean; float* nx_rstd; float* nx_lat;
  const float* nx_w2; float* nx_y2;
  int32_t M, C, L, T, P, w_layout, accumulate;
  int32_t L2, act2, w2_layout, nx_L2;
  float drop_p, drop2_p, nx_eps;
  uint64_t seed, seed2;
} gvk_skinny_up_desc;
int gvk_skinny_up(const gvk_skinny_up_desc* d, void* stream);
/* out f32 [B*T][C] rows b*T + p (p < P) += (enh[b][p] - lat[b*T + p]) . w^T, w f32 [C][L]: the prompt rows of the GPA up-projection
 * (gaviko.py:183-187) when the plain-latent part rides the MLP GEMM as K-concatenation (gvk_pack_split_bf16) */
int gvk_prompt_up_fix(const float* enh, const float* lat, const float* w, float* out, int B, int T, int P, int C, int L, void* stream);
/* The same for a layer whose FIRST LayerNorm is folded into its qkv projection (gvk_gemm_desc.ln_mean): besides fixing the P prompt rows of
 * out (and of its bf16 copy out16, the folded GEMM's A operand) it finishes the row statistics -- mean / rstd f32 [B*T] of every row, from
 * the per-row partials part f32 [nparts][B*T][2] the fc2 GEMM left (gvk_gemm_desc.stat_part: sums of (x - pivot[m]) and of its square;
 * pivot f32 [B*T] = that GEMM's stat_pivot, NULL = 0), of the prompt rows from the rows themselves (two-pass). */
int gvk_prompt_up_fix_stats(const float* enh, const float* lat, const float* w, float* out, void* out16, const float* part, int nparts,
                            const float* pivot, float* mean, float* rstd, int B, int T, int P, int C, int L, float eps, void* stream);

/* outer: out[l][c] (transposed=0) or out[c][l] (transposed=1) (+)= sum_m narrow[m][l] * wide'[m][c];
 *        colsum[c] (+)= sum_m wide'[m][c] (optional).  wide' = LN(wide) when mean/rstd(/gamma/beta) are given, times the
 *        dropout mask when drop_p > 0.  scratch: f32 [128*(L+1)*C]; M (+ M2) <= 10240.  Deterministic (two-stage, no atomics). */
typedef struct gvk_outer_desc {
  const float* narrow; const float* wide; const float* lat_override;
  const float* mean; const float* rstd; const float* ln_gamma; const float* ln_beta;
  float* scratch; float* out; float* colsum;
  const uint64_t* seed_ptr;
  const float* narrow2; const float* wide2; /* optional second source [M2][L] / [M2][C] summed into the same result in the same pass
                                               (one weight fed by two token streams: GPA proj_down, gaviko.py:155-156); plain rows only */
  int32_t M, C, L, T, P, transposed, accumulate;
  int32_t wide_act;                  /* 1: QuickGELU applied to `wide` on the fly (DVPT: dW_d = dz^T . QuickGELU(x)) */
  int32_t M2;
  float drop_p;
  uint64_t seed;
} gvk_outer_desc;
int gvk_outer_reduce(const gvk_outer_desc* d, void* stream);
/* Gradients of y = LN(x) . W^T (W [L][C]) from Q[l][c] = sum_m dy[m][l] xhat[m][c] (gvk_outer_reduce with mean/rstd but no gamma/beta)
 * and S[l] = sum_m dy[m][l]:  dW (+)= gamma*Q + beta (x) S,  dgamma (+)= sum_l W*Q,  dbeta (+)= sum_l W*S,  dbias (+)= S (may be NULL). */
int gvk_ln_lowrank_affine(const float* Q, const float* S, const float* W, const float* gamma, const float* beta, float* dW, float* dgamma,
                          float* dbeta, float* dbias, int L, int C, int accumulate, void* stream);
/* out[j][l] (+)= sum_m a[m][j] * b[m][l]  (J, L <= 64); scratch f32 [64*J*L] */
int gvk_small_wgrad(const float* a, const float* b, float* out, float* scratch, int M, int J, int L, int accumulate, void* stream);
/* Several small reductions batched into one pair of launches (deterministic, two stages).  b == NULL: out[j] (+)= sum_m a[m][j],
 * j < J.  b != NULL: out[j][l] (+)= sum_m a[m][j] * b[m][l].  Up to 8 jobs per call; jobs of one call must not share `out`.
 * scratch: f32 [32 * total number of outputs]. */
typedef struct gvk_reduce_job {
  const float* a; const float* b; float* out;
  const float* a2;                   /* column sums only (b == NULL): M2 more rows [M2][J] summed into the same outputs */
  int32_t M, J, L, accumulate, M2;
} gvk_reduce_job;
int gvk_reduce_batch(const gvk_reduce_job* jobs, int njobs, float* scratch, void* stream);
/* Every parameter gradient of one rank-L side-path module of one layer in ONE launch (csrc/paramgrad.hip): up to 3 outer products
 *   out[l][c] (transposed=0) or out[c][l] (transposed=1) (+)= sum_m narrow'[m][l] * wide'[m][c],   colsum[c] (+)= sum_m wide'[m][c]
 * (narrow f32 [M][L], wide f32 [M][C]; narrow' = narrow with the rows t < P of every T-row sample taken from lat_override [.][P][L];
 * wide' = (wide - mean[m]) * rstd[m] when mean / rstd are given, times the dropout mask of element (m, c) when drop_p > 0; a second
 * source (narrow2, wide2, M2 plain rows) extends the same sum -- one weight fed by two token streams, gaviko.py:155-156), and up to 8
 * small reductions (gvk_reduce_job) over the same rows.  With aff_w (f32 [L][C] = the projection weight behind a LayerNorm, wide' = xhat)
 * the product Q is not stored: out[l][c] (+)= gamma_c Q[l][c] + beta_c S[l], aff_dgamma[c] (+)= sum_l w[l][c] Q[l][c],
 * aff_dbeta[c] (+)= sum_l w[l][c] S[l], aff_dbias[l] (+)= S[l] with S[l] = sum_m narrow[m][l] (gaviko.py:231: autograd of LN + proj_down).
 * Deterministic: partial tiles are summed in slab order by the workgroup that arrives last at a column tile's ticket word; no atomics on
 * data.  The hand-off issues NO release fence: partials leave as write-through (sc1) stores, every storing wave drains them
 * (s_waitcnt vmcnt(0)) before the workgroup barrier behind which one lane takes the agent-scope ticket, and only the last arriver runs
 * an agent-scope ACQUIRE (drops its CU's stale L1 lines) before reading -- the write-through form of the MI355X guide's hand-off recipe;
 * the ordering rests on sc1 stores having left the XCD once vmcnt retires them, not on a language-level release.  scratch f32 [gvk_param_grads_scratch(...)], tickets int32 [n_tickets] ZERO at allocation (the
 * kernel leaves them zero); calls that share scratch / tickets must be ordered by their stream.  C % 4 == 0, L % 4 == 0, L <= 28. */
typedef struct gvk_pgrad_outer {
  const float* narrow; const float* wide; const float* narrow2; const float* wide2; const float* lat_override;
  const float* mean; const float* rstd;
  float* out; float* colsum;
  const float* aff_w; const float* aff_gamma; const float* aff_beta; float* aff_dgamma; float* aff_dbeta; float* aff_dbias;
  int32_t M, M2, T, P, transposed, accumulate;
  int32_t C;                         /* columns (= row stride) of THIS job's wide operand; 0 = the call's C */
  float drop_p;
  uint64_t seed;
} gvk_pgrad_outer;
int64_t gvk_param_grads_scratch(const gvk_pgrad_outer* outer, int n_outer, const gvk_reduce_job* small, int n_small, int C, int L);
int gvk_param_grads(const gvk_pgrad_outer* outer, int n_outer, const gvk_reduce_job* small, int n_small, float* scratch,
                    int64_t scratch_elems, int32_t* tickets, int n_tickets, const void* seed_ptr, int C, int L, void* stream);
/* out[c] (+)= sum_m x[m][c]; scratch f32 [64*C] */
int gvk_colsum(const float* x, float* out, float* scratch, int M, int C, int accumulate, void* stream);

/* ------------------------------------------------------------------ GAViKO masked-window local self-attention core (fp32)
 * qkv f32 [B*N][3L] = [q | k | v] latents; single head; scale = model_dim^-0.5 (gaviko.py:201, NOT L^-0.5).
 * The 0/-inf [N,N] mask of gaviko.py:212-227 is index arithmetic here: query (d,h,w) attends key (d',h',w') iff
 * d - kd/2 <= d' < d - kd/2 + kd (likewise h, w), clipped to the DxHxW patch grid.  attn_drop (gaviko.py:239) is a
 * counter-based mask(seed, (b*N+i)*N + j) so that the backward regenerates it.
 * fwd writes ctx [B*N][L], lse [B*N];  bwd needs those + dctx and writes dqkv [B*N][3L] (delta [B*N] is scratch). */
typedef struct gvk_window_attn_desc {
  const float* qkv; float* ctx; float* lse;
  const float* dctx; float* delta; float* dqkv;
  const uint64_t* seed_ptr;
  int32_t B, D, H, W, kd, kh, kw, L;
  float scale, drop_p;
  uint64_t seed;
} gvk_window_attn_desc;
int gvk_window_attn_fwd(const gvk_window_attn_desc* d, void* stream);
int gvk_window_attn_bwd(const gvk_window_attn_desc* d, void* stream);

/* ------------------------------------------------------------------ GAViKO gated prompt awakening core (fp32), gaviko.py:159-185
 * Works on the activated latents xl [B*T][L] (global: rows [P prompts | cls | N image]) and ll [B*N][L] (local).
 * fwd: gates + both cross-attentions -> enh [B][P][L] (what replaces the prompt rows before proj_up); the other
 *      outputs are saved for the backward.  bwd: consumes dcomb [B*T][L] (= d proj_up input) and writes the gradients
 *      wrt the proj_down pre-activations (dzx [B*T][L], dzl [B*N][L]), dqg/dql [B][P][L] (query-projection outputs;
 *      wgrad = gvk_small_wgrad(dq, prm)), and gate_partials [B][gvk_gpa_gate_param_count(L,P)] -- per-sample gradients
 *      of [ca0_g L | ca0_b L | ca1_w 64L | ca1_b 64 | ca3_w 64P | ca3_b P | gl0_g L | gl0_b L | gl1_w L | gl1_b 1]
 *      (sum over B with gvk_colsum).  scale = L^-0.5 (gaviko.py:76). */
typedef struct gvk_gpa_desc {
  const float* xl; const float* ll;
  const float* ca0_g; const float* ca0_b; const float* ca1_w; const float* ca1_b; const float* ca3_w; const float* ca3_b;
  const float* gl0_g; const float* gl0_b; const float* gl1_w; const float* gl1_b;
  const float* wgq; const float* bgq; const float* wlq; const float* blq;
  float* imp; float* gw; float* enh;
  float* prm; float* qg; float* ql; float* cg; float* cl; float* lse_g; float* lse_l;
  const float* dcomb; const float* zx; const float* zl;
  float* dimp; float* dgw_part;
  float* dqg; float* dql; float* dcg; float* dcl; float* delta_g; float* delta_l; float* dprm;
  float* dcls; float* gate_partials; float* dzx; float* dzl;
  void* enh16;            /* optional (fwd): split-bf16 copy [hi | lo | hi] of enh[b][p][:] (gvk_pack_split_bf16, activation side) into row
                             b*T + p, columns col16 .. col16 + 3L - 1 of a [B*T][ld16] GEMM operand */
  int32_t B, T, N, P, L, ld16, col16;
  float scale;
} gvk_gpa_desc;
int gvk_gpa_fwd(const gvk_gpa_desc* d, void* stream);
int gvk_gpa_bwd(const gvk_gpa_desc* d, void* stream);
int gvk_gpa_gate_param_count(int L, int P);

/* ------------------------------------------------------------------ token assembly and the pooled head
 * rows_broadcast: out[b][row_off + r][:] = src[r][:] + add[r][:]  (cls_token + pos[0]; prompts + prompt_pos;
 *                 vision_transformer.py:154-156, gaviko.py:536-543, vpt.py:127-131).  out is [B*T][C].
 * rows_batch_sum: out[r][:] (+)= sum_b dg[b][row_off + r][:]; out2 (optional) receives the same (two parameters that
 *                 enter as a sum share a gradient: prompt_embeddings / prompt_positional_embedding). */
int gvk_rows_broadcast(float* out, const float* src, const float* add, int B, int T, int row_off, int R, int C, void* stream);
int gvk_rows_batch_sum(const float* dg, float* out, float* out2, int B, int T, int row_off, int R, int C, int accumulate, void* stream);
/* head: logits[b] = Wh . mean_{r in [r0, r0+R)} LN(g[b][r]) + bh   -- the final LayerNorm is evaluated only on the rows
 * the head consumes (gaviko.py:306,316: rows 0..P; vision_transformer.py:89,161: row 0, or all rows for pool='mean').
 * bwd: dwh/dbh (+)= ...; dg rows r0..r0+R of every sample receive the LN backward (caller zero-fills the rest). */
typedef struct gvk_head_desc {
  const float* g; const float* ln_gamma; const float* ln_beta; const float* wh; const float* bh;
  float* logits; float* pooled;
  const float* dlogits; float* dg; float* dwh; float* dbh;
  int32_t B, T, C, K, r0, R, accumulate;
} gvk_head_desc;
int gvk_head_fwd(const gvk_head_desc* d, void* stream);
int gvk_head_bwd(const gvk_head_desc* d, void* stream);

/* ------------------------------------------------------------------ VPT (model/vpt.py)
 * small_linear: out[r][:] = W . x[r] + b for a handful of rows (prompt_proj = Linear(prompt_dim, C), vpt.py:56,127-131);
 *   bwd: dw[c][k] (+)= sum_r dout[r][c] x[r][k], db (+)= colsum(dout), dx[r][k] (+)= sum_c dout[r][c] w[c][k] (any output may be NULL).
 * vpt_repack: the per-layer sequence rebuild of deep VPT (vpt.py:147-153): out = [in[:,0] | prompt (P rows) | in[:, 1+skip:]],
 *   Tout = Tin - skip + P, skip = prompt_dim for layers > 0 (reference quirk: drops the old prompts AND skip-P patch tokens);
 *   bwd scatters dout back (rows 1..skip of din are zero).  Prompt-row gradients: gvk_rows_batch_sum. */
int gvk_small_linear_fwd(const float* x, const float* w, const float* b, float* out, int R, int K, int C, void* stream);
int gvk_small_linear_bwd(const float* x, const float* w, const float* dout, float* dw, float* db, float* dx, int R, int K, int C,
                         int accumulate, void* stream);
int gvk_vpt_repack_fwd(const float* in, const float* prompt, float* out, int B, int Tin, int Tout, int P, int skip, int C, void* stream);
int gvk_vpt_repack_bwd(const float* dout, float* din, int B, int Tin, int Tout, int P, int skip, int C, void* stream);
/* LoRA merge (model/melo.py:41-47): out f32 [3C][C] = W + s * [B_q.A_q ; 0 ; B_v.A_v], s = alpha // r, A [r][C], B [C][r].
 * The merged matrix is then cast to the bf16 GEMM operand (and its transpose for the dgrad) like any other weight. */
int gvk_lora_merge_f32(const float* w, const float* a_q, const float* b_q, const float* a_v, const float* b_v, float* out, int C, int r,
                       float s, void* stream);
/* bf16 [M][ld_in] column block -> dense f32 [M][C] (gradient blocks handed to the fp32 rank-r kernels) */
int gvk_cast_bf16_f32_strided(const void* in, float* out, int M, int C, int ld_in, void* stream);

/* ------------------------------------------------------------------ SSF, `--method ssf` (model/ssf.py)
 * ssf_ada(x, s, t) = x*s + t (ssf.py:24-31) always follows a LayerNorm or a Linear, so the forward runs the plain ViT
 * kernels on effective parameters and the backward adds two column sums per site.
 * gvk_ssf_fold_weight: out[n][k] = w[n][k]*s[n] (bf16, or fp32 when out_f32), out_t (may be NULL) = its transpose [K][N].
 * gvk_ssf_fold_vec:    out = a*s + t   (a NULL -> t alone: the bias-free to_qkv; t NULL -> a*s: a LayerNorm gamma).
 * gvk_ssf_colgrad:     dt[n] = sum_m dy[m][n];  ds[n] = sum_m dy[m][n]*z[m][n] with z = (y - t)/s the pre-ssf value,
 *                      y = y0 (bf16 / fp32) [- y1 fp32] [- pos[m % rows_in][n]]; optional row mapping as GVK_EPI_PATCH_F32;
 *                      scratch f32 [64*2*N].
 * gvk_ssf_ln_grad:     LayerNorm+ssf site, from the effective-affine gradients (gvk_layernorm_bwd_affine):
 *                      ds = gamma*dgamma' + beta*dbeta', dt = dbeta'.
 * gvk_ssf_head_grad:   the final norm + ssf in front of the head (only the pooled rows r0..r0+R carry gradient). */
typedef struct gvk_ssf_colgrad_desc {
  const void* dy; const void* y0; const float* y1; const float* pos;
  const float* s; const float* t; float* ds; float* dt; float* scratch;
  int32_t M, N, ld_dy, ld_y, dy_f32, y0_f32, rows_in, rows_out, row_off;
  int32_t y0_cols;   /* columns n < y0_cols of y0 are multiplied by y0_mul when read: undoes the attention pre-scale of the q block of a saved */
  float y0_mul;      /* qkv (gvk_gemm_desc.scale_cols), y0_mul = 1 / col_scale; 0 columns = off */
  float y_mul;       /* y = y_mul * (y0 - y1) - pos: a site whose output went through nn.Dropout before it was stored (to_out / fc2 + ssf_2 of an
                        unfrozen backbone, the patch embedding under emb_dropout) is read back as kept / (1 - p), so y_mul = 1 - p with dy = the
                        MASKED gradient (dropped elements then contribute nothing); 0 is taken as 1 */
} gvk_ssf_colgrad_desc;
int gvk_ssf_fold_weight(const float* w, const float* s, void* out, void* out_t, int N, int K, int out_f32, void* stream);
int gvk_ssf_fold_vec(const float* a, const float* s, const float* t, float* out, int n, void* stream);
int gvk_ssf_colgrad(const gvk_ssf_colgrad_desc* d, void* stream);
int gvk_ssf_ln_grad(const float* dgamma_eff, const float* dbeta_eff, const float* gamma, const float* beta, float* ds, float* dt, int n,
                    void* stream);
int gvk_ssf_head_grad(const float* g, const float* mean, const float* rstd, const float* wh, const float* dlogits, const float* gamma,
                      const float* beta, float* ds, float* dt, int B, int T, int C, int K, int r0, int R, void* stream);

/* ------------------------------------------------------------------ DVPT, `--method dvpt` (model/dvpt.py)
 * Latent-space core of share_MLP (dvpt.py:37-47) on z = proj_d(QuickGELU(x)) f32 [B*T][L] (gvk_skinny_down with act_in = 1;
 * rows per sample: P prompts | cls | T-P-1 patches):
 *   gvk_dvpt_fwd: enh[b][p] = softmax(scale * z_p . z_patches^T) . z_patches, lse saved            (scale = d_model^-1/2)
 *   gvk_dvpt_bwd: from dcomb = dy . W_u (NOT yet scaled by the gate) and colsum_dy = sum_m dy:
 *                 dgate = <dcomb, lat'> + <bu, colsum_dy>;  dz (all rows) = backward of the attention and of the concatenation
 * (the up-projection and the gate are gvk_skinny_up with lat_override / alpha_ptr; gvk_scale_dev applies the gate to dW_u, db_u). */
typedef struct gvk_dvpt_desc {
  const float* z; float* enh; float* lse;               /* z [B*T][L]; enh [B][P][L], lse [B][P]: written by fwd, read by bwd */
  const float* dcomb; const float* gate; const float* bu; const float* colsum_dy;
  float* delta; float* dz; float* dgate;                 /* delta [B][P] scratch; dz [B*T][L]; dgate [1] */
  int32_t B, T, P, L, C;
  float scale;
} gvk_dvpt_desc;
int gvk_dvpt_fwd(const gvk_dvpt_desc* d, void* stream);
int gvk_dvpt_bwd(const gvk_dvpt_desc* d, void* stream);
int gvk_scale_dev(float* x, const float* alpha, int64_t n, void* stream);   /* x[i] *= alpha[0], alpha on the device */

/* ------------------------------------------------------------------ EVP, `--method evp` (model/evp.py)
 * gvk_evp_highpass: PromptGenerator.fft (evp.py:126-147) as it executes on [B,1,D,H,W]: out[b,d] = |hp . img[b,d]| on the depth slices
 *   with depth_mask[d] != 0, |img[b,d]| on the others; hp f32 [H][H] = I - Re(F^-1 diag(band) F) and the slice mask are built by
 *   the host (gaviko_amd/engine.py::evp_highpass_operator) from the reference's mask indexing.  No FFT library involved.
 * The remaining entry points are glue for the rank-(dim/32) prompt latents (zero-padded to a width the rank-L kernels support):
 *   gvk_pad2d_f32   dst (drows x dcols) = src (rows x cols, optionally transposed), zero elsewhere
 *   gvk_add2d_f32   out = a + b on a rows x cols window with independent leading dimensions
 *   gvk_gelu_fwd/bwd_f32  exact (erf) GELU of the light-weight MLPs (evp.py:45-49) and dy * GELU'(x)
 *   gvk_rows_patch  tok[b][row_off+n] (= or +=) src[b*N+n] (+ pos[n])     (tokens = conv + pos; x[:,1:] += prompt_i, evp.py:235-238)
 *   gvk_rows_gather the inverse read (the prompt gradient is the patch rows of the layer-input gradient) */
int gvk_evp_highpass(const float* img, const float* hp, const int32_t* depth_mask, float* out, int B, int D, int H, int W, void* stream);
int gvk_pad2d_f32(const float* src, int ld_src, int rows, int cols, int transpose, float* dst, int ld_dst, int drows, int dcols, void* stream);
int gvk_add2d_f32(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int rows, int cols, void* stream);
int gvk_gelu_fwd_f32(const float* x, float* y, int64_t n, void* stream);
int gvk_gelu_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, void* stream);
int gvk_rows_patch(float* tok, const float* src, const float* pos, int B, int T, int N, int C, int row_off, int accumulate, void* stream);
int gvk_rows_gather(const float* tok, float* dst, int B, int T, int N, int C, int row_off, void* stream);

/* ------------------------------------------------------------------ optimisation step
 * Replaces train.py:315-319: torch.nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.Adam.step() over every
 * trainable tensor, on the engine's flat fp32 gradient buffer (exp_avg / exp_avg_sq share its layout).
 * gvk_sumsq: out[0] = sum x^2, two-stage deterministic; scratch f32 [256].
 * gvk_adam_step: one workgroup per table row; blk_tab int32 [nblocks][4] = {tensor id, offset inside that tensor, offset
 * into the flat buffers, element count (<= 1024)}; ptr_tab uint64 [ntensors] = parameter data pointers.  norm_sq (device
 * scalar from gvk_sumsq over the same flat gradient) enables clipping: g *= min(1, max_norm / (sqrt(norm_sq) + 1e-6)),
 * written back as torch does.  lr / beta1 are the OneCycleLR values of this step (train.py:197-206, host-side schedule);
 * bias_c1 = 1 - beta1_product ... exactly torch's single-tensor Adam: p -= lr/bias_c1 * m / (sqrt(v)/sqrt(bias_c2) + eps). */
typedef struct gvk_adam_desc {
  const void* ptr_tab; const void* blk_tab;
  float* grad; float* m; float* v;
  const float* norm_sq;                    /* NULL: no clipping */
  int32_t nblocks;
  float lr, beta1, beta2, eps, bias_c1, bias_c2, max_norm;
} gvk_adam_desc;
int gvk_sumsq(const float* x, int64_t n, float* scratch, float* out, void* stream);
int gvk_adam_step(const gvk_adam_desc* d, void* stream);

/* ---- loss seed (caller side of the path) --------------------------------------------------------------------------
 * Replaces train.py:176-179 + 283/306 (criterion = FocalLoss(gamma=1.2) | CrossEntropyLoss; loss = criterion(outputs,
 * labels); loss.backward() seed) and the two per-step host reads of train.py:327-328.  One launch: loss[0] = the reduced
 * loss, dlogits = d loss / d logits, and (meter != NULL) meter[0] += loss*B, meter[1] += #(argmax == target),
 * meter[2] += B.  GVK_LOSS_FOCAL follows losses/focal_loss.py:84-115 as it executes (clamp to [eps, 1-eps] + softmax,
 * TWICE; weights = per-class rescaling or NULL; ignore_index rows contribute nothing).  target is int64 [B]. */
enum { GVK_LOSS_CE = 0, GVK_LOSS_FOCAL = 1 };
typedef struct gvk_loss_desc {
  const float* logits; const void* target; const float* weights;
  float* loss; float* dlogits; float* meter;
  int32_t B, K, kind, reduction;           /* reduction 0 = 'mean', 1 = 'sum', 2 = 'none' (loss then receives the B per-sample values, dlogits each row's own gradient) */
  float gamma, eps;
  int64_t ignore_index;
} gvk_loss_desc;
int gvk_loss_fwd_bwd(const gvk_loss_desc* d, void* stream);

/* ---- data side and evaluation metrics ------------------------------------------------------------------
 * Replace the torchio transforms of train.py:38-62 on a batch of raw volumes x [B][V] (V = D*H*W, float32, resident in HBM):
 * gvk_volume_minmax: partials [B][gvk_minmax_partials()] = per-slab (min, max) pairs of every volume.
 * gvk_rescale_intensity: tio.RescaleIntensity(out_min_max): y = ((x - min) / (max - min)) * (out_max - out_min) + out_min in
 *   float32, in that order; a constant volume passes through unchanged; minmax (optional) receives [B][2].  In place allowed.
 * gvk_spatial_transform: tio.RandomAffine + tio.RandomFlip as ONE resampling pass: out[b] at output voxel q reads in[b] at
 *   A_b . mirror(q) + t_b (mats [B][12], row-major 3x4 in array-axis order) with trilinear weights; neighbours outside the volume
 *   read the volume minimum (partials of the INPUT).  flags[b]: bits 0..2 = mirror axis 0..2, bit 3 = affine live (else exact gather).
 * Replace eval.py:103-122: gvk_eval_rows: proba = softmax(logits), pred = argmax, confusion uint64 [K][K] += (target, pred);
 * gvk_ovr_auc_counts: counts uint64 [K][3] += {2*#(p_pos > p_neg) + #(p_pos == p_neg), n_pos, n_neg} per class (one-vs-rest). */
int gvk_volume_minmax(const float* x, float* partials, int B, int64_t V, void* stream);
int gvk_minmax_partials(void);
int gvk_rescale_intensity(const float* x, const float* partials, float* y, float* minmax, int B, int64_t V, float out_min, float out_max, void* stream);
int gvk_spatial_transform(const float* in, float* out, const float* mats, const int32_t* flags, const float* partials, int B, int D, int H, int W,
                          void* stream);
int gvk_eval_rows(const float* logits, const void* target, float* proba, int32_t* pred, void* confusion, int N, int K, void* stream);
int gvk_ovr_auc_counts(const float* proba, const void* target, void* counts, int N, int K, void* stream);

/* ---- nn.Dropout as its own pass (sites without a producing kernel to fuse into) ------------------------------------------
 * Replaces vision_transformer.py:157 (emb_dropout), vpt.py:129,148,152 (prompt_dropout) and carries the masks of the
 * Linear+Dropout pairs (vision_transformer.py:34,54) onto the gradient side.  out32 / out16 (either may be NULL, out32 may alias x)
 * = x * mask(seed + *seed_ptr, m*N + n) / (1 - drop_p) over logical rows m < M; with rows_in > 0 logical row m lives in buffer row
 * (m / rows_in) * rows_out + row_off + m % rows_in (a row range of every sample).  The mask function is the one of gvk_gemm_desc.drop_p. */
typedef struct gvk_dropout_desc {
  const float* x; float* out32; void* out16; const void* seed_ptr;
  int32_t M, N, ld, rows_in, rows_out, row_off;
  float drop_p;
  uint64_t seed;
} gvk_dropout_desc;
int gvk_dropout_rows(const gvk_dropout_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif
