// Argument blocks and the dropout hash shared by the rank-L side-path kernels (skinny.hip, rowwise.hip).
#pragma once
#include "common.hpp"
#include "dropout.hpp"

namespace gvk {

struct DownArgs {
  const float* x; const float* w; const float* bias;      // x [M][C]; w [L][C] (layout 0) or [C][L] (layout 1)
  const float* ln_g; const float* ln_b;                   // optional LayerNorm on the input row (eps 1e-5)
  float* mean; float* rstd;                               // saved LN statistics (optional)
  float* z; float* y;                                     // pre-activation (optional) / activated output [M][L]
  const float* w2; float* y2; int L2;                     // optional second stage y2[m][0:L2] = y . w2^T, w2 [L2][L]
  int M, C, act, w_layout;
  float eps;
  unsigned long long seed; const unsigned long long* seed_ptr; unsigned int drop_thresh; float inv_keep;   // dropout mask on the INPUT (bwd of proj_drop)
  // row-per-wave kernel only (rowwise.hip): LayerNorm kernels that also project the rows they hold
  //   mode 1: y16 = LN(x; ln_g, ln_b) (bf16), mean/rstd saved, and the projection is taken of the RAW row x
  //   mode 2: dx = dres + LN'(dy; x, mean_in, rstd_in, ln_g) (+ bf16 copy dx16), and the projection is taken of dx
  //   mode 3: QuickGELU applied to the input rows before the projection (DVPT share_MLP, dvpt.py:38)
  int mode;
  bf16* y16;
  int tile_off;                                            // sidepass.hip: first 16-row tile of this launch (chunked launches, GAVIKO_HIP_SIDE_CHUNKS)
  bf16* ysplit; int ysplit_ld, ysplit_col;                // row-per-wave kernel: split-bf16 copy [hi | lo | hi] of y into spare K columns of a GEMM operand
  const float* dy; const float* mean_in; const float* rstd_in; const float* dres; float* dx; bf16* dx16;
  const bf16* dy16;                                        // mode 2: the LayerNorm output gradient as bf16 (then dy is NULL): what a dgrad GEMM stores
};

struct UpArgs {
  const float* lat; const float* w; const float* bias;    // lat [M][L]; w [C][L] (layout 0) or [L][C] (layout 1)
  const float* res; float* out;                           // out = res + (...)  (res may be NULL / alias out); accumulate: out += (...)
  const float* lat_override; int T, P;                    // rows with (m % T) < P read lat_override[(m / T) * P + m % T][:]
  int tile_off;                                            // sidepass.hip: first 16-row tile of this launch
  const float* ln_x; const float* ln_mean; const float* ln_rstd; const float* ln_g;   // optional LayerNorm-backward epilogue
  bf16* out16;                                            // optional bf16 copy of `out` (the next dgrad GEMM's operand)
  const float* alpha_ptr;                                 // optional device scalar: v = alpha * (lat . W + bias)   (DVPT prompt_gate)
  const float* gg_x;                                      // optional [M][C]: v *= QuickGELU'(gg_x[m][c])           (DVPT dgrad through the input GELU)
  int M, C, w_layout, accumulate;
  unsigned long long seed; const unsigned long long* seed_ptr; unsigned int drop_thresh; float inv_keep;   // dropout on the projected value (proj_drop)
};


int launch_row_down(const DownArgs& a, int L, hipStream_t s);   // rowwise.hip; returns 1 when the shape is not covered (all modes)
int launch_row_up(const UpArgs& a, int L, hipStream_t s);
// sidepass.hip: 16-row tiles on the fp32 matrix cores; return 1 when the call is not covered.  side_up optionally projects the rows it
// has just written with a second weight w2 [L2][C] (+ bias2, activation act2) into z2 / y2 [M][L2].
int launch_side_down(const DownArgs& a, int L, hipStream_t s);
// ln_dy (with a's LayerNorm operands): out = base + LN'(ln_dy) + lat . W^T instead of base + LN'(lat . W^T)
// ex (with a's LayerNorm operands and w2): out = base + LN'(lat . W^T) + ex->lat2 . ex->w2up^T, and the second projection (w2 in layout
// ex->w2_layout, 0: [L][C], 1: [C][L]) reads the rows through the dropout mask (ex->seed2, ex->drop2_*)
struct UpExtra {
  const float* lat2; const float* w2up;        // [M][L], [L][C]
  int w2_layout;
  unsigned long long seed2; const unsigned long long* seed_ptr; unsigned int drop2_thresh; float inv_keep2;
};
// nx (plain epilogue only): the next layer's MWSA entry on the rows just written: lat = LayerNorm(out; g, b) . w^T + bias (w [L][C], statistics
// saved), y2 = lat . w2^T (w2 [L2][L], L2 <= 64)
struct UpNext {
  const float* w; const float* bias; const float* g; const float* b; float* mean; float* rstd; float* lat; const float* w2; float* y2;
  int L2; float eps;
};
int launch_side_up(const UpArgs& a, int L, const float* w2, const float* bias2, float* z2, float* y2, int L2, int act2, hipStream_t s,
                   const float* ln_dy = nullptr, const UpExtra* ex = nullptr, const UpNext* nx = nullptr);

}  // namespace gvk
