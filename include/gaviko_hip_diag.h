/* Diagnostics of libgaviko_hip_diag.so (`python -m gaviko_amd.build --diag`: the product sources compiled with -DGVK_DIAG).  NOT part of
 * the product ABI: the product library (libgaviko_hip.so, include/gaviko_hip.h) exports none of these and ignores every GAVIKO_HIP_* A/B
 * switch named in the kernel sources (gvk::diag_env).  tools/ loads the diag library when GAVIKO_HIP_DIAG=1.  (The experiment kernels
 * of rounds 2-3 -- eight-wave split-k tiles, stream-K, the fused gather patch GEMM -- were measured slower and deleted: DESIGN.md 7b.1,
 * 7b.5, 7c.5b keep the numbers.) */
#ifndef GAVIKO_HIP_DIAG_H
#define GAVIKO_HIP_DIAG_H
#include "gaviko_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* launches recorded on `stream` from now on are replaced by an empty kernel (contention studies; results are garbage); _clear undoes it */
int gvk_plan_nop_stream(void* stream);
int gvk_plan_nop_clear(void);

#ifdef __cplusplus
}
#endif
#endif
