/* msmc_hip.h -- C ABI of libmsmc_hip.so, the gfx950 kernels under the msmctts.networks boundary.
 *
 * The reference's drop-in boundary for this path is a Python plugin registry, not an FFI
 * (reference msmctts/networks/__init__.py:6-11).  This library sits *below* it:
 * every entry point takes plain device pointers, sizes and a HIP stream, allocates nothing, keeps no
 * global state, is stream-ordered and re-entrant per stream, and returns 0 (hipSuccess) or a
 * hipError_t / negative MSMC_E* code.  The Python host side (msmc-tts_amd/msmctts_amd) binds these with
 * ctypes from modules that carry the reference's class names; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Layout conventions
 *   frames    x        [N][D] fp32 row-major, N = B*T_s frames (padded frames included)
 *   codebook  embed    [H][d][K] fp32 (the reference's per-head buffer layout (d, K), packed over heads)
 *             embed_t  [H][K][d] fp32, enorm [H][K] fp32 -- derived per step by msmc_vq_prepare
 *   indices   ind      [N][H] int64
 *   diff      diff     [N][d] fp32 (mean over heads of the element-wise squared error)
 */
#ifndef MSMC_HIP_H
#define MSMC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* msmc_stream;          /* hipStream_t */

#define MSMC_E_SHAPE (-2)           /* unsupported shape (see each function) */
#define MSMC_E_WORKSPACE (-3)       /* workspace too small */

/* Library identity: "gfx950" for the product build, "emu" for the CPU interpreter used by tests. */
const char* msmc_backend(void);
int msmc_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * V1/V2  multi-head nearest-codeword search with EMA update.
 * Replaces Quantize.forward / MultiHeadQuantize.forward,
 *   reference msmctts/networks/vqgantts/modules.py:24-67 and :137-151.
 * ------------------------------------------------------------------------------------------- */

/* embed [H][d][K] -> embed_t [H][K][d], enorm[h][k] = sum_j embed[h][j][k]^2   (modules.py:29). */
int msmc_vq_prepare(const float* embed, float* embed_t, float* enorm, int H, int d, int K,
                    msmc_stream stream);

/* Search + gather + straight-through value + head-averaged squared error, all heads, one launch.
 *   dist = (|x|^2 - 2 x.e_k) + |e_k|^2 in fp32, first-minimum tie rule   (modules.py:26-31)
 *   quant[n] = x + (e_best - x) ; diff[n] = (sum_h (e_best - x_h)^2) / H  (modules.py:33,59-60,147)
 * Requires d % 4 == 0, K % 16 == 0, one head's transposed codebook <= 80 KiB of LDS.
 * quant may alias x.  Returns MSMC_E_SHAPE otherwise. */
int msmc_vq_search(const float* x, const float* embed_t, const float* enorm, float* quant, float* diff,
                   int64_t* ind, int N, int D, int H, int K, msmc_stream stream);

/* Bytes of scratch msmc_vq_ema_update needs for these sizes. */
size_t msmc_vq_ema_workspace(int N, int D, int H, int K);

/* EMA statistics over the valid frames (t < length[b]) followed by the in-place buffer update
 *   cluster_size <- decay*cluster_size + (1-decay)*count ; embed_avg <- decay*embed_avg + (1-decay)*sum
 *   embed <- embed_avg / ((cluster_size+eps)/(n+K*eps)*n)                     (modules.py:35-57)
 * Deterministic: per-tile partial sums are reduced in a fixed order.
 * x [B*T][D], ind [B*T][H], length [B] int64. */
int msmc_vq_ema_update(const float* x, const int64_t* ind, const int64_t* length, float* embed,
                       float* cluster_size, float* embed_avg, void* workspace, size_t workspace_bytes,
                       int B, int T, int D, int H, int K, float decay, float eps, msmc_stream stream);

/* Backward of (quant, diff) wrt x:  gx = g_quant + g_diff * 2*(x - quant)/H   (straight-through +
 * the un-reduced commitment term, modules.py:59-60).  g_diff may be NULL (treated as zero). */
int msmc_vq_backward(const float* g_quant, const float* g_diff, const float* x, const float* quant,
                     float* gx, int N, int D, int H, msmc_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MSMC_HIP_H */
