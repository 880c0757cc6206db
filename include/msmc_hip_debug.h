/* msmc_hip_debug.h -- NOT part of the product ABI (include/msmc_hip.h).
 *
 * Process-global A/B switches, ablation masks and one experimental entry point that libmsmc_hip.so also exports for the
 * perf tools (tools/), the CPU kernel-interpreter tests and the forced-variant GPU tests.  A production caller never
 * includes this header: every kernel choice that matters to a caller is per call (msmc_conv_desc.variant / split_shift),
 * and the product package (msmc-tts_amd/msmctts_amd) touches none of these except the two environment-driven sweeps read
 * once in hip/lib.py (MSMC_WGRAD_TPW, MSMC_GATHER4_GROUPING).  All switches are plain ints read at launch time.
 */
#ifndef MSMC_HIP_DEBUG_H
#define MSMC_HIP_DEBUG_H
#include "msmc_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* DIAGNOSTICS (tools/bench_vq.py ABLATE=...), 0 in production (any other value selects a separate diagnostics build of the
 * kernel; results are garbage for bits 0-4): bit 0 one codeword tile instead of K/16, 1 no exact paths, 2 no codeword-row
 * gather, 3 no stores, 4 no frame loads, 5 phase timers: shader cycles per phase of a step of every wave of workgroup 0 into
 * slow_count[2 + 8 w ..] (slow_count then holds 2 + 8 * 8 words). */
void msmc_vq_set_shortlist_ablate(int mask);

/* Perf-experiment switch: 0 selects the LDS-tile search kernel for every shape; default 1. */
void msmc_vq_set_variant(int v);

/* Tests / sweeps: workgroups of the persistent grid of msmc_conv_gather variant 32 (0 = one or two per CU). */
void msmc_conv_set_gather4_grid(int n);
/* 1: the variant-32 members of msmc_conv_gather_group share ONE persistent grid (interpreter-tested, not yet timed on the
 * GPU); default 0: one launch per member. */
void msmc_conv_set_gather4_grouping(int on);
/* Perf-experiment switch: 0 selects the simple (un-pipelined) gather kernel everywhere; default 1. */
void msmc_conv_set_pipeline(int on);
/* Perf-sweep switches: force the weight-gradient pixel split (0 = model), allow 32-channel N tiles for small grids. */
void msmc_conv_set_wgrad_split(int n);
/* Perf-sweep switch: accumulators (32x32 output blocks) per wave of the second-generation weight gradient, 1..5 (default 5:
 * fewest re-reads of the staged tiles; fewer = more workgroups per CU). */
void msmc_conv_set_wgrad_tpw(int n);
/* 2 (default) = second-generation bf16 weight-gradient kernel, 1 = first generation (A/B tests) */
void msmc_conv_set_wgrad_generation(int n);
/* 2 (default) = second-generation forward / data-gradient gather kernel, 1 = first generation (A/B tests) */
void msmc_conv_set_gather_generation(int n);
void msmc_conv_set_narrow(int on);

/* 0: grouped entry points launch their members one by one (A/B tests); default 1. */
void msmc_conv_set_grouping(int on);

/* DIAGNOSTICS (tools/bench_wgrad_splits.py ABLATE=...), 0 in production: 1 skips the MFMA steps of the fourth-generation
 * weight gradient, 2 its LDS-DMA stream; results are then garbage. */
void msmc_conv_set_wgrad4_ablate(int mask);

/* EXPERIMENTAL (measured by tools/bench_resunit.py, not on the train step's path): one ResBlock1 unit (reference
 * msmctts/networks/hifigan/common.py:44-51, one (c1, c2) pair) as ONE launch, bf16, C = 32 / 64 channels, odd k <= 11:
 *   a = lrelu(conv1d(lrelu(x), w1, dilation dil1) + b1);  y = conv1d(a, w2, dilation 1) + b2 + x
 * x, a, y [B][L][C]; w1, w2 [k][C][C] in the forward layout (tap, output channel, input channel); b1, b2 fp32 [C];
 * nt = tiles of 32 rows per wave step (2 or 3; anything else: chosen from the LDS footprint). */
int msmc_resunit_forward(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* a, void* y,
                         int B, int L, int C, int k, int dil1, float slope, int nt, msmc_stream stream);

#ifdef __cplusplus
}
#endif
#endif
