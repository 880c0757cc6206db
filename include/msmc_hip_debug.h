/* msmc_hip_debug.h -- NOT part of the product ABI (include/msmc_hip.h).
 *
 * Process-global A/B switches, ablation masks and the profiling observers that libmsmc_hip.so also exports for the perf
 * tools (tools/), bench.py's per-kernel table, the CPU kernel-interpreter tests and the forced-variant GPU tests.  A production caller never
 * includes this header: every kernel choice that matters to a caller is per call (msmc_conv_desc.variant / split_shift),
 * and the product package (msmc-tts_amd/msmctts_amd) touches none of these.
 * All switches are plain ints read at launch time.
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

/* ---- observers (read-only: nothing here changes what a product call computes or launches) ---- */
/* Symbol of the search kernel the calling thread's most recent msmc_vq_search launched (profiling aid). */
const char* msmc_vq_last_kernel(void);
/* Per-launch profiling log (process-wide; bench.py's kernel table): while enabled, every kernel this library launches
 * -- from any thread: the backward pass runs on the autograd engine's -- is bracketed by a HIP event pair recorded on the launch's own stream and logged under the
 * symbol rocprofv3 prints for it (template arguments included where the launcher knows the instantiation, the template's
 * name otherwise).  msmc_prof_enable(1) clears the log and starts recording, (0) stops; msmc_prof_read synchronises
 * on the record's end event and returns its duration in milliseconds (0 on success).  At most 16384 records; off by
 * default (cost when off: one thread-local flag test per launch). */
void msmc_prof_enable(int on);
int msmc_prof_count(void);
int msmc_prof_read(int i, char* name, int cap, float* ms);
/* Symbol of the kernel the calling thread's most recent msmc_conv_gather / msmc_conv_wgrad launched (profiling aid). */
const char* msmc_conv_last_kernel(void);
/* Number of kernels the calling thread's msmc_conv_gather / msmc_conv_wgrad calls have launched so far. */
long msmc_conv_launch_count(void);
#ifdef __cplusplus
}
#endif
#endif
