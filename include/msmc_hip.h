/* msmc_hip.h -- C ABI of libmsmc_hip.so, the gfx950 kernels under the msmctts.networks boundary.
 *
 * The reference's drop-in boundary for this path is a Python plugin registry, not an FFI
 * (reference msmctts/networks/__init__.py:6-11).  This library sits *below* it:
 * every entry point takes plain device pointers, sizes and a HIP stream, allocates nothing (msmc_stream_create, which
 * makes a HIP stream, is the one exception), is stream-ordered and re-entrant per stream, and returns 0 (hipSuccess) or a
 * hipError_t / negative MSMC_E* code.  No entry point of this header changes or observes process-global state: kernel choices are
 * per call (msmc_conv_desc.variant / split_shift); the one piece of per-thread state is the sink of msmc_conv_wgrad_defer_begin
 * / _end, armed around single calls.  The A/B switches, ablation masks, the profiling observers (last-kernel tags, launch
 * counter, the msmc_prof_* launch log) that the perf tools, bench.py's kernel table and the tests use live in
 * include/msmc_hip_debug.h, outside the product ABI.  The Python host side (msmc-tts_amd/msmctts_amd) binds these with
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

/* Streams of the library's own (hipStreamNonBlocking), for the side branches of a step: the weight gradients' streams and the
 * resolution discriminators' branch (host side: hip/convnet.py own_streams).  The host framework's stream pool hands the same
 * few streams out again and again; a side stream that IS the stream another part of the step captures on is not a branch.
 * No reference counterpart (the reference runs one stream). */
int msmc_stream_create(msmc_stream* out);
int msmc_stream_destroy(msmc_stream stream);

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

/* The same search -- identical indices, quant and diff, bit for bit -- under the HBM roof for K >= 64
 * (msmc-tts_amd/csrc/vq_shortlist.inc): all K distances approximately on the bf16 matrix cores (two-piece bf16 splits of
 * x and e, three products), a running top-2 per frame, and a rigorous error bound that decides per (frame, head) whether
 * the approximate winner IS the exact kernel's first minimum; where it is not (near-ties, duplicate codewords) the
 * 16-frame tile is searched again with the exact fp32 chain.  The codebook comes as a prepared image
 * (msmc_vq_prepare_shortlist, from embed_t / enorm of msmc_vq_prepare; msmc_vq_shortlist_bytes = its size, 0 when the
 * kernel does not take the shape: d = D/H in {32, 64}, K % 16 == 0, one head's image <= 78 KiB).  slow_count (may be NULL):
 * two counters, incremented once per 16-frame tile and head that took [0] the exact re-rank of the best two candidates,
 * [1] the exact re-search over all K.  Finite inputs; magnitudes above bf16 underflow. */
size_t msmc_vq_shortlist_bytes(int H, int d, int K);
int msmc_vq_prepare_shortlist(const float* embed_t, const float* enorm, void* image, int H, int d, int K,
                              msmc_stream stream);
int msmc_vq_search_shortlist(const float* x, const float* embed_t, const float* enorm, const void* image, float* quant,
                             float* diff, int64_t* ind, unsigned long long* slow_count, int N, int D, int H, int K,
                             msmc_stream stream);

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

/* The two halves of msmc_vq_ema_update, for data-parallel codebook synchronisation (an option the reference does not
 * have: its ranks EMA-update from their local batch, distributed.py:154-204 never touches buffers): _stats writes
 * stats = [H][K][d] per-codeword sums followed by [H][K] counts of THIS rank's valid frames; the host sums `stats` over
 * ranks (one all-reduce) and _apply performs the buffer update from the summed statistics.  _stats + _apply on one rank
 * is bit-identical to msmc_vq_ema_update.  stats: H*K*(D/H + 2) floats (sums, counts, H*K of scratch for _apply). */
int msmc_vq_ema_stats(const float* x, const int64_t* ind, const int64_t* length, float* stats, void* workspace,
                      size_t workspace_bytes, int B, int T, int D, int H, int K, msmc_stream stream);
int msmc_vq_ema_apply(float* stats, float* embed, float* cluster_size, float* embed_avg, int D, int H, int K,
                      float decay, float eps, msmc_stream stream);

/* Backward of (quant, diff) wrt x:  gx = g_quant + g_diff * 2*(x - quant)/H   (straight-through +
 * the un-reduced commitment term, modules.py:59-60).  g_diff may be NULL (treated as zero). */
int msmc_vq_backward(const float* g_quant, const float* g_diff, const float* x, const float* quant,
                     float* gx, int N, int D, int H, msmc_stream stream);

/* ---------------------------------------------------------------------------------------------
 * G1/G2/D1/D2  channels-last implicit-GEMM convolution family on the matrix cores.
 * Replaces the cuDNN/MIOpen convolutions behind
 *   Generator.forward / ResBlock1.forward   reference msmctts/networks/hifigan/generator.py:40-55, common.py:44-51
 *   DiscriminatorP.forward                  reference msmctts/networks/hifigan/discriminator.py:135-154
 *   DiscriminatorR.forward                  reference msmctts/networks/hifigan/discriminator.py:71-76
 * One "gather" kernel serves forward convolutions (dilated / strided / reflect- or zero-padded),
 * transposed convolutions and every data-gradient (as per-phase sub-lattices); one kernel computes
 * weight gradients.  Activations are channels-last [B][H][W][C] (1-D signals: H = 1); dtype 0 = fp32
 * (exact-fp32 MFMA), 1 = bf16 storage with fp32 accumulation.
 * ------------------------------------------------------------------------------------------- */
#define MSMC_CONV_MAX_TAPS 16

typedef struct msmc_conv_desc {
    const void* x;          /* input  [B][Hin][Win][Cin]                                                  */
    const void* w;          /* weight slices [nslice][Cout][Cin], same dtype as x                          */
    const float* bias;      /* [Cout] fp32 or NULL                                                         */
    const void* mask_src;   /* NULL, or [B][Hout][Wout][Cout]: v *= (mask_src > 0 ? 1 : mask_slope)        */
    const void* res;        /* NULL, or [B][Hout][Wout][Cout]: v = v + res                                 */
    const void* res2;       /* NULL, or same shape: v = res2 + v                                           */
    void* out;              /* [B][Hout][Wout][Cout]                                                       */
    int dtype;              /* 0 fp32, 1 bf16                                                              */
    int B, Hin, Win, Cin, Hout, Wout, Cout;
    /* output sub-lattice: (oy, ox) = (oy0 + qy*osy, ox0 + qx*osx), qy < QH, qx < QW                       */
    int QH, QW, oy0, osy, ox0, osx;
    /* input coordinate of tap t at lattice point q: (qy*isy + iy0 + tap_dy[t], qx*isx + ix0 + tap_dx[t])  */
    int isy, isx, iy0, ix0;
    int ntaps;
    int tap_dy[MSMC_CONV_MAX_TAPS], tap_dx[MSMC_CONV_MAX_TAPS], tap_w[MSMC_CONV_MAX_TAPS];
    int pad_mode;           /* 0 = zeros outside the input, 1 = reflect                                    */
    float in_slope;         /* leaky-ReLU slope applied to x on load (1 = identity)                        */
    float mask_slope;
    float out_div;          /* v = v / out_div when != 1                                                   */
    float out_slope;        /* leaky-ReLU slope applied to v last (1 = identity)                           */
    int variant;            /* kernel choice, 0 = library heuristic.  msmc_conv_gather: 1 = first-generation dispatch
                               (pipelined / simple kernel), 2 = second generation with 128-byte channel chunks where
                               they fit, 3 = second generation, 64-byte chunks, 4 / 5 = as 2 / 3 with eight instead of
                               four weight vectors in flight per work-item, 6 / 7 = as 2 / 3 with the whole channel chunk in flight
                               and the next chunk prefetched (small grids; MSMC_E_SHAPE when the chunk exceeds 16 vectors
                               per work-item), 8 = direct (matrix-core-free) kernels for <= 8 -> <= 16 channel layers,
                               C -> 1 and 1 -> C layers (MSMC_E_SHAPE otherwise), 16..23 = third generation (bf16: 64-row wave
                               tiles, LDS-DMA weight stream, one barrier per chunk; 16 + (256- instead of 128-point tiles) +
                               2*(64- instead of 32-column wave tiles) + 4*(128- instead of 64-byte chunks); MSMC_E_SHAPE
                               where a configuration does not apply), 24..31 = 16..23 with the halo tile by LDS-DMA as well (no staging
                               registers; unpadded source-swizzled rows, zero chunk for padding pixels, in-place input activation,
                               0 <= in_slope <= 1), 32 = persistent thin-layer kernel (csrc/gather4.inc: Cin, Cout in {32, 64}, unit strides, zero
                               padding, taps along one axis; weights of all taps resident in LDS, halo tiles by LDS-DMA, epilogue in
                               registers; MSMC_E_SHAPE outside that scope), 33 = 32 with the epilogue of a tile deferred into the next
                               iteration (measured no faster: not a tuner candidate), 34 = 1-tap unit-stride layers as a plain channel GEMM
                               (csrc/gemm1.inc: 128 x 128 tiles, both operands by LDS-DMA in 64-channel chunks, epilogue in
                               registers; bf16 with Cin % 8 == 0, or exact fp32 -- v_mfma_f32_32x32x2_f32, fp32 output -- with Cin % 4 == 0;
                               Cout % 4 == 0; MSMC_E_SHAPE for anything but a kernel-size-1 layer on the identity lattice), 35 = 34 with
                               64 x 128 tiles on eight waves (GEMMs with few pixel rows), 36 / 37 = 34 / 35 for fp32 data times a CONSTANT matrix
                               given as its pre-split bf16 image (w: [Cout][ceil(Cin / 32)][hi 32 | lo 32] bf16, hi = bf16(w), lo = bf16(w - hi);
                               three products w_hi x_hi + w_lo x_hi + w_hi x_lo on v_mfma_f32_32x32x16_bf16 with fp32 accumulation and fp32
                               output, ~2^-16 relative; no epilogue operands: the framed-DFT / mel GEMMs of the bf16 configuration), 40..46 = fifth generation (csrc/gather5.inc, bf16,
                               >= 2 taps, Cin % 64 == 0, Cout % 8 == 0: sixteen waves, stages of (64-channel chunk, tap) with the weight
                               slices in an LDS-DMA ring and the halo tile of a chunk shared by its taps, epilogue in registers with 16-byte
                               stores; tiles 40: 128 x 128, 41: 256 x 128, 42 / 43: 128 x 256, 44: 64 x 256, 45: 64 x 128 with the
                               contraction split over two groups of eight waves (Cin % 128 == 0), 46: 256 x 64, 47: 128 x 128 for one-chunk layers
                               with halo tiles of up to 640 pixels; MSMC_E_SHAPE where a configuration does not apply), 50 = thin-channel
                               kernel (csrc/gather6.inc: Cin in {2, 4, 8, 16, 32, 64}, Cout <= 64, Cin * taps <= 704; B fragments loaded
                               straight from global memory, weights in LDS, no workgroup barriers in the tile loop), 9 = 32-point tiles with the channel
                               chunks split over the four waves (deep reductions on small grids).  msmc_conv_wgrad (bf16): 1 = first,
                               2 = second generation (fp32 atomics), 3 = third (split partials + fixed-order reduce), 4 / 5 / 6 = fourth (the
                               third's result contract; pixel tiles flow through an LDS-DMA ring: 4 = three stages, fragment reads two
                               steps ahead of the MFMAs, 5 = one step ahead, 6 = two stages, one step ahead;
                               MSMC_E_SHAPE outside its scope: unit strides, zero padding, channel counts multiples of 64,
                               taps along one axis), 7 = the third generation's lattice tiles with both operands staged by LDS-DMA into
                               a two-stage ring (csrc/wgrad5.inc: strided, 2-D and reflection-padded layers with channel counts that
                               are multiples of 64; a tuner candidate since round 3), 8 = direct kernel for thin layers (csrc/wgrad6.inc:
                               one lattice point per work-item, a block of <= 128 elements of dW in registers, atomics into the
                               privatised copies as generations 1 / 2; MSMC_E_SHAPE when dW needs more than 96 such blocks).  The host
                               layer times the candidates once per layer shape.       */
    int split_shift;        /* msmc_conv_wgrad: pixel split = model << split_shift (>> when negative).  msmc_conv_gather
                               variants 16..31 and 40..46 only: DIAGNOSTICS mask, 0 in production (1 skip the MFMAs, 2 the weight stream,
                               4 the halo loads, 8 the epilogue: tools/bench_gather3.py ABLATE=...; results are then garbage) */
    int dw_copies;          /* msmc_conv_wgrad: R > 1 = dw is [R][ntaps][Cout][Cin] and db [R][Cout]; workgroup i adds
                               into copy i % R (same-address atomics retire serially; R copies shorten the chain R
                               times); the consumer sums the copies (msmc_wn_backward_multi does)                      */
} msmc_conv_desc;

/* ---------------------------------------------------------------------------------------------
 * Attention core of the FFT blocks (bf16, head size 64): softmax(q k^T * scale + bias) (dropout) v per (batch, head),
 * q / k / v read in place from the fused projection, heads merged on the way out.  Replaces
 *   ScaledDotProductAttention.forward + head split / merge   reference acoustic_models/transformer.py:237-259,296-315
 * qkv [B][T][H][192] bf16 (q | k | v, 64 each); bias [B][Tp] fp32 additive key bias (0 = attend, -inf = padding), Tp = T
 * rounded up to a multiple of 32 with a -inf tail; out [B][T][H*64] bf16; lse [B*H][T] fp32 (log-sum-exp of the scaled,
 * biased scores: the backward pass recomputes the probabilities from it).  Dropout (p_drop) acts on the probabilities
 * after the softmax; masks are a counter hash of (seed word on the device, salt, (b, h, query, key)).
 * ------------------------------------------------------------------------------------------- */
int msmc_attn_fwd(const void* qkv, const float* bias, void* out, float* lse, int B, int T, int H, int Tp, float scale,
                  float p_drop, const long long* seed, long long salt, msmc_stream stream);
/* Backward: dqkv [B][T][H][192] (dq | dk | dv, every element written) from dout [B][T][H*64]; the probabilities are
 * recomputed from lse and the same dropout hash (same seed word and salt as the forward call).  dsum [B*H][T] fp32 scratch
 * (dO . O per query, written by the first of the two launches, read by the second). */
int msmc_attn_bwd(const void* qkv, const float* bias, const void* out, const float* lse, const void* dout, void* dqkv,
                  float* dsum, int B, int T, int H, int Tp, float scale, float p_drop, const long long* seed, long long salt,
                  msmc_stream stream);

/* out[q] = epilogue( sum_t sum_ci w[tap_w[t]][co][ci] * act(x[in(q, t)][ci]) + bias[co] ). */
int msmc_conv_gather(const msmc_conv_desc* desc, msmc_stream stream);
/* n (<= 16) INDEPENDENT convolutions, each with msmc_conv_gather semantics, issued in as few launches as their kernel
 * choices allow: members that resolve to the same second-generation kernel instantiation share one grid (the three
 * parallel ResBlocks of a generator stage; one layer of the five period / six resolution sub-discriminators).  Each
 * alone is a grid of tens to a few hundred workgroups; together they fill the 256 CUs. */
int msmc_conv_gather_group(const msmc_conv_desc* descs, int n, msmc_stream stream);

/* dw[tap_w[t]][co][ci] += sum_{b,q} g[b][out(q)][co] * act(x[b][in(q, t)][ci])   (fp32 atomics; caller zeroes dw).
 * Geometry fields as for the forward convolution it differentiates; desc->x = x, desc->out unused,
 * g has the forward output's shape [B][Hout][Wout][Cout] and dtype; desc->mask_slope is a leaky-ReLU
 * slope applied to g on load (1 = identity).  db (may be NULL): db[co] += sum_{b,q} g[b][out(q)][co]. */
int msmc_conv_wgrad(const msmc_conv_desc* desc, const void* g, float* dw, float* db, msmc_stream stream);
/* n (<= 16) independent weight gradients, msmc_conv_wgrad semantics each (db may be NULL, or db[i] NULL); bf16
 * second-generation members share one grid per (at most six) members. */
int msmc_conv_wgrad_group(const msmc_conv_desc* descs, const void* const* g, float* const* dw, float* const* db, int n,
                          msmc_stream stream);
/* Third generation (desc->variant == 3, bf16): NO atomics.  The pixel reduction is split only as far as it takes to fill
 * the chip; every split stores its partial dW (and db) into its own region of a caller-provided workspace with plain
 * stores, and a second launch adds the regions to dw / db in split order (bit-reproducible).  With a single split the
 * kernel accumulates straight into dw and needs no workspace.  msmc_conv_wgrad_workspace: bytes this descriptor needs
 * (0 for other variants / dtypes); a group needs the sum over its members.  MSMC_E_WORKSPACE when it is too small.
 * Fourth generation (desc->variant 4 / 5 / 6, msmc-tts_amd/csrc/wgrad4.inc): same contract and second stage; the
 * workgroup's pixel tiles (output gradient rows + the x rows all its taps touch) stream global -> LDS through a ring
 * filled by global_load_lds_dwordx4 while the matrix cores work on the previous tile.  Inside msmc_conv_wgrad_group_ws
 * such members join the shared grid as third-generation members.  Channel counts: multiples of 8 (tiles of 64 that end past the
 * channels read zeros).  desc->variant 9 (msmc-tts_amd/csrc/wgrad7.inc): the same scope and contract on 128 x 128 channel tiles,
 * eight waves per workgroup, for layers with >= 128 channels on both sides and enough pixels to fill the chip at that size (the
 * 600 <-> 1536 feed-forward layers of the predictor at B = 64); always a launch of its own inside a group call. */
size_t msmc_conv_wgrad_workspace(const msmc_conv_desc* desc, const void* g);
int msmc_conv_wgrad_ws(const msmc_conv_desc* desc, const void* g, float* dw, float* db, void* workspace,
                       size_t workspace_bytes, msmc_stream stream);
int msmc_conv_wgrad_group_ws(const msmc_conv_desc* descs, const void* const* g, float* const* dw, float* const* db, int n,
                             void* workspace, size_t workspace_bytes, msmc_stream stream);
/* As msmc_conv_wgrad_group_ws; group4 != 0: the fourth-generation members of the call share grids of their own kernel
 * (every member planned for its share of the chip, one second-stage launch for all of them) and the rest is issued as
 * msmc_conv_wgrad_group_ws would; 0 = msmc_conv_wgrad_group_ws. */
int msmc_conv_wgrad_group_ws4(const msmc_conv_desc* descs, const void* const* g, float* const* dw, float* const* db, int n,
                              void* workspace, size_t workspace_bytes, msmc_stream stream, int group4);

/* Deferred second stage.  Between msmc_conv_wgrad_defer_begin(sink, capacity) and msmc_conv_wgrad_defer_end() the
 * msmc_conv_wgrad_ws / _group_ws / _group_ws4 calls OF THE CALLING THREAD run their first stage only and append one record
 * per partial-result set to `sink` (up to `capacity`; sets that do not fit are reduced at once, as without a sink);
 * _defer_end returns the number recorded.  The workspace regions the records point into must stay untouched until
 * msmc_conv_wgrad_reduce_pending has added them up: n records in ceil(n / 16) (+ the two-level members' first level)
 * launches instead of one or two per weight gradient -- the host issues it once per backward pass of a network.  The
 * sums and their order are those of the immediate second stage (bit-identical results). */
typedef struct msmc_wg_pending {
    const float* ws;        /* nsplit partial regions of `stride` floats: dW (n_dw) then db (n_db) */
    float* mid;             /* intermediate regions of a two-level reduction (behind the partials) */
    float* dw;              /* accumulated into */
    float* db;              /* may be NULL */
    long stride, n_dw;
    int n_db, nsplit;
} msmc_wg_pending;
void msmc_conv_wgrad_defer_begin(msmc_wg_pending* sink, int capacity);
int msmc_conv_wgrad_defer_end(void);
int msmc_conv_wgrad_reduce_pending(const msmc_wg_pending* items, int n, msmc_stream stream);

/* Weight-norm (torch weight_norm, dim=0) for MANY convolutions in one launch.
 * Item i: v [A][Bc][T] fp32 contiguous (A = dim 0, the normalised axis; T = taps), g [A] fp32.
 *   prepare : w = v * (g / ||v||) written to up to two kernel layouts
 *             dst_k[tap*s_k[0] + a*s_k[1] + b*s_k[2]]  (dtype: 0 fp32, 1 bf16), inv_norm[a] = 1/||v||
 *   backward: from dw (fp32, layout s_1):  gg[a] = sum(dw*v)*inv_norm,
 *             gv = (g*inv_norm) * (dw - v * sum(dw*v) * inv_norm^2);  gb = db;
 *             dw and db are ZEROED as they are consumed (ready for the next accumulation)
 * ``items`` is a DEVICE array; item i owns blocks [block0, block0 + A) of the grid of total_blocks. */
typedef struct msmc_wn_item {
    const float* v;
    const float* g;         /* NULL: plain (not weight-normalised) layer -- w = v, gv = dW, gg / inv_norm unused */
    void* dst1;
    void* dst2;             /* may be NULL */
    float* inv_norm;        /* [A] */
    const float* dw;        /* backward: gradient wrt w in layout 1 (fp32) */
    float* gv;              /* backward: [A][Bc][T] */
    float* gg;              /* backward: [A] */
    long s1[3];
    long s2[3];
    int A, Bc, T, dtype, block0, nbias;
    float* db;              /* backward: [nbias] bias-gradient accumulator (consumed and zeroed), may be NULL */
    float* gb;              /* backward: [nbias] bias gradient out */
    int copies;             /* backward: dw / db hold this many privatised copies (0 or 1 = one), summed here */
    int tblock0;            /* msmc_wn_prepare_multi_tiled: first tile-block of this item (see there) */
    long dw_copy_stride;    /* elements between copies of dw */
    long db_copy_stride;    /* elements between copies of db */
} msmc_wn_item;

int msmc_wn_prepare_multi(const msmc_wn_item* items, int nitems, int total_blocks, msmc_stream stream);
/* Same result in two launches: the row pass writes layout 1 and inv_norm, then a tiled pass writes layout 2 -- the
 * transpose of the parameter's own order -- through LDS with 64 consecutive elements per store instead of one 2-byte store
 * per cache line.  Item i owns tile-blocks [tblock0, tblock0 + ceil(A/64) * ceil(Bc/16)) of total_tile_blocks (items
 * without dst2 own none); T <= MSMC_CONV_MAX_TAPS. */
int msmc_wn_prepare_multi_tiled(const msmc_wn_item* items, int nitems, int total_blocks, int total_tile_blocks,
                                msmc_stream stream);
/* Round 6: both layouts from ONE read of the parameters.  A norms-only row pass over the weight-normalised rows and a tiled
 * pass in which a workgroup reads a tile of rows x columns x all taps in the parameter's own order and writes it out twice --
 * layout 1 with b fastest, layout 2 with a fastest: both must have unit stride there (s1[2] == 1, s2[1] == 1), dst2 may be NULL.
 * Item i owns tile-blocks [tblock0, tblock0 + msmc_wn_tile_blocks(A, Bc, T)) of total_tile_blocks; ``max_taps`` = the largest T
 * among the items (sizes the tile buffer), <= MSMC_CONV_MAX_TAPS.  Device index maps replace the per-workgroup search of the
 * item table: row_item[r] = item of row r (r as block0 counts rows, total_blocks entries), norm_rows[k] = the k-th row of a
 * weight-normalised item (n_norm_rows entries, 0: no such item -- no row pass), tile_item[t] = item of tile-block t (may be NULL:
 * searched).  Replaces msmc_wn_prepare_multi_tiled's row pass (strided re-read of v, a workgroup per row) and its 64 x 16
 * transposing pass: the autoencoder's 36.7 M weights 91 + 138 us -> profiles/README.md. */
int msmc_wn_tile_blocks(int A, int Bc, int T);
int msmc_wn_prepare_multi_tiles(const msmc_wn_item* items, int nitems, int total_blocks, int total_tile_blocks, int max_taps,
                                const int* row_item, const int* norm_rows, int n_norm_rows, const int* tile_item,
                                msmc_stream stream);
int msmc_wn_backward_multi(const msmc_wn_item* items, int nitems, int total_blocks, msmc_stream stream);
/* accumulate != 0: gv / gg / gb += instead of = (a second backward before the gradients were reset: torch .grad semantics) */
int msmc_wn_backward_multi_acc(const msmc_wn_item* items, int nitems, int total_blocks, int accumulate, msmc_stream stream);
/* the same with the longest normalised row (Bc * T parameters) of the items given: rows of up to 4096 parameters run with 128
 * work-items per row and a row buffer sized to max_row -- up to 16 instead of 6 rows in flight per CU (the pass is a chain of
 * memory round trips per row) */
int msmc_wn_backward_multi_rows(const msmc_wn_item* items, int nitems, int total_blocks, int accumulate, int max_row,
                                msmc_stream stream);

/* Backward of ReflectionPad2d(p) fused with the leaky-ReLU' mask, channels-last:
 *   gx[b][y][x][c] = (sum of gp over the padded positions that reflect onto (y, x)) * (mask_src > 0 ? 1 : slope)
 * gp [B][H+2p][W+2p][C], gx and mask_src [B][H][W][C] (mask_src may be NULL). */
int msmc_reflect_fold(const void* gp, const void* mask_src, void* gx, int B, int H, int W, int C, int p, float slope,
                      int dtype, msmc_stream stream);

/* gx = g * (y > 0 ? 1 : slope): backward of a leaky ReLU from its OUTPUT y (sign-preserving), n elements. */
int msmc_lrelu_bwd(const void* g, const void* y, void* gx, long n, float slope, int dtype, msmc_stream stream);

/* Multi-tensor forms of the two helpers above: n (<= 6) tensors per launch (the same layer of all resolution
 * sub-discriminators), 16-byte channel vectors where every member allows. */
int msmc_reflect_fold_multi(const void* const* gp, const void* const* mask_src, void* const* gx, const int* B, const int* H,
                            const int* W, const int* C, int n, int p, float slope, int dtype, msmc_stream stream);
/* as msmc_reflect_fold_multi with a third input per tensor: gx = fold(gp) * lrelu'(mask_src) + res  (res[k] may be NULL):
 * the gradient of a second consumer of the padded layer's input (a feature-matching tap) without a separate add. */
int msmc_reflect_fold_multi_res(const void* const* gp, const void* const* mask_src, const void* const* res, void* const* gx,
                                const int* B, const int* H, const int* W, const int* C, int n, int p, float slope, int dtype,
                                msmc_stream stream);
/* as msmc_reflect_fold_multi_res with the third input added BEFORE the mask: gx = (fold(gp) + tap) * lrelu'(mask_src).
 * For a padded layer whose input is an ACTIVATED map with a second reader (the resolution discriminators' feature
 * maps, reference msmctts/networks/hifigan/discriminator.py DiscriminatorR.forward: the in-place leaky ReLU makes the
 * stored map the activated one): the producer's leaky-ReLU backward happens here instead of in a launch of its own. */
int msmc_reflect_fold_multi_tap(const void* const* gp, const void* const* mask_src, const void* const* tap, void* const* gx,
                                const int* B, const int* H, const int* W, const int* C, int n, int p, float slope, int dtype,
                                msmc_stream stream);
int msmc_lrelu_bwd_multi(const void* const* g, const void* const* y, void* const* gx, const long* nelem, int n, float slope,
                         int dtype, msmc_stream stream);


/* Column sums: out[c] = sum_rows g[row][c] (bias gradients); g dtype as above, out fp32 (overwritten). */
int msmc_colsum(const void* g, float* out, long rows, int C, int dtype, msmc_stream stream);
/* The same without atomics (bit-reproducible): per-block partial sums into ``workspace`` (msmc_colsum_workspace bytes), then a
 * fixed-order sum; accumulate != 0: out += (the bias-gradient accumulators of a bank).  Two launches. */
size_t msmc_colsum_workspace(long rows, int C);
int msmc_colsum_ws(const void* g, float* out, long rows, int C, int dtype, int accumulate, void* workspace,
                   size_t workspace_bytes, msmc_stream stream);

/* ---------------------------------------------------------------------------------------------
 * D3/L1  spectral front-ends as framed-DFT GEMMs on the matrix cores (exact-fp32 MFMA through
 * msmc_conv_gather with one tap) plus three fused element-wise kernels.  Replaces torch.stft (cuFFT/hipFFT)
 * and the surrounding glue of
 *   TorchSTFT.transform / MelScale.forward   reference msmctts/utils/audio.py:398-419, 348-376
 *   MelLoss.mel_spectrogram                  reference msmctts/trainers/criterions/stft_loss.py:76-108
 * All fp32.
 * ------------------------------------------------------------------------------------------- */

/* frames[b][t][j] = x[b][reflect(t*hop + j - pad)] for j < n_fft, 0 for n_fft <= j < NP (row pitch NP).
 * pad = n_fft/2 (centred STFT) or (n_fft-hop)/2 (MelLoss); x [B][L], frames [B][T][NP]. */
int msmc_stft_frames_fwd(const float* x, float* frames, int B, int L, int T, int n_fft, int NP, int hop, int pad,
                         msmc_stream stream);
/* overlap-add adjoint: gx[b][l] = sum of gframes over every (t, j) that read sample l (reflections included). */
int msmc_stft_frames_bwd(const float* gframes, float* gx, int B, int L, int T, int n_fft, int NP, int hop, int pad,
                         msmc_stream stream);

/* spec [R][CP] holds re in columns [0,F) and im in [F,2F) -> mag[r][f] = sqrt(clamp(re^2+im^2, lo)) when
 * clamp_mode = 1 (audio.py:403-404) or sqrt(re^2+im^2+lo) when clamp_mode = 0 (stft_loss.py:104);
 * mag [R][FP], columns F..FP zeroed. */
int msmc_spec_mag_fwd(const float* spec, float* mag, long R, int F, int CP, int FP, float lo, int clamp_mode,
                      msmc_stream stream);
int msmc_spec_mag_bwd(const float* spec, const float* mag, const float* gmag, float* gspec, long R, int F, int CP,
                      int FP, float lo, int clamp_mode, msmc_stream stream);

/* MRD image, channels-last [B][F][T][2] from mel [B*T][FP]: ch0 = mel, ch1 = clamp((20 log10(mel) - ref + 100)/100, 0, 1)
 * (audio.py:411-419 'double' domain, ref_level_db = 20, min_level_db = -100). */
int msmc_mrd_image_fwd(const float* mel, float* img, int B, int T, int F, int FP, msmc_stream stream);
/* the same with the image (and its gradient) in the discriminator stack's compute dtype: 0 fp32, 1 bf16 */
int msmc_mrd_image_fwd_dt(const float* mel, void* img, int B, int T, int F, int FP, int dtype, msmc_stream stream);
int msmc_mrd_image_bwd_dt(const float* mel, const void* gimg, float* gmel, int B, int T, int F, int FP, int dtype,
                          msmc_stream stream);
int msmc_mrd_image_bwd(const float* mel, const float* gimg, float* gmel, int B, int T, int F, int FP,
                       msmc_stream stream);

/* Waveform fan-out of the discriminator (reference msmctts/networks/hifigan/discriminator.py:102-116,135-145,180-190): the period
 * sub-discriminators read the waveform cast to the stack's dtype and reflection-padded on the right to a multiple of their
 * period; the resolution sub-discriminators read it in fp32.  Forward: copies[k] [B][padded_len[k]] (dtype 0 fp32 / 1 bf16),
 * L <= padded_len[k] <= 2L - 1, all n (<= 8) copies in one launch.  Backward: gy [B][L] fp32 = sum of the n32 (<= 8) fp32
 * gradients g32[j] [B][L] and of the copies' gradients folded at the reflected tail (entries may be NULL: no gradient) -- one
 * launch instead of two pad gradients, a cast gradient and ~20 accumulations by the autograd engine. */
int msmc_wave_fan_fwd(const float* y, void* const* copies, const int* padded_len, int n, int B, int L, int dtype,
                      msmc_stream stream);
int msmc_wave_fan_bwd(const float* const* g32, int n32, const void* const* gcopies, const int* padded_len, int n, float* gy,
                      int B, int L, int dtype, msmc_stream stream);
/* Several element-wise stages of the spectral front-ends in ONE launch (the resolution discriminators' five chains advance in lock
 * step: five framings, five magnitudes, five images -- forward and backward).  kind: 0 msmc_stft_frames_fwd (a = x, out = frames),
 * 1 _bwd (a = gframes, out = gx), 2 msmc_spec_mag_fwd (a = spec, out = mag), 3 _bwd (a = spec, b = mag, c = gmag, out = gspec),
 * 4 msmc_mrd_image_fwd_dt (a = mel, out = img), 5 _bwd_dt (a = mel, b = gimg, out = gmel), 6 msmc_log_clamp_fwd (a = x, out = y, R = n),
 * 7 _bwd (a = x, b = g, out = gx, R = n); the remaining fields are the arguments of those entry points. */
#define MSMC_SPECTRAL_MULTI_MAX 8
typedef struct msmc_spectral_op {
    int kind, dtype;
    const void* a;
    const void* b;
    const void* c;
    void* out;
    int B, L, T, n_fft, NP, hop, pad;
    int F, CP, FP, clamp_mode;
    float lo;
    long R;
} msmc_spectral_op;
int msmc_spectral_multi(const msmc_spectral_op* ops, int n, msmc_stream stream);
/* Vocoder windows of a captured step (reference msmctts_trainer.py:211-219, window starts on the device): frames[b][i] = starts[b] + i
 * (int64, i < nframes) and target[b][j] = wav[b][starts[b] * hop + j] (j < nframes * hop; wav [B][L] fp32).  The caller guarantees
 * starts[b] * hop + nframes * hop <= L (VQGANTrainer checks the batch on the host). */
int msmc_window_gather(const long* starts, const float* wav, long* frames, float* target, int B, int nframes, int hop, long L,
                       msmc_stream stream);

/* y = log(max(x, lo)) and its backward gx = g * (x > lo ? 1/x : 0) over n elements (stft_loss.py:110-114). */
int msmc_log_clamp_fwd(const float* x, float* y, long n, float lo, msmc_stream stream);
int msmc_log_clamp_bwd(const float* x, const float* g, float* gx, long n, float lo, msmc_stream stream);

/* ---------------------------------------------------------------------------------------------
 * T2  GAN loss terms over MANY tensors in one launch (feature matching has 55 map pairs, the LSGAN terms
 * 10 score tensors): replaces the per-tensor l1_loss / MSELoss loops of
 *   VQGANTrainer.train_step, reference msmctts/trainers/msmctts_trainer.py:165-171,187-193.
 * Tensors are dense (any permutation of a contiguous layout; a and b share it); dtype 0 fp32 / 1 bf16.
 * ------------------------------------------------------------------------------------------- */
#define MSMC_MAX_TENSORS 64
typedef struct msmc_tensor_table {
    const void* a[MSMC_MAX_TENSORS];
    const void* b[MSMC_MAX_TENSORS];      /* second operand (L1) or unused */
    void* ga[MSMC_MAX_TENSORS];           /* backward: gradient wrt a */
    long n[MSMC_MAX_TENSORS];
    int count, dtype;
} msmc_tensor_table;

/* out[0] = sum_i mean_e |a_i[e] - b_i[e]| */
int msmc_l1_multi_fwd(const msmc_tensor_table* t, float* out, msmc_stream stream);
/* ga_i[e] = gout[0] * sign(a_i[e] - b_i[e]) / n_i */
/* the same forward sums without atomics: per-block partial sums into ``partial`` (msmc_loss_multi_parts() floats), added by one
 * workgroup in a fixed order (tensor by tensor, block by block): bit-reproducible, and no thousands of same-address atomics */
int msmc_loss_multi_parts(void);
int msmc_l1_multi_fwd_ws(const msmc_tensor_table* t, float* partial, float* out, msmc_stream stream);
int msmc_mse_const_multi_fwd_ws(const msmc_tensor_table* t, float target, float* partial, float* out, msmc_stream stream);
int msmc_l1_multi_bwd(const msmc_tensor_table* t, const float* gout, msmc_stream stream);
/* out[0] = sum_i mean_e (a_i[e] - target)^2 ;  ga_i[e] = gout[0] * 2 (a_i[e] - target) / n_i */
int msmc_mse_const_multi_fwd(const msmc_tensor_table* t, float target, float* out, msmc_stream stream);
int msmc_mse_const_multi_bwd(const msmc_tensor_table* t, float target, const float* gout, msmc_stream stream);

/* Weighted sums of loss scalars: out[0] = sum_i weights[i] * terms[i][0] (terms: n <= 64 device pointers to fp32 scalars; weights:
 * n HOST floats, copied into the launch), in term order; backward gvec[i] = gout[0] * weights[i].  Replaces the multiply-and-add
 * chains of 0-dim tensors of reference msmctts_trainer.py:52-62,129-133,160-195 (one launch per sum instead of two per term). */
int msmc_scalar_wsum_fwd(const float* const* terms, const float* weights, int n, float* out, msmc_stream stream);
int msmc_scalar_wsum_bwd(const float* gout, const float* weights, int n, float* gvec, msmc_stream stream);

/* Length-masked means over [B][T][C] tensors (rows t >= lengths[b] are padding): the scalar terms
 *   QuantizerLoss                      reference msmctts/trainers/msmctts_trainer.py:52-62
 *   frame loss                         reference msmctts/trainers/msmctts_trainer.py:129-133
 *   'mse' embedding loss               reference msmctts/networks/vqgantts/msmc_vqgan.py:228-247
 * as two launches forward (partial sums over the valid rows, then a fixed-order sum: bit-reproducible) and one backward,
 * instead of ~10 stock kernels per term.  mode 0: out[0] = sum_valid a / sum_b lengths[b] / C; mode 1: the same over
 * (a - b)^2.  a / b dtype codes 0 fp32, 1 bf16 (mode 0 ignores b); lengths int64 (len_is_64) or int32.  partial: fp32
 * scratch of msmc_masked_mean_parts(B) floats; out: fp32[2] ([1] = 1 / (sum lengths * C), read by the backward pass).
 * Backward: ga = gout[0] * d out[0] / d a (zeros on padding rows), gb = -ga; either may be NULL. */
int msmc_masked_mean_parts(int B);
int msmc_masked_mean_fwd(const void* a, const void* b, const void* lengths, int len_is_64, int B, int T, int C, int a_dtype,
                         int b_dtype, int mode, float* partial, float* out, msmc_stream stream);
int msmc_masked_mean_bwd(const void* a, const void* b, const void* lengths, int len_is_64, int B, int T, int C, int a_dtype,
                         int b_dtype, int mode, const float* out, const float* gout, void* ga, void* gb, msmc_stream stream);

/* Triple (hinge) loss of predictor training against a frozen codebook (BASELINE configuration #4): Quantize.compute_triple_loss,
 * reference msmctts/networks/vqgantts/modules.py:86-116, for all heads of a MultiHeadQuantize (:152-168) in one launch.
 *   p [N][D] fp32 predictions, trg [N][H] int64 target indices, embed_t [H][K][D/H] / enorm [H][K] from msmc_vq_prepare;
 *   lossh [N][H] = reduce_k [t_k != 0] max(t_k + margin, 0) / d,  t_k = sum_c (p_c - e_trg,c)^2 - ((|p|^2 - 2 p.e_k) + |e_k|^2),
 *   reduce = sum (mean == 0) or mean over the K codewords;  gp [N][D] = d lossh[n][h] / d p (the caller scales it by the
 *   incoming gradient and takes the mean over heads).  d = D/H in {16, 32, 64, 128}; (K d + K) floats of LDS. */
int msmc_triple_loss(const float* p, const int64_t* trg, const float* embed_t, const float* enorm, float* lossh, float* gp, int N,
                     int D, int H, int K, float margin, int mean, msmc_stream stream);

/* ---------------------------------------------------------------------------------------------
 * E1/V3  fused element-wise / row-normalisation kernels of the FFT blocks and the quantiser glue (csrc/norm.hip).
 * Replace the stock-kernel chains behind
 *   layer_norm(dropout(h) + residual) [* non_pad_mask]   reference acoustic_models/transformer.py:262-266, 318-323, 352-356
 *   tanh(a) * sigmoid(b) (+ dropout)                     reference vqgantts/modules.py:172-179, 241
 *   Tanh of the 1x1 stacks                               reference vqgantts/msmc_vqgan.py:115-136
 * Rows [N][C] fp32 (dtype 0) / bf16 (dtype 1); statistics and parameter gradients fp32.  Dropout masks are functions of
 * (seed[0] on the device, salt, element index): nothing is stored, both passes regenerate them; seed may be NULL when
 * p_drop == 0.  C <= 1024.
 * ------------------------------------------------------------------------------------------- */
/* v = drop(x) + res (res may be NULL); y = LayerNorm(v) * gamma + beta, rows with keep_row[n] == 0 zeroed (keep_row may be
 * NULL); v, mean [N], rstd [N] are kept for the backward pass. */
int msmc_add_ln_fwd(const void* x, const void* res, const float* gamma, const float* beta, const unsigned char* keep_row,
                    void* y, void* v, float* mean, float* rstd, long N, int C, float eps, float p_drop,
                    const long long* seed, long long salt, int dtype, msmc_stream stream);
/* The attention sub-layer's tail in one launch (bf16): y = layer_norm(dropout(a W^T + bias) + res) * keep_row with the same mask
 * hash, saved tensors (v, mean, rstd) and rounding points as msmc_conv_gather (1 tap) followed by msmc_add_ln_fwd -- reference
 * acoustic_models/transformer.py:259-266.  a [N][K], W [C][K] bf16 (the projection's forward slice), bias fp32 [C] (required),
 * res / y / v [N][C] bf16; K % 32 == 0, C % 4 == 0, C <= 640.  The backward pass is the two-launch chain's (msmc_add_ln_bwd, then the
 * projection's data / weight gradient). */
int msmc_fc_add_ln_fwd(const void* a, const void* W, const float* bias, const void* res, const float* gamma, const float* beta,
                       const unsigned char* keep_row, void* y, void* v, float* mean, float* rstd, long N, int C, int K, float eps,
                       float p_drop, const long long* seed, long long salt, msmc_stream stream);
size_t msmc_add_ln_bwd_workspace(long N, int C);
/* gx = d/dx, gres = d/dres (may be NULL), dgamma / dbeta fp32 [C] (accumulate != 0: +=), fixed reduction order.
 * dgamma == dbeta == NULL: only the per-workgroup partials are produced -- they stay in `workspace`
 * ([ceil(N / 16)][2][C] fp32, which the caller then keeps) for msmc_add_ln_param_multi. */
int msmc_add_ln_bwd(const void* g, const void* v, const float* mean, const float* rstd, const float* gamma,
                    const unsigned char* keep_row, void* gx, void* gres, float* dgamma, float* dbeta, void* workspace,
                    size_t workspace_bytes, long N, int C, float p_drop, const long long* seed, long long salt, int accumulate,
                    int dtype, msmc_stream stream);
/* The parameter gradients of several LayerNorms from the partials their msmc_add_ln_bwd calls left behind, in launches of up
 * to MSMC_LN_PARAM_MAX items (the 24 LayerNorms of the FFT stacks: one launch at the end of the backward pass instead of one
 * 10 us launch behind every LayerNorm backward, on the critical path of the block chain).  `items` is a HOST array (passed to
 * the kernel by value); same fixed reduction order as msmc_add_ln_bwd's own second stage; two items must not name the same
 * dgamma / dbeta.  Reference: the weight / bias gradients autograd derives for nn.LayerNorm in
 * acoustic_models/transformer.py:270-288,330-352. */
#define MSMC_LN_PARAM_MAX 32
typedef struct {
    const float* part;      /* the workspace of the msmc_add_ln_bwd call: [nblocks][2][C] */
    float* dgamma;
    float* dbeta;
    int nblocks;            /* ceil(N / 16) of that call */
    int C;
    int accumulate;         /* != 0: += */
    int reserved;
} msmc_ln_param_item;
int msmc_add_ln_param_multi(const msmc_ln_param_item* items, int nitems, msmc_stream stream);
/* Head of FFTBlocks.forward (reference msmctts/networks/acoustic_models/transformer.py:375-395) with the positions of
 * vqgantts/msmc_vqgan.py:56-58 folded in: out[b][t][:] = seq[b][t][:] + table[t < len[b] ? t + 1 : 0][:] (seq / out
 * [B][T][C] in in_dtype / out_dtype, table fp32 [table_rows][C], T + 1 <= table_rows), keep_row[b T + t] = t < len[b],
 * key_bias (may be NULL) [B][Tp] = 0 on real frames, -inf on padding and on the tail T .. Tp - 1.  lengths: int32 or int64. */
int msmc_fft_prologue(const void* seq, const void* lengths, int len_is_64, const float* table, int table_rows, void* out,
                      unsigned char* keep_row, float* key_bias, int B, int T, int C, int Tp, int in_dtype, int out_dtype,
                      msmc_stream stream);
/* x [N][2C] -> y [N][C] = drop(tanh(x[:, :C]) * sigmoid(x[:, C:])); backward recomputes from x. */
int msmc_gate_fwd(const void* x, void* y, long N, int C, float p_drop, const long long* seed, long long salt, int dtype,
                  msmc_stream stream);
int msmc_gate_bwd(const void* x, const void* g, void* gx, long N, int C, float p_drop, const long long* seed, long long salt,
                  int dtype, msmc_stream stream);
/* Element-wise glue of the quantiser and of the generator's backward pass (round 6; dtype 0 fp32 / 1 bf16, n % 4 == 0, operands
 * 16-byte (fp32) / 8-byte (bf16) aligned).
 *   msmc_sum_n: out = ((a + b) + c) + d, c and d optional -- the input gradients of the parallel ResBlocks (hifigan/generator.py:47-52);
 *   msmc_dropout_add_fwd: y = dropout(x) + res (res optional) with the counter-hash masks of msmc_add_ln_fwd (element index = position),
 *   msmc_dropout_bwd: gx = g * mask * scale -- F.dropout + the residual adds of vqgantts/msmc_vqgan.py:141-176;
 *   msmc_row_mask: keep[b][t] = (t < lengths[b]) as 1 / 0 in ``dtype`` -- ~get_mask_from_lengths (utils/utils.py:9-16) cast to the compute dtype. */
int msmc_sum_n(const void* a, const void* b, const void* c, const void* d, void* out, long n, int dtype, msmc_stream stream);
int msmc_dropout_add_fwd(const void* x, const void* res, void* y, long n, float p_drop, const long long* seed, long long salt, int dtype,
                         msmc_stream stream);
int msmc_dropout_bwd(const void* g, void* gx, long n, float p_drop, const long long* seed, long long salt, int dtype, msmc_stream stream);
int msmc_row_mask(const void* lengths, int lengths_are_int64, void* keep, int B, int T, int dtype, msmc_stream stream);
/* y = tanh(x); gx = g * (1 - y*y) over n elements. */
int msmc_tanh_fwd(const void* x, void* y, long n, int dtype, msmc_stream stream);
int msmc_tanh_bwd(const void* y, const void* g, void* gx, long n, int dtype, msmc_stream stream);
/* y (fp32) = tanh(x) for x in the compute dtype (``dtype``: 0 fp32, 1 bf16) -- the vocoder's output activation, reference
 * hifigan/generator.py:52-54 -- and its backward gx (compute dtype) = g (fp32) * (1 - y^2) */
int msmc_tanh_f32_fwd(const void* x, float* y, long n, int dtype, msmc_stream stream);
int msmc_tanh_f32_bwd(const float* y, const float* g, void* gx, long n, int dtype, msmc_stream stream);

/* ---------------------------------------------------------------------------------------------
 * O1  gradient-norm clipping + AdamW for all tensors of one child in three launches (csrc/optim.hip).
 * Replaces clip_grad_norm_ + the per-child AdamW step of
 *   reference msmctts/trainers/optimizers/__init__.py:53-78, msmctts/trainers/msmctts_trainer.py:205-206.
 * ``table`` is a DEVICE array sorted by first_chunk; tensor i owns workgroups [first_chunk, first_chunk + ceil(n / chunk)).
 * lr [1], step [1] (incremented by the call, as float), norm_coef [2] (out: total gradient norm, clip coefficient) and
 * partial [nblocks] live on the device.  max_norm <= 0: no clipping (coefficient 1).  write_grads != 0: the clipped
 * gradients are written back (clip_grad_norm_'s in-place semantics).  torch.optim.AdamW arithmetic, fp32.
 * ------------------------------------------------------------------------------------------- */
typedef struct msmc_opt_tensor {
    float* p;               /* parameter */
    float* g;               /* gradient */
    float* m;               /* exp_avg */
    float* v;               /* exp_avg_sq */
    long n;                 /* elements */
    int first_chunk;
    int pad_;
} msmc_opt_tensor;
int msmc_opt_chunk(void);   /* elements per workgroup */
int msmc_opt_clip_adamw(const msmc_opt_tensor* table, int ntensors, int nblocks, float max_norm, float* partial,
                        float* norm_coef, const float* lr, float* step, float beta1, float beta2, float eps,
                        float weight_decay, int write_grads, msmc_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MSMC_HIP_H */
