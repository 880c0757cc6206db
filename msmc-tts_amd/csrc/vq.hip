// vq.hip -- multi-head nearest-codeword search with EMA codebook update for gfx950.
//
// Replaces Quantize.forward / MultiHeadQuantize.forward
//   (reference msmctts/networks/vqgantts/modules.py:24-67, :137-151; numerical spec SURVEY.md app. B).
//
// msmc_vq_search: one persistent launch for all heads.  Each wave owns 16-frame tiles whose rows
// are streamed HBM -> registers (prefetch, one tile ahead) -> LDS with full-row coalesced
// loads; the transposed codebook of the resident heads sits in LDS; x.e products run on the
// f32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, k-ordered fmaf chain, two independent
// accumulators per wave to cover the 40-cycle dependent latency); the arg-min is a per-lane
// running minimum over the accumulator fragment followed by a two-step wave xor-reduce; the
// gather, straight-through value and squared error are formed in LDS in place and written back
// with full-row stores.  Algorithmic HBM bytes per frame: 4D (x) + 4D (quant) + 8H (ind) + 4D/H (diff).
//
// msmc_vq_ema_update: deterministic two-stage reduction (per-tile counting sort of the indices in
// LDS, ordered per-codeword sums, fixed-order reduction over tiles) then the EMA / Laplace
// smoothing / renormalisation of the three buffers in place.
#include <msmc_rt.hpp>
#include <msmc_hip.h>
#include <msmc_hip_debug.h>

#define VQ_TILE 16
#define VQ_LDS_LIMIT (160 * 1024)

// A/B switch (msmc_vq_set_variant): 1 = register-resident search kernel where d % 16 == 0, 0 = LDS-tile kernel.
static int vq_use_reg_kernel = 1;
extern "C" void msmc_vq_set_variant(int v) { vq_use_reg_kernel = v; }
// symbol of the search kernel the calling thread's most recent msmc_vq_search launched (bench.py's micro-benchmark table)
static thread_local const char* msmc_vq_last = "";
extern "C" const char* msmc_vq_last_kernel(void) { return msmc_vq_last; }

// ------------------------------------------------------------------------------------------------
// prepare: embed [H][d][K] -> embed_t [H][K][d], enorm [H][K]
// ------------------------------------------------------------------------------------------------
// A workgroup owns VQP_K codewords of one head: the d x VQP_K block is read with k fastest (128-byte runs), goes through an
// LDS tile and leaves with j fastest (whole rows of embed_t); the squared norms are summed by one work-item per codeword in
// the order j = 0 .. d-1 (the order of the previous one-work-item-per-codeword form: 33 us for 256 KB -- every store
// instruction of a wave touched 64 cache lines).
#define VQP_K 32
__global__ __launch_bounds__(256) void vq_prepare_kernel(const float* __restrict__ embed, float* __restrict__ embed_t,
                                                        float* __restrict__ enorm, int d, int K) {
    MSMC_DYN_LDS(smem);
    float* tile = (float*)smem;                        // [VQP_K][d + 1]
    const int h = blockIdx.y, k0 = blockIdx.x * VQP_K, pitch = d + 1;
    const float* e = embed + (size_t)h * d * K;
    for (int idx = threadIdx.x; idx < d * VQP_K; idx += 256) {
        const int j = idx / VQP_K, kk = idx - j * VQP_K;
        if (k0 + kk < K) tile[kk * pitch + j] = e[(size_t)j * K + k0 + kk];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < d * VQP_K; idx += 256) {
        const int kk = idx / d, j = idx - kk * d;
        if (k0 + kk < K) embed_t[((size_t)h * K + k0 + kk) * d + j] = tile[kk * pitch + j];
    }
    if (threadIdx.x < VQP_K && k0 + (int)threadIdx.x < K) {
        const float* row = tile + threadIdx.x * pitch;
        float acc = 0.f;
        for (int j = 0; j < d; ++j) {
            const float sq = row[j] * row[j];          // pow(2) then sum(0): square rounded, then added (modules.py:29)
            acc = acc + sq;
        }
        enorm[(size_t)h * K + k0 + threadIdx.x] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// search
// ------------------------------------------------------------------------------------------------
struct VqLds {
    int cb, en, xt, dacc, bidx, total;      // byte offsets
};

static inline VqLds vq_lds_layout(int D, int H, int K, int hpg, int nw) {
    const int d = D / H;
    VqLds L;
    L.cb = 0;
    L.en = L.cb + hpg * K * (d + 4) * 4;
    L.xt = L.en + hpg * K * 4;
    L.dacc = L.xt + nw * VQ_TILE * (D + 4) * 4;
    L.bidx = L.dacc + nw * VQ_TILE * d * 4;
    L.total = L.bidx + nw * VQ_TILE * H * 4;
    return L;
}

template <int NLD>
__global__ __launch_bounds__(256) void vq_search_kernel(const float* __restrict__ x, const float* __restrict__ embed_t,
                                                       const float* __restrict__ enorm, float* __restrict__ quant,
                                                       float* __restrict__ diff, int64_t* __restrict__ ind, int N, int D,
                                                       int H, int K, int hpg, VqLds L) {
    MSMC_DYN_LDS(smem);
    const int d = D / H;
    const int ES = d + 4;
    const int XS = D + 4;
    const int DV = D / 4;
    const int nw = blockDim.x >> 6;
    const int w = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int f = lane & 15;
    const int g = lane >> 4;
    float* cb = (float*)(smem + L.cb);
    float* en = (float*)(smem + L.en);
    float* xt = (float*)(smem + L.xt) + w * VQ_TILE * XS;
    float* dacc = (float*)(smem + L.dacc) + w * VQ_TILE * d;
    int* bidx = (int*)(smem + L.bidx) + w * VQ_TILE * H;

    const int ngroups = (H + hpg - 1) / hpg;
    const int numTiles = (N + VQ_TILE - 1) / VQ_TILE;
    const int numIters = (numTiles + nw - 1) / nw;

    // ---- stage the codebook of heads [h0, h0+cnt) : rows of d floats -> stride ES
    auto stage_group = [&](int h0, int cnt) {
        const int rows = cnt * K;
        const int dv = d >> 2;
        const f32x4* src = (const f32x4*)(embed_t + (size_t)h0 * K * d);
        for (int e = threadIdx.x; e < rows * dv; e += blockDim.x) {
            int r = e / dv, c4 = e - r * dv;
            *(f32x4*)(cb + r * ES + c4 * 4) = src[e];
        }
        for (int e = threadIdx.x; e < rows; e += blockDim.x) en[e] = enorm[(size_t)h0 * K + e];
    };

    if (ngroups == 1) {
        stage_group(0, H);
        __syncthreads();
    }

    f32x4 pre[NLD];
    auto prefetch = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int e = i * 64 + lane;
            int row = e / DV, c4 = e - row * DV;
            int n = tile * VQ_TILE + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < VQ_TILE * DV && n < N) v = *(const f32x4*)(x + (size_t)n * D + c4 * 4);
            pre[i] = v;
        }
    };

    int it = blockIdx.x;
    if (it < numIters && it * nw + w < numTiles) prefetch(it * nw + w);

    for (; it < numIters; it += gridDim.x) {
        const int tile = it * nw + w;
        const bool active = tile < numTiles;
        if (active) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                int e = i * 64 + lane;
                int row = e / DV, c4 = e - row * DV;
                if (e < VQ_TILE * DV) *(f32x4*)(xt + row * XS + c4 * 4) = pre[i];
            }
        }
        wave_sync();                 // tile rows were written by other lanes of this wave
        {   // issue the next tile's loads now; they land while this tile computes
            const int nt = (it + gridDim.x) * nw + w;
            if (it + (int)gridDim.x < numIters && nt < numTiles) prefetch(nt);
        }

        for (int grp = 0; grp < ngroups; ++grp) {
            const int h0 = grp * hpg;
            const int cnt = (H - h0 < hpg) ? (H - h0) : hpg;
            if (ngroups > 1) {
                __syncthreads();
                stage_group(h0, cnt);
                __syncthreads();
            }
            if (!active) continue;
            for (int hl = 0; hl < cnt; ++hl) {
                const int h = h0 + hl;
                const float* cbh = cb + hl * K * ES;
                const float* enh = en + hl * K;
                float* xr = xt + f * XS + h * d;
                // |x|^2 : four interleaved partial sums, combined across the 4 lane groups
                float xx = 0.f;
                for (int j = g; j < d; j += 4) {
                    float v = xr[j];
                    float sq = v * v;
                    xx = xx + sq;
                }
                xx = xx + wave_xor(xx, 16);
                xx = xx + wave_xor(xx, 32);

                float best = __builtin_inff();
                int bi = 0;
                const int ntile = K >> 4;
                int ct = 0;
                for (; ct + 2 <= ntile; ct += 2) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    const float* a0 = cbh + (ct * 16 + f) * ES + g;
                    const float* a1 = a0 + 16 * ES;
                    const float* b = xr + g;
                    for (int s = 0; s < d; s += 4) {
                        float bv = b[s];
                        acc0 = mfma_f32_16x16x4(a0[s], bv, acc0);
                        acc1 = mfma_f32_16x16x4(a1[s], bv, acc1);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int code = ct * 16 + 4 * g + r;
                        float t2 = 2.f * acc0[r];
                        float dist = (xx - t2) + enh[code];
                        if (dist < best) { best = dist; bi = code; }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int code = ct * 16 + 16 + 4 * g + r;
                        float t2 = 2.f * acc1[r];
                        float dist = (xx - t2) + enh[code];
                        if (dist < best) { best = dist; bi = code; }
                    }
                }
                if (ct < ntile) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
                    const float* a0 = cbh + (ct * 16 + f) * ES + g;
                    const float* b = xr + g;
                    for (int s = 0; s < d; s += 4) acc0 = mfma_f32_16x16x4(a0[s], b[s], acc0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int code = ct * 16 + 4 * g + r;
                        float t2 = 2.f * acc0[r];
                        float dist = (xx - t2) + enh[code];
                        if (dist < best) { best = dist; bi = code; }
                    }
                }
                // first-minimum across the four lane groups that share frame f
#pragma unroll
                for (int m = 16; m <= 32; m <<= 1) {
                    float od = wave_xor(best, m);
                    int oi = wave_xor(bi, m);
                    if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
                }
                if (g == 0) bidx[f * H + h] = bi;

                // gather + straight-through value + squared error, in place
                const float* qrow = cbh + bi * ES;
                float* drow = dacc + f * d;
                const int q4 = d >> 2;
                for (int j = g * q4; j < (g + 1) * q4; ++j) {
                    float xv = xr[j];
                    float e = qrow[j] - xv;
                    xr[j] = xv + e;
                    float sq = e * e;
                    drow[j] = (h == 0) ? sq : (drow[j] + sq);
                }
            }
        }

        wave_sync();                 // epilogue results (xt, dacc, bidx) are read by other lanes below
        if (active) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                int e = i * 64 + lane;
                int row = e / DV, c4 = e - row * DV;
                int n = tile * VQ_TILE + row;
                if (e < VQ_TILE * DV && n < N)
                    *(f32x4*)(quant + (size_t)n * D + c4 * 4) = *(const f32x4*)(xt + row * XS + c4 * 4);
            }
            const int q4 = d >> 2;
            const float fh = (float)H;
            for (int e = lane; e < VQ_TILE * q4; e += 64) {
                int row = e / q4, c4 = e - row * q4;
                int n = tile * VQ_TILE + row;
                if (n < N) {
                    f32x4 v = *(const f32x4*)(dacc + row * d + c4 * 4);
                    if (H > 1) { v[0] = v[0] / fh; v[1] = v[1] / fh; v[2] = v[2] / fh; v[3] = v[3] / fh; }
                    *(f32x4*)(diff + (size_t)n * d + c4 * 4) = v;
                }
            }
            for (int e = lane; e < VQ_TILE * H; e += 64) {
                int row = e / H;
                int n = tile * VQ_TILE + row;
                if (n < N) ind[(size_t)tile * VQ_TILE * H + e] = (int64_t)bidx[e];
            }
        }
        wave_sync();                 // next iteration overwrites the tile
    }
}

// ------------------------------------------------------------------------------------------------
// search, register-resident variant (d % 16 == 0): no frame tile in LDS.  Lane (f, g) of a wave keeps
// the 4*D4H values x[f][16t + 4g + jj] of the current head in registers (one 16-byte load per t, the
// next head's / tile's loads are issued before the MFMAs of the current one), feeds them as the MFMA B
// operand while the A operand streams from the LDS codebook with 16-byte reads (one per 4 MFMAs), and
// forms quant / diff in registers.  LDS holds only the codebook, so two workgroups share a CU.
// fp32 summation order: channels are visited as (t, jj, g) -> 16t + 4g + jj; oracle/c/vq_oracle.c mirrors it.
// ------------------------------------------------------------------------------------------------
template <int D4H>
__global__ __launch_bounds__(256, 2) void vq_search_reg_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ embed_t,
                                                              const float* __restrict__ enorm,
                                                              float* __restrict__ quant, float* __restrict__ diff,
                                                              int64_t* __restrict__ ind, int N, int D, int H, int K,
                                                              int hpg) {
    MSMC_DYN_LDS(smem);
    const int d = 16 * D4H;
    const int ES = d + 4;
    float* cb = (float*)smem;                        // [hpg*K][ES]
    float* en = cb + (size_t)hpg * K * ES;           // [hpg*K]
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = lane & 15, g = lane >> 4;
    const int ngroups = (H + hpg - 1) / hpg;
    const int numTiles = (N + VQ_TILE - 1) / VQ_TILE;
    const int numIters = (numTiles + nw - 1) / nw;

    auto stage_group = [&](int h0, int cnt) {
        const int rows = cnt * K, dv = d >> 2;
        const f32x4* src = (const f32x4*)(embed_t + (size_t)h0 * K * d);
        for (int e = threadIdx.x; e < rows * dv; e += blockDim.x) {
            int r = e / dv, c4 = e - r * dv;
            *(f32x4*)(cb + (size_t)r * ES + c4 * 4) = src[e];
        }
        for (int e = threadIdx.x; e < rows; e += blockDim.x) en[e] = enorm[(size_t)h0 * K + e];
    };
    if (ngroups == 1) {
        stage_group(0, H);
        __syncthreads();
    }

    f32x4 xb[D4H], xn[D4H], dacc[D4H];
    auto load_frag = [&](f32x4 (&dst)[D4H], int tile, int h) {
        const int n = tile * VQ_TILE + f;
#pragma unroll
        for (int t = 0; t < D4H; ++t) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (tile < numTiles && n < N) v = *(const f32x4*)(x + (size_t)n * D + h * d + 16 * t + 4 * g);
            dst[t] = v;
        }
    };

    int it = blockIdx.x;
    if (it < numIters) load_frag(xn, it * nw + w, 0);
    for (; it < numIters; it += gridDim.x) {
        const int tile = it * nw + w;
        const bool active = tile < numTiles;
        const int n = tile * VQ_TILE + f;
        for (int grp = 0; grp < ngroups; ++grp) {
            const int h0 = grp * hpg;
            const int cnt = (H - h0 < hpg) ? (H - h0) : hpg;
            if (ngroups > 1) {
                __syncthreads();
                stage_group(h0, cnt);
                __syncthreads();
            }
            for (int hl = 0; hl < cnt; ++hl) {
                const int h = h0 + hl;
#pragma unroll
                for (int t = 0; t < D4H; ++t) xb[t] = xn[t];
                // next fragment in flight: next head of this tile, else head 0 of this wave's next tile
                if (h + 1 < H) load_frag(xn, tile, h + 1);
                else if (it + (int)gridDim.x < numIters) load_frag(xn, (it + gridDim.x) * nw + w, 0);
                if (!active) continue;
                const float* cbh = cb + (size_t)hl * K * ES;
                const float* enh = en + hl * K;
                float xx = 0.f;
#pragma unroll
                for (int t = 0; t < D4H; ++t)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        float sq = xb[t][jj] * xb[t][jj];
                        xx = xx + sq;
                    }
                xx = xx + wave_xor(xx, 16);
                xx = xx + wave_xor(xx, 32);

                float best = __builtin_inff();
                int bi = 0;
                const int ntile = K >> 4;
                int ct = 0;
                for (; ct + 2 <= ntile; ct += 2) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    const float* a0 = cbh + (size_t)(ct * 16 + f) * ES + 4 * g;
                    const float* a1 = a0 + 16 * ES;
#pragma unroll
                    for (int t = 0; t < D4H; ++t) {
                        const f32x4 av0 = *(const f32x4*)(a0 + 16 * t);
                        const f32x4 av1 = *(const f32x4*)(a1 + 16 * t);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            acc0 = mfma_f32_16x16x4(av0[jj], xb[t][jj], acc0);
                            acc1 = mfma_f32_16x16x4(av1[jj], xb[t][jj], acc1);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int code = ct * 16 + 4 * g + r;
                        float t2 = 2.f * acc0[r];
                        float dist = (xx - t2) + enh[code];
                        if (dist < best) { best = dist; bi = code; }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int code = ct * 16 + 16 + 4 * g + r;
                        float t2 = 2.f * acc1[r];
                        float dist = (xx - t2) + enh[code];
                        if (dist < best) { best = dist; bi = code; }
                    }
                }
                if (ct < ntile) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
                    const float* a0 = cbh + (size_t)(ct * 16 + f) * ES + 4 * g;
#pragma unroll
                    for (int t = 0; t < D4H; ++t) {
                        const f32x4 av0 = *(const f32x4*)(a0 + 16 * t);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) acc0 = mfma_f32_16x16x4(av0[jj], xb[t][jj], acc0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int code = ct * 16 + 4 * g + r;
                        float t2 = 2.f * acc0[r];
                        float dist = (xx - t2) + enh[code];
                        if (dist < best) { best = dist; bi = code; }
                    }
                }
#pragma unroll
                for (int m = 16; m <= 32; m <<= 1) {
                    float od = wave_xor(best, m);
                    int oi = wave_xor(bi, m);
                    if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
                }
                const bool row_ok = n < N;
                if (g == 0 && row_ok) ind[(size_t)n * H + h] = (int64_t)bi;
                const float* qrow = cbh + (size_t)bi * ES + 4 * g;
#pragma unroll
                for (int t = 0; t < D4H; ++t) {
                    const f32x4 q4 = *(const f32x4*)(qrow + 16 * t);
                    f32x4 o4, s4;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        float e = q4[jj] - xb[t][jj];
                        o4[jj] = xb[t][jj] + e;
                        float sq = e * e;
                        s4[jj] = (h == 0) ? sq : (dacc[t][jj] + sq);
                    }
                    dacc[t] = s4;
                    if (row_ok) *(f32x4*)(quant + (size_t)n * D + h * d + 16 * t + 4 * g) = o4;
                }
            }
        }
        if (active && n < N) {
            const float fh = (float)H;
#pragma unroll
            for (int t = 0; t < D4H; ++t) {
                f32x4 v = dacc[t];
                if (H > 1) { v[0] = v[0] / fh; v[1] = v[1] / fh; v[2] = v[2] / fh; v[3] = v[3] / fh; }
                *(f32x4*)(diff + (size_t)n * d + 16 * t + 4 * g) = v;
            }
        }
    }
}

#include "vq_shortlist.inc"

typedef void (*vq_search_reg_fn)(const float*, const float*, const float*, float*, float*, int64_t*, int, int, int, int,
                                 int);

typedef void (*vq_search_fn)(const float*, const float*, const float*, float*, float*, int64_t*, int, int, int, int,
                             int, VqLds);

// ------------------------------------------------------------------------------------------------
// EMA statistics, stage 1: per (tile, head) ordered per-codeword sums
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_stats_kernel(const float* __restrict__ x, const int64_t* __restrict__ ind,
                                                      const int64_t* __restrict__ length, float* __restrict__ part,
                                                      float* __restrict__ pcnt, int N, int T, int D, int H, int K,
                                                      int TN) {
    MSMC_DYN_LDS(smem);
    int* ids = (int*)smem;              // [TN]
    int* sorted = ids + TN;             // [TN]
    int* cnt = sorted + TN;             // [K]
    int* off = cnt + K;                 // [K + 1]
    const int tile = blockIdx.x, h = blockIdx.y;
    const int d = D / H;
    const int n0 = tile * TN;
    for (int e = threadIdx.x; e < TN; e += blockDim.x) {
        int n = n0 + e;
        int code = -1;
        if (n < N) {
            int b = n / T, t = n - b * T;
            if ((int64_t)t < length[b]) code = (int)ind[(size_t)n * H + h];
        }
        ids[e] = code;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        int c = 0;
        for (int e = 0; e < TN; ++e) c += (ids[e] == k) ? 1 : 0;
        cnt[k] = c;
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= K; k += blockDim.x) {
        int o = 0;
        for (int q = 0; q < k; ++q) o += cnt[q];
        off[k] = o;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        int p = off[k];
        for (int e = 0; e < TN; ++e)
            if (ids[e] == k) sorted[p++] = e;
    }
    __syncthreads();
    // the head's slice of every valid row of the tile -> LDS, all loads in flight at once (the ordered sums below walk the
    // rows of one codeword after another: from global memory that is one exposed load latency per row)
    float* xs = (float*)(ids + ((2 * TN + 2 * K + 1 + 3) & ~3));   // [TN][d], 16-byte aligned
    for (int e = threadIdx.x; e < TN * (d / 4); e += blockDim.x) {
        const int r = e / (d / 4), c4 = e - r * (d / 4);
        if (ids[r] >= 0) *(f32x4*)(xs + (size_t)r * d + 4 * c4) = *(const f32x4*)(x + (size_t)(n0 + r) * D + h * d + 4 * c4);
    }
    __syncthreads();
    // ordered sums: a "slot" of min(d, 64) lanes owns one codeword at a time
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int lanes_per_code = d < 64 ? d : 64;
    const int slots_per_wave = 64 / lanes_per_code;
    const int slot = w * slots_per_wave + lane / lanes_per_code;
    const int nslots = nw * slots_per_wave;
    const int j0 = lane % lanes_per_code;
    const bool lane_on = (lane / lanes_per_code) < slots_per_wave;
    const size_t pbase = ((size_t)tile * H + h) * K;
    for (int k = slot; k < K; k += nslots) {
        if (!lane_on) continue;
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        const int r1 = off[k + 1];
        for (int r = off[k]; r < r1; ++r) {
            const float* row = xs + (size_t)sorted[r] * d + j0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j0 + 64 * q < d) acc[q] = acc[q] + row[64 * q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (j0 + 64 * q < d) part[(pbase + k) * d + j0 + 64 * q] = acc[q];
        if (j0 == 0) pcnt[pbase + k] = (float)cnt[k];
    }
}

// ------------------------------------------------------------------------------------------------
// EMA statistics, stage 2: fixed-order reduction over tiles + buffer update.
//   vq_ema_cs_kernel : new cluster sizes of every (head, codeword) into scratch (cluster_size itself is not written, so
//                      every workgroup of the next launch sees the same values)
//   vq_ema_kernel    : grid (head, slice): each workgroup re-derives n = sum_k cluster size of its head (same summation
//                      order everywhere), updates its slice of embed_avg / embed; slice 0 publishes cluster_size
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_ema_cs_kernel(const float* __restrict__ pcnt, const float* __restrict__ cluster_size,
                                                       float* __restrict__ cs_new, int ntiles, int HK, float decay,
                                                       float omd) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= HK) return;
    float c = 0.f;
    for (int t = 0; t < ntiles; ++t) c = c + pcnt[(size_t)t * HK + e];
    float v = cluster_size[e] * decay;
    cs_new[e] = fmaf(c, omd, v);
}

__global__ __launch_bounds__(256) void vq_ema_kernel(const float* __restrict__ part, const float* __restrict__ cs_new,
                                                    float* __restrict__ embed, float* __restrict__ cluster_size,
                                                    float* __restrict__ embed_avg, int ntiles, int H, int d, int K,
                                                    float decay, float omd, float eps, float keps) {
    MSMC_DYN_LDS(smem);
    float* cs = (float*)smem;           // [K] updated cluster sizes
    float* red = cs + K;                // [256]
    const int h = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float v = cs_new[(size_t)h * K + k];
        cs[k] = v;
        if (blockIdx.y == 0) cluster_size[(size_t)h * K + k] = v;
    }
    __syncthreads();
    float p = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) p = p + cs[k];
    red[threadIdx.x] = p;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + s];
        __syncthreads();
    }
    const float n = red[0];
    const float den = n + keps;
    const int per = (d * K + gridDim.y - 1) / gridDim.y;
    const int e1 = min(d * K, (int)(blockIdx.y + 1) * per);
    for (int e = blockIdx.y * per + threadIdx.x; e < e1; e += blockDim.x) {
        // (k fastest; j fastest -- coalesced partial-sum reads, strided embed / embed_avg accesses -- measured slower: 42
        //  against 35 us per call, round 4)
        const int j = e / K, k = e - j * K;
        float s = 0.f;
        for (int t = 0; t < ntiles; ++t) s = s + part[(((size_t)t * H + h) * K + k) * d + j];
        const size_t o = ((size_t)h * d + j) * K + k;
        float a = embed_avg[o] * decay;
        a = fmaf(s, omd, a);
        embed_avg[o] = a;
        float sm = (cs[k] + eps) / den * n;
        embed[o] = a / sm;
    }
}

// ------------------------------------------------------------------------------------------------
// EMA statistics alone (data-parallel codebook synchronisation): fixed-order reduction over tiles into
// stats = [H][K][d] sums followed by [H][K] counts -- the layout vq_ema_kernel reads with ntiles = 1
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_stats_reduce_kernel(const float* __restrict__ part, const float* __restrict__ pcnt,
                                                             float* __restrict__ stats, int ntiles, long nsum, long ncnt) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < nsum + ncnt; e += (long)gridDim.x * blockDim.x) {
        const float* src = e < nsum ? part + e : pcnt + (e - nsum);
        const long stride = e < nsum ? nsum : ncnt;
        float s = 0.f;
        for (int t = 0; t < ntiles; ++t) s = s + src[(size_t)t * stride];
        stats[e] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_backward_kernel(const float* __restrict__ gq, const float* __restrict__ gd,
                                                         const float* __restrict__ x, const float* __restrict__ q,
                                                         float* __restrict__ gx, long total4, int D, int d, float invH) {
    const int DV = D >> 2;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (long)gridDim.x * blockDim.x) {
        long n = e / DV;
        int c = (int)(e - n * DV) * 4;
        f32x4 g = *(const f32x4*)(gq + e * 4);
        if (gd) {
            f32x4 xv = *(const f32x4*)(x + e * 4);
            f32x4 qv = *(const f32x4*)(q + e * 4);
            f32x4 dv = *(const f32x4*)(gd + n * d + (c % d));
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = g[i] + (dv[i] * invH) * (2.f * (xv[i] - qv[i]));
        }
        *(f32x4*)(gx + e * 4) = g;
    }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int msmc_vq_prepare(const float* embed, float* embed_t, float* enorm, int H, int d, int K, msmc_stream stream) {
    if (H <= 0 || d <= 0 || K <= 0) return MSMC_E_SHAPE;
    dim3 grid((K + VQP_K - 1) / VQP_K, H);
    const size_t lds = (size_t)VQP_K * (d + 1) * sizeof(float);
    int rc = msmc_allow_lds((const void*)vq_prepare_kernel, (int)lds);
    if (rc) return rc;
    MSMC_LAUNCH(vq_prepare_kernel, grid, dim3(256), lds, (msmc_stream_t)stream, embed, embed_t, enorm, d, K);
    return msmc_check_launch();
}

int msmc_vq_search(const float* x, const float* embed_t, const float* enorm, float* quant, float* diff,
                   int64_t* ind, int N, int D, int H, int K, msmc_stream stream) {
    if (N < 0 || H <= 0 || D <= 0 || D % H) return MSMC_E_SHAPE;
    const int d = D / H;
    if (d % 4 || K % 16 || K <= 0) return MSMC_E_SHAPE;
    if (N == 0) return 0;
    if (d % 16 == 0 && vq_use_reg_kernel) {
        const int d4h = d / 16;
        vq_search_reg_fn rf = nullptr;
        if (d4h == 1) rf = vq_search_reg_kernel<1>;
        else if (d4h == 2) rf = vq_search_reg_kernel<2>;
        else if (d4h == 4) rf = vq_search_reg_kernel<4>;
        else if (d4h == 8) rf = vq_search_reg_kernel<8>;
        else if (d4h == 16) rf = vq_search_reg_kernel<16>;
        if (rf) {
            // most resident heads with <= 80 KiB (two workgroups per CU); at least one head must fit 160 KiB
            int hpg = H;
            size_t lds;
            for (;;) {
                lds = ((size_t)hpg * K * (d + 4) + (size_t)hpg * K) * sizeof(float);
                if (lds <= 80 * 1024 || hpg == 1) break;
                hpg = (hpg + 1) / 2;
            }
            if (lds <= VQ_LDS_LIMIT) {
                int rc = msmc_allow_lds((const void*)rf, (int)lds);
                if (rc) return rc;
                const int nw = 4;
                const int numTiles = (N + VQ_TILE - 1) / VQ_TILE;
                const int numIters = (numTiles + nw - 1) / nw;
                const int wgs = (lds <= 80 * 1024 ? 2 : 1) * MSMC_NUM_CU;
                const int grid = numIters < wgs ? numIters : wgs;
                MSMC_LAUNCH(rf, dim3(grid), dim3(64 * nw), lds, (msmc_stream_t)stream, x, embed_t, enorm, quant, diff,
                            ind, N, D, H, K, hpg);
                msmc_vq_last = msmc_prof_name("vq_search_reg_kernel");
                return msmc_check_launch();
            }
        }
    }
    // pick the widest workgroup and the most resident heads that fit LDS
    int nw = 4, hpg = H;
    VqLds L;
    for (;;) {
        L = vq_lds_layout(D, H, K, hpg, nw);
        if (L.total <= VQ_LDS_LIMIT) break;
        if (hpg > 1) { hpg = (hpg + 1) / 2; continue; }
        if (nw > 1) { nw >>= 1; hpg = H; continue; }
        return MSMC_E_SHAPE;
    }
    const int nld = (VQ_TILE * (D / 4) + 63) / 64;
    vq_search_fn fn = nullptr;
    if (nld <= 1) fn = vq_search_kernel<1>;
    else if (nld <= 2) fn = vq_search_kernel<2>;
    else if (nld <= 4) fn = vq_search_kernel<4>;
    else if (nld <= 8) fn = vq_search_kernel<8>;
    else if (nld <= 16) fn = vq_search_kernel<16>;
    else if (nld <= 32) fn = vq_search_kernel<32>;
    else return MSMC_E_SHAPE;
    int rc = msmc_allow_lds((const void*)fn, L.total);
    if (rc) return rc;
    const int numTiles = (N + VQ_TILE - 1) / VQ_TILE;
    const int numIters = (numTiles + nw - 1) / nw;
    const int grid = numIters < MSMC_NUM_CU ? numIters : MSMC_NUM_CU;
    MSMC_LAUNCH(fn, dim3(grid), dim3(64 * nw), (size_t)L.total, (msmc_stream_t)stream, x, embed_t, enorm, quant, diff,
                ind, N, D, H, K, hpg, L);
    msmc_vq_last = msmc_prof_name("vq_search_kernel");
    return msmc_check_launch();
}

// ---- shortlist search (vq_shortlist.inc) ----------------------------------------------------------------------------
size_t msmc_vq_shortlist_bytes(int H, int d, int K) {
    if (H <= 0 || d <= 0 || K <= 0 || !vqs_mode(H, d, K)) return 0;
    return (size_t)H * vqs_blob_bytes(d, K);
}

int msmc_vq_prepare_shortlist(const float* embed_t, const float* enorm, void* image, int H, int d, int K,
                              msmc_stream stream) {
    if (H <= 0 || d <= 0 || K <= 0 || !vqs_mode(H, d, K)) return MSMC_E_SHAPE;
    MSMC_LAUNCH(vq_prepare_sl_kernel, dim3(H), dim3(256), 0, (msmc_stream_t)stream, embed_t, enorm, (char*)image, d, K,
                (int)vqs_blob_bytes(d, K));
    return msmc_check_launch();
}

int msmc_vq_search_shortlist(const float* x, const float* embed_t, const float* enorm, const void* image, float* quant,
                             float* diff, int64_t* ind, unsigned long long* slow_count, int N, int D, int H, int K,
                             msmc_stream stream) {
    if (N < 0 || H <= 0 || D <= 0 || D % H) return MSMC_E_SHAPE;
    const int d = D / H;
    const int mode = vqs_mode(H, d, K);
    if (!mode) return MSMC_E_SHAPE;
    if (N == 0) return 0;
    // two 16-frame column tiles per wave, eight waves per workgroup (one column tile x sixteen waves measured slower:
    // profiles/r03_vq_shortlist.md); the DIAG instantiations carry the ablation mask and the phase timers
    const int nsub = 2, nw = 8;
    if ((double)N * D * 4.0 >= 4294967296.0) return MSMC_E_SHAPE;     // (the kernel addresses frames with 32-bit byte offsets)
    const int blob = (int)vqs_blob_bytes(d, K);
    size_t lds = (size_t)(mode == 1 ? H : 2) * blob;
    // resident images with room to spare: the fp32 codebook rows next to them (winner rows and exact-path rows from LDS)
    int erows_off = 0;
    if (mode == 1 && !vqs_no_lds_rows && lds + (size_t)H * K * d * 4 <= 156 * 1024 && (H * K) % (1024 / (4 * d)) == 0) {
        erows_off = (int)lds;
        lds += (size_t)H * K * d * 4;
    } else if (mode == 2 && d == 64 && !vqs_no_lds_rows && (size_t)blob + (size_t)K * d * 4 <= 156 * 1024) {
        // one head at a time: ONE image buffer + that head's rows, each refilled while the other is in use (vq_shortlist.inc).
        // Measured on one box against two image buffers with the rows from L2 (tools/bench_vq.py, ABLATE=64): 4 x 256 (d = 64)
        // 983 -> 963 us at N = 2^20 and 143 -> 128 us at N = 131 072; 8 x 512 (d = 32) 1591 -> 1622 us and 231 -> 224 us: taken
        // for d = 64 only.  The steps of these shapes are bound by instruction issue, not by the latency this removes.
        erows_off = blob;
        lds = (size_t)blob + (size_t)K * d * 4;
    }
    vq_search_sl_fn fn;
    if (vqs_ablate) fn = d == 64 ? (vq_search_sl_fn)vq_search_sl_kernel<4, 2, 8, true, false> : (vq_search_sl_fn)vq_search_sl_kernel<2, 2, 8, true, false>;
    else if (erows_off) fn = d == 64 ? (vq_search_sl_fn)vq_search_sl_kernel<4, 2, 8, false, true> : (vq_search_sl_fn)vq_search_sl_kernel<2, 2, 8, false, true>;
    else fn = d == 64 ? (vq_search_sl_fn)vq_search_sl_kernel<4, 2, 8, false, false> : (vq_search_sl_fn)vq_search_sl_kernel<2, 2, 8, false, false>;
    if (vqs_ablate && erows_off) {                  // (the diagnostics instantiations read rows from L2)
        lds = (size_t)(mode == 1 ? H : 2) * blob;
        erows_off = 0;
    }
    int rc = msmc_allow_lds((const void*)fn, (int)lds);
    if (rc) return rc;
    int bits = 0;
    while ((1 << bits) < K / 4) ++bits;
    const unsigned int ibmask = (1u << bits) - 1u;
    // |key - u_k| <= c_approx * (|x| max|e| + max|e|^2 / 2): split residuals + dropped lo.lo products (3.02 * 2^-16), fp32
    // accumulation of 3d products onto the initial value (one rounding of <= 2^-23 relative each), the index bits planted
    // in the mantissa (2^(bits - 23)), and the exact chain's own d roundings (d * 2^-24)
    const float c_approx = 3.02f / 65536.f + (float)(3 * d + 1) / 8388608.f + (float)(1u << bits) / 8388608.f +
                           (float)d / 16777216.f;
    const int numTiles = (N + 16 * nsub - 1) / (16 * nsub);
    const int numIters = (numTiles + nw - 1) / nw;
    const int grid = numIters < MSMC_NUM_CU ? numIters : MSMC_NUM_CU;
    MSMC_LAUNCH(fn, dim3(grid), dim3(64 * nw), lds, (msmc_stream_t)stream, x, embed_t, enorm, (const char*)image,
                quant, diff, ind, slow_count, N, D, H, K, mode == 1 ? 1 : 0, blob, ibmask, c_approx, vqs_ablate, erows_off);
    msmc_vq_last = msmc_prof_name("vq_search_sl_kernel");
    return msmc_check_launch();
}

static inline int vq_stats_tile(int N, int d) {
    int cap = 16384 / d;                    // the tile's rows of one head are staged in LDS: TN * d floats <= 64 KB
    if (cap > 4096) cap = 4096;
    int TN = cap < 256 ? cap : 256;
    while (TN < cap && (N + TN - 1) / TN > 64) TN <<= 1;
    return TN;
}

static int vq_ema_launch(const float* part, const float* pcnt, float* cs_new, float* embed, float* cluster_size,
                         float* embed_avg, int ntiles, int H, int d, int K, float decay, float eps, msmc_stream stream) {
    const float omd = (float)(1.0 - (double)decay);
    const float keps = (float)((double)K * (double)eps);
    MSMC_LAUNCH(vq_ema_cs_kernel, dim3((H * K + 255) / 256), dim3(256), 0, (msmc_stream_t)stream, pcnt,
                (const float*)cluster_size, cs_new, ntiles, H * K, decay, omd);
    int rc = msmc_check_launch();
    if (rc) return rc;
    int slices = (d * K + 2047) / 2048;     // ~8 elements per work-item
    if (slices > 64) slices = 64;
    const size_t lds2 = (size_t)(K + 256) * sizeof(float);
    MSMC_LAUNCH(vq_ema_kernel, dim3(H, slices), dim3(256), lds2, (msmc_stream_t)stream, part, (const float*)cs_new, embed,
                cluster_size, embed_avg, ntiles, H, d, K, decay, omd, eps, keps);
    return msmc_check_launch();
}

static int vq_stats_launch(const float* x, const int64_t* ind, const int64_t* length, float* part, float* pcnt, int N,
                           int T, int D, int H, int K, int TN, int ntiles, msmc_stream stream) {
    const int d = D / H;
    const size_t lds1 = (size_t)((2 * TN + 2 * K + 1 + 3) & ~3) * sizeof(int) + (size_t)TN * d * sizeof(float);
    int rc = msmc_allow_lds((const void*)vq_stats_kernel, (int)lds1);
    if (rc) return rc;
    MSMC_LAUNCH(vq_stats_kernel, dim3(ntiles, H), dim3(256), lds1, (msmc_stream_t)stream, x, ind, length, part, pcnt, N,
                T, D, H, K, TN);
    return msmc_check_launch();
}

size_t msmc_vq_ema_workspace(int N, int D, int H, int K) {
    if (N <= 0 || H <= 0 || D % H) return 0;
    const int d = D / H;
    const int TN = vq_stats_tile(N, d);
    const size_t ntiles = (size_t)(N + TN - 1) / TN;
    return (ntiles * H * K * (size_t)(d + 1) + (size_t)H * K) * sizeof(float);
}

int msmc_vq_ema_update(const float* x, const int64_t* ind, const int64_t* length, float* embed, float* cluster_size,
                       float* embed_avg, void* workspace, size_t workspace_bytes, int B, int T, int D, int H, int K,
                       float decay, float eps, msmc_stream stream) {
    const int N = B * T;
    if (N <= 0 || H <= 0 || D % H || K <= 0) return MSMC_E_SHAPE;
    const int d = D / H;
    if (d > 512 || d % 4) return MSMC_E_SHAPE;
    if (workspace_bytes < msmc_vq_ema_workspace(N, D, H, K)) return MSMC_E_WORKSPACE;
    const int TN = vq_stats_tile(N, d);
    const int ntiles = (N + TN - 1) / TN;
    float* part = (float*)workspace;
    float* pcnt = part + (size_t)ntiles * H * K * d;
    float* cs_new = pcnt + (size_t)ntiles * H * K;
    int rc = vq_stats_launch(x, ind, length, part, pcnt, N, T, D, H, K, TN, ntiles, stream);
    if (rc) return rc;
    return vq_ema_launch(part, pcnt, cs_new, embed, cluster_size, embed_avg, ntiles, H, d, K, decay, eps, stream);
}

int msmc_vq_ema_stats(const float* x, const int64_t* ind, const int64_t* length, float* stats, void* workspace,
                      size_t workspace_bytes, int B, int T, int D, int H, int K, msmc_stream stream) {
    const int N = B * T;
    if (N <= 0 || H <= 0 || D % H || K <= 0) return MSMC_E_SHAPE;
    const int d = D / H;
    if (d > 512 || d % 4) return MSMC_E_SHAPE;
    if (workspace_bytes < msmc_vq_ema_workspace(N, D, H, K)) return MSMC_E_WORKSPACE;
    const int TN = vq_stats_tile(N, d);
    const int ntiles = (N + TN - 1) / TN;
    float* part = (float*)workspace;
    float* pcnt = part + (size_t)ntiles * H * K * d;
    int rc = vq_stats_launch(x, ind, length, part, pcnt, N, T, D, H, K, TN, ntiles, stream);
    if (rc) return rc;
    const long nsum = (long)H * K * d, ncnt = (long)H * K;
    long blocks = (nsum + ncnt + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    MSMC_LAUNCH(vq_stats_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream, (const float*)part,
                (const float*)pcnt, stats, ntiles, nsum, ncnt);
    return msmc_check_launch();
}

int msmc_vq_ema_apply(float* stats, float* embed, float* cluster_size, float* embed_avg, int D, int H, int K, float decay,
                      float eps, msmc_stream stream) {
    if (H <= 0 || D % H || K <= 0) return MSMC_E_SHAPE;
    const int d = D / H;
    float* pcnt = stats + (size_t)H * K * d;
    return vq_ema_launch(stats, pcnt, pcnt + (size_t)H * K, embed, cluster_size, embed_avg, 1, H, d, K, decay, eps, stream);
}

int msmc_vq_backward(const float* g_quant, const float* g_diff, const float* x, const float* quant, float* gx, int N,
                     int D, int H, msmc_stream stream) {
    if (N < 0 || H <= 0 || D % H || (D / H) % 4) return MSMC_E_SHAPE;
    if (N == 0) return 0;
    const long total4 = (long)N * (D / 4);
    long blocks = (total4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    MSMC_LAUNCH(vq_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream, g_quant, g_diff, x,
                quant, gx, total4, D, D / H, 1.0f / (float)H);
    return msmc_check_launch();
}

}  // extern "C"
