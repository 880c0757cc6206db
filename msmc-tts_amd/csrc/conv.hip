// conv.hip -- channels-last implicit-GEMM convolution family for gfx950 (matrix cores).
//
// Replaces the convolutions behind HifiGAN's Generator / ResBlock1
// (reference msmctts/networks/hifigan/generator.py:40-55, common.py:44-51) and the MPD / MRD
// discriminators (reference msmctts/networks/hifigan/discriminator.py:71-76, 135-154).
//
// conv_gather_kernel<T, NT>: out[q][co] = epilogue(sum_t sum_ci W[t][co][ci] * act(x[in(q,t)][ci])).
//   * M = 128 lattice points (a TH x TW block of output pixels of one batch item), N = 32*NT output
//     channels, K = Cin x taps.  4 waves, each a 32 x (32*NT) tile of v_mfma_f32_32x32x16_bf16
//     (bf16 storage) or v_mfma_f32_32x32x2_f32 (fp32 storage: exact fp32, k-ordered fmaf chains).
//   * per 64-byte channel chunk the input halo tile [IH*IW pixels][chunk] is staged ONCE in LDS with the
//     padding rule (zero / reflect) and the input leaky-ReLU applied; every tap then reads its
//     A fragment from the same tile at a shifted row -- the channels-last layout makes a tap a pure
//     row offset, so dilation, stride and 2-D kernels cost nothing extra.
//   * the output lattice (oy0 + qy*osy, ...) and the tap table come from the host, so the same kernel
//     is the forward of strided / dilated convolutions, the forward of transposed convolutions
//     (one launch per output phase) and the data-gradient of all of them.
//   * epilogue in registers: + bias, * leaky-ReLU'(mask_src), + res, res2 + ., / out_div, store
//     (64/128 contiguous bytes per row of the accumulator fragment).
// conv_wgrad_kernel<T>: dW[t][co][ci] += sum_{b,q} g[q][co] * act(x[in(q,t)][ci]); the reduction runs
//   over pixels, so both operands are transposed on their way into LDS (4x4 register transposes,
//   8-byte LDS writes) and every tap accumulates into its own fragment of the same wave.
#include <msmc_rt.hpp>
#include <msmc_hip.h>

#define CV_BM 128

template <typename T> struct Elt;
template <> struct Elt<float> {
    static constexpr int VEC = 4;       // elements per 16 bytes
    static constexpr int CK = 16;       // channels per 64-byte chunk
    static MSMC_DEV_INLINE float ld(const float* p) { return *p; }
    static MSMC_DEV_INLINE void st(float* p, float v) { *p = v; }
};
template <> struct Elt<unsigned short> {
    static constexpr int VEC = 8;
    static constexpr int CK = 32;
    static MSMC_DEV_INLINE float ld(const unsigned short* p) { return bf16_bits_to_f32(*p); }
    static MSMC_DEV_INLINE void st(unsigned short* p, float v) { *p = f32_to_bf16_bits(v); }
};

MSMC_DEV int reflect_index(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    if (i < 0) i = 0;
    if (i >= n) i = n - 1;
    return i;
}

struct CvGeom {
    int TH, TW, IH, IW, dyMin, dxMin, tilesX, tilesY, xt_elems;
};

// 16 channels of one K-step: A fragment rows = pixels, B fragment rows = output channels.
template <int NT>
MSMC_DEV void mma_chunk16(const float* ap, const float* bp, int bstride32, int g, f32x16 (&acc)[NT]) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        f32x4 a4 = *(const f32x4*)(ap + 8 * tt + 4 * g);
        f32x4 b4[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) b4[n] = *(const f32x4*)(bp + n * bstride32 + 8 * tt + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = mfma_f32_32x32x2(a4[e], b4[n][e], acc[n]);
    }
}
template <int NT>
MSMC_DEV void mma_chunk16(const unsigned short* ap, const unsigned short* bp, int bstride32, int g,
                          f32x16 (&acc)[NT]) {
    bf16x8 a = __builtin_bit_cast(bf16x8, *(const u16x8*)(ap + 8 * g));
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        bf16x8 b = __builtin_bit_cast(bf16x8, *(const u16x8*)(bp + n * bstride32 + 8 * g));
        acc[n] = mfma_bf16_32x32x16(a, b, acc[n]);
    }
}

template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_gather_kernel(msmc_conv_desc d, CvGeom G) {
    MSMC_DYN_LDS(smem);
    constexpr int VEC = Elt<T>::VEC, CK = Elt<T>::CK, CKV = CK / VEC, XS = CK + VEC, BN = 32 * NT;
    T* xt = (T*)smem;                       // [IH*IW][XS]
    T* wt = xt + G.xt_elems;                // [ntaps][BN][XS]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane & 31, g = lane >> 5;
    int bt = blockIdx.x;
    const int tx_ = bt % G.tilesX;
    bt /= G.tilesX;
    const int ty_ = bt % G.tilesY;
    const int b = bt / G.tilesY;
    const int co0 = blockIdx.y * BN;
    const int qy0 = ty_ * G.TH, qx0 = tx_ * G.TW;
    const int iyBase = qy0 * d.isy + d.iy0 + G.dyMin, ixBase = qx0 * d.isx + d.ix0 + G.dxMin;
    const int IW = G.IW, npix = G.IH * G.IW;

    int arow;
    {
        int m = 32 * w + i;
        int mty = m / G.TW, mtx = m - mty * G.TW;
        arow = (mty < G.TH) ? (mty * d.isy) * IW + mtx * d.isx : 0;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    const T* xb = (const T*)d.x + (size_t)b * d.Hin * d.Win * d.Cin;
    const T* wg = (const T*)d.w;
    const bool vec_ok = (d.Cin % VEC) == 0;
    const float slope = d.in_slope;

    for (int c0 = 0; c0 < d.Cin; c0 += CK) {
        __syncthreads();
        // ---- input halo tile, padding rule and input activation applied once per element
        for (int e = tid; e < npix * CKV; e += 256) {
            const int pi = e / CKV, v = e - pi * CKV;
            const int ry = pi / IW, rx = pi - ry * IW;
            int iy = iyBase + ry, ix = ixBase + rx;
            bool inside = true;
            if (d.pad_mode == 1) {
                iy = reflect_index(iy, d.Hin);
                ix = reflect_index(ix, d.Win);
            } else {
                inside = (iy >= 0) && (iy < d.Hin) && (ix >= 0) && (ix < d.Win);
            }
            const int c = c0 + v * VEC;
            alignas(16) T vals[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) vals[q] = 0;
            if (inside && c < d.Cin) {
                const T* src = xb + ((size_t)iy * d.Win + ix) * d.Cin + c;
                if (vec_ok) {
                    *(u32x4*)vals = *(const u32x4*)src;
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q)
                        if (c + q < d.Cin) vals[q] = src[q];
                }
                if (slope != 1.f) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        float f = Elt<T>::ld(&vals[q]);
                        f = f > 0.f ? f : f * slope;
                        Elt<T>::st(&vals[q], f);
                    }
                }
            }
            *(u32x4*)(xt + (size_t)pi * XS + v * VEC) = *(const u32x4*)vals;
        }
        // ---- weight slices of every tap for this channel chunk
        for (int e = tid; e < d.ntaps * BN * CKV; e += 256) {
            const int v = e % CKV;
            const int row = e / CKV;                // t * BN + co_l
            const int t = row / BN, col = row - t * BN;
            const int co = co0 + col, c = c0 + v * VEC;
            alignas(16) T vals[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) vals[q] = 0;
            if (co < d.Cout && c < d.Cin) {
                const T* src = wg + ((size_t)d.tap_w[t] * d.Cout + co) * d.Cin + c;
                if (vec_ok) {
                    *(u32x4*)vals = *(const u32x4*)src;
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q)
                        if (c + q < d.Cin) vals[q] = src[q];
                }
            }
            *(u32x4*)(wt + (size_t)row * XS + v * VEC) = *(const u32x4*)vals;
        }
        __syncthreads();
        for (int t = 0; t < d.ntaps; ++t) {
            const T* ap = xt + (size_t)(arow + (d.tap_dy[t] - G.dyMin) * IW + (d.tap_dx[t] - G.dxMin)) * XS;
            const T* bp = wt + (size_t)(t * BN + i) * XS;
#pragma unroll
            for (int ks = 0; ks < CK / 16; ++ks) mma_chunk16<NT>(ap + ks * 16, bp + ks * 16, 32 * XS, g, acc);
        }
    }

    // ---- epilogue: D fragment reg r -> row (r&3) + 8*(r>>2) + 4*g (pixel), col i (output channel)
    const T* mask = (const T*)d.mask_src;
    const T* res = (const T*)d.res;
    const T* res2 = (const T*)d.res2;
    T* out = (T*)d.out;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = co0 + n * 32 + i;
        if (co >= d.Cout) continue;
        const float bv = d.bias ? d.bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * g;
            const int mty = m / G.TW, mtx = m - mty * G.TW;
            const int qy = qy0 + mty, qx = qx0 + mtx;
            if (mty >= G.TH || qy >= d.QH || qx >= d.QW) continue;
            const int oy = d.oy0 + qy * d.osy, ox = d.ox0 + qx * d.osx;
            const size_t o = (((size_t)b * d.Hout + oy) * d.Wout + ox) * d.Cout + co;
            float v = acc[n][r] + bv;
            if (mask) v = v * (Elt<T>::ld(mask + o) > 0.f ? 1.f : d.mask_slope);
            if (res) v = v + Elt<T>::ld(res + o);
            if (res2) v = Elt<T>::ld(res2 + o) + v;
            if (d.out_div != 1.f) v = v / d.out_div;
            if (d.out_slope != 1.f) v = v > 0.f ? v : v * d.out_slope;
            Elt<T>::st(out + o, v);
        }
    }
}

static int cv_geometry(const msmc_conv_desc* d, CvGeom* G, int elt_bytes, int XS, int BN, size_t* lds) {
    if (d->ntaps <= 0 || d->ntaps > MSMC_CONV_MAX_TAPS) return MSMC_E_SHAPE;
    int dyMin = d->tap_dy[0], dyMax = d->tap_dy[0], dxMin = d->tap_dx[0], dxMax = d->tap_dx[0];
    for (int t = 1; t < d->ntaps; ++t) {
        if (d->tap_dy[t] < dyMin) dyMin = d->tap_dy[t];
        if (d->tap_dy[t] > dyMax) dyMax = d->tap_dy[t];
        if (d->tap_dx[t] < dxMin) dxMin = d->tap_dx[t];
        if (d->tap_dx[t] > dxMax) dxMax = d->tap_dx[t];
    }
    int TH, TW;
    if (d->QH == 1) { TH = 1; TW = CV_BM; }
    else if (d->QW <= 16) { TW = d->QW; TH = CV_BM / TW; }
    else { TW = 16; TH = 8; }
    G->TH = TH; G->TW = TW;
    G->dyMin = dyMin; G->dxMin = dxMin;
    G->IH = (TH - 1) * d->isy + (dyMax - dyMin) + 1;
    G->IW = (TW - 1) * d->isx + (dxMax - dxMin) + 1;
    G->tilesY = (d->QH + TH - 1) / TH;
    G->tilesX = (d->QW + TW - 1) / TW;
    G->xt_elems = G->IH * G->IW * XS;
    *lds = ((size_t)G->xt_elems + (size_t)d->ntaps * BN * XS) * elt_bytes;
    return 0;
}

template <typename T>
static int cv_launch(const msmc_conv_desc* d, msmc_stream stream) {
    constexpr int XS = Elt<T>::CK + Elt<T>::VEC;
    CvGeom G;
    size_t lds;
    int NT = d->Cout > 32 ? 2 : 1;
    int rc = cv_geometry(d, &G, sizeof(T), XS, 32 * NT, &lds);
    if (rc) return rc;
    if (lds > 160 * 1024 && NT == 2) {
        NT = 1;
        rc = cv_geometry(d, &G, sizeof(T), XS, 32, &lds);
        if (rc) return rc;
    }
    if (lds > 160 * 1024) return MSMC_E_SHAPE;
    dim3 grid((unsigned)(G.tilesX * G.tilesY * d->B), (unsigned)((d->Cout + 32 * NT - 1) / (32 * NT)));
    if (NT == 2) {
        rc = msmc_allow_lds((const void*)conv_gather_kernel<T, 2>, (int)lds);
        if (rc) return rc;
        MSMC_LAUNCH((conv_gather_kernel<T, 2>), grid, dim3(256), lds, (msmc_stream_t)stream, *d, G);
    } else {
        rc = msmc_allow_lds((const void*)conv_gather_kernel<T, 1>, (int)lds);
        if (rc) return rc;
        MSMC_LAUNCH((conv_gather_kernel<T, 1>), grid, dim3(256), lds, (msmc_stream_t)stream, *d, G);
    }
    return msmc_check_launch();
}

extern "C" int msmc_conv_gather(const msmc_conv_desc* d, msmc_stream stream) {
    if (!d || d->B <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->QH <= 0 || d->QW <= 0) return MSMC_E_SHAPE;
    if (d->dtype == 0) return cv_launch<float>(d, stream);
    if (d->dtype == 1) return cv_launch<unsigned short>(d, stream);
    return MSMC_E_SHAPE;
}

// ================================================================================================
// weight gradient
// ================================================================================================
template <typename T> struct WgTraits;
template <> struct WgTraits<float> { static constexpr int KP = 64, KSTEP = 2; };
template <> struct WgTraits<unsigned short> { static constexpr int KP = 128, KSTEP = 16; };

struct WgGeom {
    int P, chunksPerItem, totalChunks, chunksPerWg;
};

// Stage a [KP points][64 channels] operand TRANSPOSED into lds[ch][KP (+pad)]: each work-item moves
// 4 points x VEC4 channels through a register transpose (global reads stay 8/16-byte wide along C).
template <typename T, bool IS_X>
MSMC_DEV void wg_stage(T* lds, int LS, const msmc_conv_desc& d, const T* base, int C, int c0, int p0, int P, int tap,
                       float slope, int tid) {
    constexpr int KP = WgTraits<T>::KP;
    // 64 channels = 16 groups of 4; KP points = KP/4 groups of 4
    for (int e = tid; e < 16 * (KP / 4); e += 256) {
        const int cg = e & 15, pg = e >> 4;
        const int c = c0 + cg * 4;
        float v[4][4];                              // [point][channel]
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int p = p0 + pg * 4 + pp;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) v[pp][cc] = 0.f;
            if (p >= P || c >= C) continue;
            const int qy = p / d.QW, qx = p - qy * d.QW;
            size_t off;
            bool inside = true;
            if (IS_X) {
                int iy = qy * d.isy + d.iy0 + d.tap_dy[tap], ix = qx * d.isx + d.ix0 + d.tap_dx[tap];
                if (d.pad_mode == 1) {
                    iy = reflect_index(iy, d.Hin);
                    ix = reflect_index(ix, d.Win);
                } else {
                    inside = (iy >= 0) && (iy < d.Hin) && (ix >= 0) && (ix < d.Win);
                }
                off = ((size_t)iy * d.Win + ix) * C + c;
            } else {
                const int oy = d.oy0 + qy * d.osy, ox = d.ox0 + qx * d.osx;
                off = ((size_t)oy * d.Wout + ox) * C + c;
            }
            if (!inside) continue;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                if (c + cc < C) {
                    float f = Elt<T>::ld(base + off + cc);
                    v[pp][cc] = (slope != 1.f && f <= 0.f) ? f * slope : f;
                }
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            T* dst = lds + (size_t)(cg * 4 + cc) * LS + pg * 4;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) Elt<T>::st(dst + pp, v[pp][cc]);
        }
    }
}

MSMC_DEV f32x16 wg_mma(const float* ap, const float* bp, int g, f32x16 acc) {
    // KP = 64 points: lane group g takes points 8t + 4g + e, both operands alike
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        f32x4 a4 = *(const f32x4*)(ap + 8 * t + 4 * g);
        f32x4 b4 = *(const f32x4*)(bp + 8 * t + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma_f32_32x32x2(a4[e], b4[e], acc);
    }
    return acc;
}
MSMC_DEV f32x16 wg_mma(const unsigned short* ap, const unsigned short* bp, int g, f32x16 acc) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {               // KP = 128 points, 16 per MFMA
        bf16x8 a = __builtin_bit_cast(bf16x8, *(const u16x8*)(ap + 16 * t + 8 * g));
        bf16x8 b = __builtin_bit_cast(bf16x8, *(const u16x8*)(bp + 16 * t + 8 * g));
        acc = mfma_bf16_32x32x16(a, b, acc);
    }
    return acc;
}

template <typename T, int TAPS>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(msmc_conv_desc d, const T* __restrict__ gptr,
                                                        float* __restrict__ dw, WgGeom G) {
    MSMC_DYN_LDS(smem);
    constexpr int KP = WgTraits<T>::KP, LS = KP + Elt<T>::VEC;
    T* gt = (T*)smem;                   // [64 co][LS]
    T* xt = gt + 64 * LS;               // [64 ci][LS]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane & 31, g = lane >> 5;
    const int wm = w >> 1, wn = w & 1;  // wave tile: co rows [32*wm, +32), ci cols [32*wn, +32)
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
    // a wave whose 32x32 tile lies entirely beyond Cout x Cin has nothing to compute (thin layers)
    const bool wave_live = (co0 + 32 * wm < d.Cout) && (ci0 + 32 * wn < d.Cin);
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int ch0 = blockIdx.x * G.chunksPerWg;
    int ch1 = ch0 + G.chunksPerWg;
    if (ch1 > G.totalChunks) ch1 = G.totalChunks;
    for (int ch = ch0; ch < ch1; ++ch) {
        const int b = ch / G.chunksPerItem;
        const int p0 = (ch - b * G.chunksPerItem) * KP;
        const T* gb = gptr + (size_t)b * d.Hout * d.Wout * d.Cout;
        const T* xb = (const T*)d.x + (size_t)b * d.Hin * d.Win * d.Cin;
        __syncthreads();
        wg_stage<T, false>(gt, LS, d, gb, d.Cout, co0, p0, G.P, 0, d.mask_slope, tid);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            if (t < d.ntaps) {
                if (t > 0) __syncthreads();
                wg_stage<T, true>(xt, LS, d, xb, d.Cin, ci0, p0, G.P, t, d.in_slope, tid);
                __syncthreads();
                if (wave_live)
                    acc[t] = wg_mma(gt + (size_t)(32 * wm + i) * LS, xt + (size_t)(32 * wn + i) * LS, g, acc[t]);
            }
        }
    }
    // D fragment: row (co) = 32*wm + (r&3) + 8*(r>>2) + 4*g, col (ci) = 32*wn + i
    const int ci = ci0 + 32 * wn + i;
    if (!wave_live || ci >= d.Cin) return;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        if (t >= d.ntaps) continue;
        float* dst = dw + (size_t)d.tap_w[t] * d.Cout * d.Cin;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (co < d.Cout) atomicAdd(dst + (size_t)co * d.Cin + ci, acc[t][r]);
        }
    }
}

template <typename T>
static int wg_launch(const msmc_conv_desc* d, const void* g, float* dw, msmc_stream stream) {
    constexpr int KP = WgTraits<T>::KP, LS = KP + Elt<T>::VEC;
    WgGeom G;
    G.P = d->QH * d->QW;
    G.chunksPerItem = (G.P + KP - 1) / KP;
    G.totalChunks = G.chunksPerItem * d->B;
    const int tiles = ((d->Cout + 63) / 64) * ((d->Cin + 63) / 64);
    int nsplit = (2 * MSMC_NUM_CU + tiles - 1) / tiles;
    if (nsplit > G.totalChunks) nsplit = G.totalChunks;
    if (nsplit < 1) nsplit = 1;
    G.chunksPerWg = (G.totalChunks + nsplit - 1) / nsplit;
    nsplit = (G.totalChunks + G.chunksPerWg - 1) / G.chunksPerWg;
    dim3 grid((unsigned)nsplit, (unsigned)((d->Cout + 63) / 64), (unsigned)((d->Cin + 63) / 64));
    const size_t lds = (size_t)2 * 64 * LS * sizeof(T);
    const T* gp = (const T*)g;
    if (d->ntaps <= 4) MSMC_LAUNCH((conv_wgrad_kernel<T, 4>), grid, dim3(256), lds, (msmc_stream_t)stream, *d, gp, dw, G);
    else if (d->ntaps <= 8) MSMC_LAUNCH((conv_wgrad_kernel<T, 8>), grid, dim3(256), lds, (msmc_stream_t)stream, *d, gp, dw, G);
    else if (d->ntaps <= 12) MSMC_LAUNCH((conv_wgrad_kernel<T, 12>), grid, dim3(256), lds, (msmc_stream_t)stream, *d, gp, dw, G);
    else MSMC_LAUNCH((conv_wgrad_kernel<T, 16>), grid, dim3(256), lds, (msmc_stream_t)stream, *d, gp, dw, G);
    return msmc_check_launch();
}

extern "C" int msmc_conv_wgrad(const msmc_conv_desc* d, const void* g, float* dw, msmc_stream stream) {
    if (!d || !g || !dw || d->B <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->QH <= 0 || d->QW <= 0) return MSMC_E_SHAPE;
    if (d->ntaps <= 0 || d->ntaps > MSMC_CONV_MAX_TAPS) return MSMC_E_SHAPE;
    if (d->dtype == 0) return wg_launch<float>(d, g, dw, stream);
    if (d->dtype == 1) return wg_launch<unsigned short>(d, g, dw, stream);
    return MSMC_E_SHAPE;
}

// ================================================================================================
// weight norm (multi-tensor) and bias gradient
// ================================================================================================
MSMC_DEV float block_sum(float v, float* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = red[tid] + red[tid + s];
        __syncthreads();
    }
    float r = red[0];
    __syncthreads();
    return r;
}

MSMC_DEV int wn_find(const msmc_wn_item* items, int n, int blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (items[mid].block0 <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

MSMC_DEV void wn_store(void* dst, int dtype, long off, float v) {
    if (dtype == 0) ((float*)dst)[off] = v;
    else ((unsigned short*)dst)[off] = f32_to_bf16_bits(v);
}

__global__ __launch_bounds__(256) void wn_prepare_kernel(const msmc_wn_item* __restrict__ items, int nitems) {
    __shared__ float red[256];
    const msmc_wn_item it = items[wn_find(items, nitems, blockIdx.x)];
    const int a = blockIdx.x - it.block0;
    const int n = it.Bc * it.T;
    const float* v = it.v + (size_t)a * n;
    float ss = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) ss = fmaf(v[e], v[e], ss);
    ss = block_sum(ss, red);
    const float norm = sqrtf(ss);
    const float scale = it.g[a] / norm;
    if (threadIdx.x == 0) it.inv_norm[a] = 1.f / norm;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int b = e / it.T, t = e - b * it.T;
        const float wv = v[e] * scale;
        wn_store(it.dst1, it.dtype, t * it.s1[0] + a * it.s1[1] + b * it.s1[2], wv);
        if (it.dst2) wn_store(it.dst2, it.dtype, t * it.s2[0] + a * it.s2[1] + b * it.s2[2], wv);
    }
}

__global__ __launch_bounds__(256) void wn_backward_kernel(const msmc_wn_item* __restrict__ items, int nitems) {
    __shared__ float red[256];
    const msmc_wn_item it = items[wn_find(items, nitems, blockIdx.x)];
    const int a = blockIdx.x - it.block0;
    const int n = it.Bc * it.T;
    const float* v = it.v + (size_t)a * n;
    float dot = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int b = e / it.T, t = e - b * it.T;
        dot = fmaf(it.dw[t * it.s1[0] + a * it.s1[1] + b * it.s1[2]], v[e], dot);
    }
    dot = block_sum(dot, red);
    const float inv = it.inv_norm[a], gval = it.g[a];
    if (threadIdx.x == 0) it.gg[a] = dot * inv;
    const float k1 = gval * inv, k2 = dot * inv * inv;
    float* gv = it.gv + (size_t)a * n;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int b = e / it.T, t = e - b * it.T;
        gv[e] = k1 * (it.dw[t * it.s1[0] + a * it.s1[1] + b * it.s1[2]] - v[e] * k2);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, float* __restrict__ out, long rows, int C,
                                                    int rows_per_block) {
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (long r = r0; r < r1; ++r) s = s + Elt<T>::ld(g + r * C + c);
        atomicAdd(out + c, s);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void reflect_fold_kernel(const T* __restrict__ gp, const T* __restrict__ mask,
                                                          T* __restrict__ gx, int B, int H, int W, int C, int p,
                                                          float slope, long total) {
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long r = e / C;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        // padded rows that reflect onto y: y + p, plus p - y (top border) and 2(H-1) - y + p (bottom border)
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + p;
        if (y >= 1 && y <= p) ys[ny++] = p - y;
        if (y <= H - 2 && y >= H - 1 - p) ys[ny++] = 2 * (H - 1) - y + p;
        xs[nx++] = x + p;
        if (x >= 1 && x <= p) xs[nx++] = p - x;
        if (x <= W - 2 && x >= W - 1 - p) xs[nx++] = 2 * (W - 1) - x + p;
        float s = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int q = 0; q < nx; ++q) s = s + Elt<T>::ld(gp + (((size_t)b * Hp + ys[a]) * Wp + xs[q]) * C + c);
        if (mask) s = s * (Elt<T>::ld(mask + e) > 0.f ? 1.f : slope);
        Elt<T>::st(gx + e, s);
    }
}

__global__ void zero_kernel(float* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

extern "C" {

int msmc_wn_prepare_multi(const msmc_wn_item* items, int nitems, int total_blocks, msmc_stream stream) {
    if (!items || nitems <= 0 || total_blocks <= 0) return MSMC_E_SHAPE;
    MSMC_LAUNCH(wn_prepare_kernel, dim3(total_blocks), dim3(256), 0, (msmc_stream_t)stream, items, nitems);
    return msmc_check_launch();
}

int msmc_wn_backward_multi(const msmc_wn_item* items, int nitems, int total_blocks, msmc_stream stream) {
    if (!items || nitems <= 0 || total_blocks <= 0) return MSMC_E_SHAPE;
    MSMC_LAUNCH(wn_backward_kernel, dim3(total_blocks), dim3(256), 0, (msmc_stream_t)stream, items, nitems);
    return msmc_check_launch();
}

int msmc_reflect_fold(const void* gp, const void* mask_src, void* gx, int B, int H, int W, int C, int p, float slope,
                      int dtype, msmc_stream stream) {
    if (!gp || !gx || B <= 0 || H <= p || W <= p || C <= 0 || p < 0) return MSMC_E_SHAPE;
    const long total = (long)B * H * W * C;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == 0)
        MSMC_LAUNCH(reflect_fold_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream,
                    (const float*)gp, (const float*)mask_src, (float*)gx, B, H, W, C, p, slope, total);
    else if (dtype == 1)
        MSMC_LAUNCH(reflect_fold_kernel<unsigned short>, dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream,
                    (const unsigned short*)gp, (const unsigned short*)mask_src, (unsigned short*)gx, B, H, W, C, p,
                    slope, total);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}

int msmc_colsum(const void* g, float* out, long rows, int C, int dtype, msmc_stream stream) {
    if (!g || !out || rows <= 0 || C <= 0) return MSMC_E_SHAPE;
    MSMC_LAUNCH(zero_kernel, dim3((C + 255) / 256), dim3(256), 0, (msmc_stream_t)stream, out, C);
    int rpb = 256;
    while ((rows + rpb - 1) / rpb > 4096) rpb <<= 1;
    dim3 grid((unsigned)((rows + rpb - 1) / rpb));
    if (dtype == 0) MSMC_LAUNCH(colsum_kernel<float>, grid, dim3(256), 0, (msmc_stream_t)stream, (const float*)g, out, rows, C, rpb);
    else if (dtype == 1) MSMC_LAUNCH(colsum_kernel<unsigned short>, grid, dim3(256), 0, (msmc_stream_t)stream, (const unsigned short*)g, out, rows, C, rpb);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}

}  // extern "C"
