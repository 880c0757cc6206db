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
