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
#include <cstdio>
#include <msmc_rt.hpp>
#include <msmc_hip.h>
#include <msmc_hip_debug.h>

#define CV_BM 128

// A/B switch (msmc_conv_set_pipeline): 0 = simple kernel, 1 = pipelined kernel with automatic M-tile width,
// 2 / 4 = pipelined kernel forced to 256- / 512-point M tiles (tests).
static int msmc_conv_pipeline_enabled = 1;
extern "C" void msmc_conv_set_pipeline(int on) { msmc_conv_pipeline_enabled = on; }
static int msmc_conv_narrow_when_small = 1;
extern "C" void msmc_conv_set_narrow(int on) { msmc_conv_narrow_when_small = on; }
// name of the kernel the most recent msmc_conv_gather / msmc_conv_wgrad call of this thread launched (profiling aid:
// bench.py attributes its per-launch HIP-event timings to the same symbols rocprofv3 reports)
static thread_local const char* msmc_conv_last = "";
extern "C" const char* msmc_conv_last_kernel(void) { return msmc_conv_last; }
static thread_local long msmc_conv_launches = 0;          // kernels launched by this thread's gather / wgrad calls
extern "C" long msmc_conv_launch_count(void) { return msmc_conv_launches; }
template <typename T> struct EltName;
template <> struct EltName<float> { static constexpr const char* v = "float"; };
template <> struct EltName<unsigned short> { static constexpr const char* v = "unsigned short"; };
static const char* msmc_kname2(const char* base, const char* elt, int a, int b, int c) {
    static thread_local char buf[96];
    snprintf(buf, sizeof(buf), "%s<%s, %d, %d, %d>", base, elt, a, b, c);
    return buf;
}
static const char* msmc_kname(const char* base, const char* elt, int a, int b) {
    static thread_local char buf[96];
    if (b >= 0 && !elt) snprintf(buf, sizeof(buf), "%s<%d, %d>", base, a, b);
    else if (b >= 0) snprintf(buf, sizeof(buf), "%s<%s, %d, %d>", base, elt, a, b);
    else if (elt) snprintf(buf, sizeof(buf), "%s<%s, %d>", base, elt, a);
    else snprintf(buf, sizeof(buf), "%s<%d>", base, a);
    return buf;
}
#define MSMC_GROUP_LIMIT 16         // members one grouped call may carry (split into launches of <= MSMC_GROUP_MAX)
extern "C" int msmc_conv_gather(const msmc_conv_desc* d, msmc_stream stream);
static int msmc_gather_generation = 2;          // 1 = first-generation forward / data-gradient kernels (A/B tests)
extern "C" void msmc_conv_set_gather_generation(int n) { msmc_gather_generation = n; }
static int msmc_wgrad_generation = 2;           // 1 = first-generation bf16 weight-gradient kernel (A/B tests)
extern "C" void msmc_conv_set_wgrad_generation(int n) { msmc_wgrad_generation = n; }
static int msmc_wgrad_tpw_cap = 5;              // accumulators per wave of the second-generation weight gradient (perf sweeps)
extern "C" void msmc_conv_set_wgrad_tpw(int n) { msmc_wgrad_tpw_cap = n < 1 ? 1 : n > 5 ? 5 : n; }
static int msmc_wgrad_split_override = 0;       // tests / perf sweeps: force the pixel-split factor
extern "C" void msmc_conv_set_wgrad_split(int n) { msmc_wgrad_split_override = n; }

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

// Epilogue of N (4 or 8) consecutive bf16 output channels held as floats: mask (leaky-ReLU derivative from the sign of a packed
// bf16 operand), two residuals, division by out_div, output leaky-ReLU.  ONE wave-uniform branch per optional operand around
// all N values: written inside a per-value loop the options were if-converted -- every value paid ~28 vector instructions, an
// 11-instruction IEEE division included, whether or not the operand was there (SQ counters / ISA, round 4: the epilogue was
// 460 of a thin-layer tile's ~740 vector instructions).  The division is a multiplication by the reciprocal (<= 1 ulp in fp32,
// before the bf16 rounding); the fp32 kernels keep the exact division.
template <int N>
MSMC_DEV void cv_ep(float (&v)[N], const unsigned int* mk, const unsigned int* r1, const unsigned int* r2, bool has_mask,
                    bool has_res, bool has_res2, float mslope, float odiv, float oslope) {
    if (has_mask) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float mj = __uint_as_float((j & 1) ? (mk[j >> 1] & 0xffff0000u) : (mk[j >> 1] << 16));
            v[j] = v[j] * (mj > 0.f ? 1.f : mslope);
        }
    }
    if (has_res) {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = v[j] + __uint_as_float((j & 1) ? (r1[j >> 1] & 0xffff0000u) : (r1[j >> 1] << 16));
    }
    if (has_res2) {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = __uint_as_float((j & 1) ? (r2[j >> 1] & 0xffff0000u) : (r2[j >> 1] << 16)) + v[j];
    }
    if (odiv != 1.f) {
        const float rdiv = 1.f / odiv;
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = v[j] * rdiv;
    }
    if (oslope != 1.f) {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * oslope;
    }
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

// ------------------------------------------------------------------------------------------------
// Pipelined variant: each work-item owns a fixed list of 16-byte staging slots (computed once, so no
// index arithmetic in the K loop); the global loads of channel chunk c+1 are issued into registers
// before the MFMAs of chunk c and written to LDS afterwards, and a wave covers MT 32-row sub-tiles so a
// staged weight chunk is reused MT times (M tile = 128*MT lattice points).
// ------------------------------------------------------------------------------------------------
#define CV_XLD 12
#define CV_WLD 12

template <typename T, int NT, int MT>
__global__ __launch_bounds__(256) void conv_gather_pipe_kernel(msmc_conv_desc d, CvGeom G) {
    MSMC_DYN_LDS(smem);
    constexpr int VEC = Elt<T>::VEC, CK = Elt<T>::CK, CKV = CK / VEC, XS = CK + VEC, BN = 32 * NT;
    T* xt = (T*)smem;
    T* wt = xt + G.xt_elems;
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
    const T* xb = (const T*)d.x + (size_t)b * d.Hin * d.Win * d.Cin;
    const T* wg = (const T*)d.w;
    const float slope = d.in_slope;

    // ---- staging slots: global element offset (without the chunk base) or -1, and LDS element offset
    long xsrc[CV_XLD], wsrc[CV_WLD];
    int xdst[CV_XLD], wdst[CV_WLD], xch[CV_XLD], wch[CV_WLD];
#pragma unroll
    for (int j = 0; j < CV_XLD; ++j) {
        const int e = tid + 256 * j;
        xsrc[j] = -1; xdst[j] = -1; xch[j] = 0;
        if (e < npix * CKV) {
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
            xdst[j] = pi * XS + v * VEC;
            xch[j] = v * VEC;
            if (inside) xsrc[j] = ((long)iy * d.Win + ix) * d.Cin + v * VEC;
        }
    }
#pragma unroll
    for (int j = 0; j < CV_WLD; ++j) {
        const int e = tid + 256 * j;
        wsrc[j] = -1; wdst[j] = -1; wch[j] = 0;
        if (e < d.ntaps * BN * CKV) {
            const int v = e % CKV, row = e / CKV;
            const int t = row / BN, col = row - t * BN;
            wdst[j] = row * XS + v * VEC;
            wch[j] = v * VEC;
            if (co0 + col < d.Cout) wsrc[j] = ((long)d.tap_w[t] * d.Cout + co0 + col) * d.Cin + v * VEC;
        }
    }
    int arow[MT];
#pragma unroll
    for (int sI = 0; sI < MT; ++sI) {
        const int m = 32 * (4 * sI + w) + i;
        const int mty = m / G.TW, mtx = m - mty * G.TW;
        arow[sI] = (mty < G.TH) ? (mty * d.isy) * IW + mtx * d.isx : 0;
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int sI = 0; sI < MT; ++sI)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[sI][n][r] = 0.f;

    u32x4 xreg[CV_XLD], wreg[CV_WLD];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto fetch = [&](int c0) {               // Cin % VEC == 0 is guaranteed by the dispatcher
#pragma unroll
        for (int j = 0; j < CV_XLD; ++j) {
            xreg[j] = zero4;
            if (xsrc[j] >= 0 && c0 + xch[j] < d.Cin) xreg[j] = *(const u32x4*)(xb + xsrc[j] + c0);
        }
#pragma unroll
        for (int j = 0; j < CV_WLD; ++j) {
            wreg[j] = zero4;
            if (wsrc[j] >= 0 && c0 + wch[j] < d.Cin) wreg[j] = *(const u32x4*)(wg + wsrc[j] + c0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < CV_XLD; ++j) {
            if (xdst[j] < 0) continue;
            u32x4 v = xreg[j];
            if (slope != 1.f) {
                alignas(16) T vals[VEC];
                *(u32x4*)vals = v;
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    float f = Elt<T>::ld(&vals[q]);
                    f = f > 0.f ? f : f * slope;
                    Elt<T>::st(&vals[q], f);
                }
                v = *(const u32x4*)vals;
            }
            *(u32x4*)(xt + xdst[j]) = v;
        }
#pragma unroll
        for (int j = 0; j < CV_WLD; ++j)
            if (wdst[j] >= 0) *(u32x4*)(wt + wdst[j]) = wreg[j];
    };

    fetch(0);
    for (int c0 = 0; c0 < d.Cin; c0 += CK) {
        __syncthreads();                     // every wave is done reading the previous chunk
        commit();
        __syncthreads();
        if (c0 + CK < d.Cin) fetch(c0 + CK); // in flight during the MFMAs below
        for (int t = 0; t < d.ntaps; ++t) {
            const int toff = (d.tap_dy[t] - G.dyMin) * IW + (d.tap_dx[t] - G.dxMin);
            const T* bp = wt + (size_t)(t * BN + i) * XS;
#pragma unroll
            for (int sI = 0; sI < MT; ++sI) {
                const T* ap = xt + (size_t)(arow[sI] + toff) * XS;
#pragma unroll
                for (int ks = 0; ks < CK / 16; ++ks) mma_chunk16<NT>(ap + ks * 16, bp + ks * 16, 32 * XS, g, acc[sI]);
            }
        }
    }

    const T* mask = (const T*)d.mask_src;
    const T* res = (const T*)d.res;
    const T* res2 = (const T*)d.res2;
    T* out = (T*)d.out;
#pragma unroll
    for (int sI = 0; sI < MT; ++sI)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = co0 + n * 32 + i;
            if (co >= d.Cout) continue;
            const float bv = d.bias ? d.bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * (4 * sI + w) + (r & 3) + 8 * (r >> 2) + 4 * g;
                const int mty = m / G.TW, mtx = m - mty * G.TW;
                const int qy = qy0 + mty, qx = qx0 + mtx;
                if (mty >= G.TH || qy >= d.QH || qx >= d.QW) continue;
                const int oy = d.oy0 + qy * d.osy, ox = d.ox0 + qx * d.osx;
                const size_t o = (((size_t)b * d.Hout + oy) * d.Wout + ox) * d.Cout + co;
                float v = acc[sI][n][r] + bv;
                if (mask) v = v * (Elt<T>::ld(mask + o) > 0.f ? 1.f : d.mask_slope);
                if (res) v = v + Elt<T>::ld(res + o);
                if (res2) v = Elt<T>::ld(res2 + o) + v;
                if (d.out_div != 1.f) v = v / d.out_div;
                if (d.out_slope != 1.f) v = v > 0.f ? v : v * d.out_slope;
                Elt<T>::st(out + o, v);
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Second-generation gather kernel.  Same tiling and MFMA schedule as conv_gather_kernel; what changed is
// everything around the matrix cores, which is where the time went (the first generation executed
// ~2000-6000 scalar/vector ALU instructions per work-item around 2-16 MFMAs):
//   * pixel -> global-offset tables for the input halo tile and the output tile are built ONCE per workgroup
//     in LDS (padding rule, reflect, lattice stride and tile raggedness folded in); the channel-chunk loop
//     and the epilogue index them instead of dividing;
//   * the accumulators leave through an fp32 LDS tile, so bias / mask / residual / activation / store
//     run on 16-byte vectors of consecutive channels (one load and one store instruction per 8 bf16)
//     instead of one 2-byte access per element.
// ------------------------------------------------------------------------------------------------
template <typename T, int NT, int CKM, int SB>
MSMC_DEV void cv2_body(const msmc_conv_desc& d, const CvGeom& G, const int block_x, const int block_y) {
    MSMC_DYN_LDS(smem);
    // CKM = 2: 128-byte channel chunks (half the load -> LDS -> MFMA round trips of a deep reduction);
    // SB: weight-slice vectors a work-item keeps in flight per batch (4 or 8)
    constexpr int VEC = Elt<T>::VEC, CK = Elt<T>::CK * CKM, CKV = CK / VEC, XS = CK + VEC, BN = 32 * NT, OS = BN + 4;
    constexpr int BNV = BN / VEC;
    const int npix = G.IH * G.IW, IW = G.IW;
    int* in_off = (int*)smem;                                   // [npix] element offset of halo pixel, -1 = zero
    int* out_off = in_off + npix;                               // [128]  pixel index of lattice point, -1 = none
    int* tapw = out_off + 128;                                  // [16]   weight slice of tap t
    char* region = smem + (((size_t)(npix + 128 + 16) * sizeof(int) + 15) & ~(size_t)15);
    T* xt = (T*)region;                                         // [npix][XS]
    T* wt = xt + (size_t)npix * XS;                             // [ntaps][BN][XS]
    float* ot = (float*)region;                                 // [128][OS] epilogue tile (aliases xt / wt)
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane & 31, g = lane >> 5;
    int bt = block_x;
    const int tx_ = bt % G.tilesX;
    bt /= G.tilesX;
    const int ty_ = bt % G.tilesY;
    const int b = bt / G.tilesY;
    const int co0 = block_y * BN;
    const int qy0 = ty_ * G.TH, qx0 = tx_ * G.TW;
    const int iyBase = qy0 * d.isy + d.iy0 + G.dyMin, ixBase = qx0 * d.isx + d.ix0 + G.dxMin;

    for (int pi = tid; pi < npix; pi += 256) {
        const int ry = pi / IW, rx = pi - ry * IW;
        int iy = iyBase + ry, ix = ixBase + rx;
        bool inside = true;
        if (d.pad_mode == 1) {
            iy = reflect_index(iy, d.Hin);
            ix = reflect_index(ix, d.Win);
        } else {
            inside = (iy >= 0) && (iy < d.Hin) && (ix >= 0) && (ix < d.Win);
        }
        in_off[pi] = inside ? (iy * d.Win + ix) * d.Cin : -1;
    }
    if (tid < 128) {
        const int mty = tid / G.TW, mtx = tid - mty * G.TW;
        const int qy = qy0 + mty, qx = qx0 + mtx;
        const bool valid = mty < G.TH && qy < d.QH && qx < d.QW;
        out_off[tid] = valid ? (d.oy0 + qy * d.osy) * d.Wout + (d.ox0 + qx * d.osx) : -1;
    }
#pragma unroll
    for (int t = 0; t < MSMC_CONV_MAX_TAPS; ++t)
        if (tid == 128 + t) tapw[t] = t < d.ntaps ? d.tap_w[t] : 0;

    int arow;
    {
        const int m = 32 * w + i;
        const int mty = m / G.TW, mtx = m - mty * G.TW;
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
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int nxv = npix * CKV, nwv = d.ntaps * BN * CKV;

    // staging helpers: loads of a batch are all issued before its LDS stores (one global-load latency per batch)
    auto load_x = [&](int c0, int e0, u32x4 (&vals)[4], int (&dst)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            dst[u] = -1;
            vals[u] = zero4;
            if (e < nxv) {
                const int pi = e / CKV, v = e - pi * CKV;
                const int off = in_off[pi], c = c0 + v * VEC;
                dst[u] = pi * XS + v * VEC;
                if (off >= 0 && c < d.Cin) {
                    const T* src = xb + off + c;
                    if (vec_ok) {
                        vals[u] = *(const u32x4*)src;
                    } else {
                        alignas(16) T tmp[VEC];
#pragma unroll
                        for (int q = 0; q < VEC; ++q) tmp[q] = (c + q < d.Cin) ? src[q] : (T)0;
                        vals[u] = *(const u32x4*)tmp;
                    }
                }
            }
        }
    };
    auto store_x = [&](u32x4 (&vals)[4], int (&dst)[4]) {   // input activation applied once per element
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (dst[u] < 0) continue;
            if (slope != 1.f) {
                if (sizeof(T) == 2 && slope >= 0.f && slope <= 1.f) {      // packed: max(x, slope x) on bf16 pairs, one conversion per pair
#pragma unroll
                    for (int q = 0; q < 4; ++q) vals[u][q] = bf16x2_leaky(vals[u][q], slope);
                } else {
                    alignas(16) T tmp[VEC];
                    *(u32x4*)tmp = vals[u];
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        float f = Elt<T>::ld(&tmp[q]);
                        f = f > 0.f ? f : f * slope;
                        Elt<T>::st(&tmp[q], f);
                    }
                    vals[u] = *(const u32x4*)tmp;
                }
            }
            *(u32x4*)(xt + dst[u]) = vals[u];
        }
    };
    auto load_w = [&](int c0, int e0, u32x4 (&vals)[SB], int (&dst)[SB]) {
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int e = e0 + 256 * u;
            dst[u] = -1;
            vals[u] = zero4;
            if (e < nwv) {
                const int row = e / CKV, v = e - row * CKV;       // row = t * BN + output channel
                const int t = row / BN, col = row - t * BN;
                const int co = co0 + col, c = c0 + v * VEC;
                dst[u] = row * XS + v * VEC;
                if (co < d.Cout && c < d.Cin) {
                    const T* src = wg + ((size_t)tapw[t] * d.Cout + co) * d.Cin + c;
                    if (vec_ok) {
                        vals[u] = *(const u32x4*)src;
                    } else {
                        alignas(16) T tmp[VEC];
#pragma unroll
                        for (int q = 0; q < VEC; ++q) tmp[q] = (c + q < d.Cin) ? src[q] : (T)0;
                        vals[u] = *(const u32x4*)tmp;
                    }
                }
            }
        }
    };
    auto store_w = [&](u32x4 (&vals)[SB], int (&dst)[SB]) {
#pragma unroll
        for (int u = 0; u < SB; ++u)
            if (dst[u] >= 0) *(u32x4*)(wt + dst[u]) = vals[u];
    };

    auto mma_taps = [&]() {
        for (int t = 0; t < d.ntaps; ++t) {
            const T* ap = xt + (size_t)(arow + (d.tap_dy[t] - G.dyMin) * IW + (d.tap_dx[t] - G.dxMin)) * XS;
            const T* bp = wt + (size_t)(t * BN + i) * XS;
#pragma unroll
            for (int ks = 0; ks < CK / 16; ++ks) mma_chunk16<NT>(ap + ks * 16, bp + ks * 16, 32 * XS, g, acc);
        }
    };
    if constexpr (SB == 16) {
        // Small grids (about one workgroup per CU, nothing to overlap with): the WHOLE channel chunk -- halo tile and
        // weight slices, up to 16 vectors per work-item -- is in flight at once and the loads of chunk c+1 are issued
        // before the MFMAs of chunk c, so a chunk costs one global-load latency instead of one per 4-vector batch.
        // One index space for the staging vectors: [0, nxv) input halo tile, [nxv, nv) weight slices.
        const int nv = nxv + nwv;
        u32x4 regs[16];
        auto fetch = [&](int c0) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = u * 256 + tid;
                regs[u] = zero4;
                if (e >= nv) continue;
                const T* src = nullptr;
                int c;
                if (e < nxv) {
                    const int pi = e / CKV;
                    const int off = in_off[pi];
                    c = c0 + (e - pi * CKV) * VEC;
                    if (off >= 0 && c < d.Cin) src = xb + off + c;
                } else {
                    const int e2 = e - nxv;
                    const int row = e2 / CKV;
                    const int t = row / BN, co = co0 + (row - t * BN);
                    c = c0 + (e2 - row * CKV) * VEC;
                    if (co < d.Cout && c < d.Cin) src = wg + ((size_t)tapw[t] * d.Cout + co) * d.Cin + c;
                }
                if (src) regs[u] = *(const u32x4*)src;            // launcher guarantees Cin % VEC == 0 here
            }
        };
        auto commit = [&]() {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = u * 256 + tid;
                if (e >= nv) continue;
                if (e < nxv) {
                    const int pi = e / CKV;
                    u32x4 v = regs[u];
                    if (slope != 1.f) {
                        alignas(16) T tmp[VEC];
                        *(u32x4*)tmp = v;
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            float f = Elt<T>::ld(&tmp[q]);
                            f = f > 0.f ? f : f * slope;
                            Elt<T>::st(&tmp[q], f);
                        }
                        v = *(const u32x4*)tmp;
                    }
                    *(u32x4*)(xt + (size_t)pi * XS + (e - pi * CKV) * VEC) = v;
                } else {
                    const int e2 = e - nxv;
                    const int row = e2 / CKV;
                    *(u32x4*)(wt + (size_t)row * XS + (e2 - row * CKV) * VEC) = regs[u];
                }
            }
        };
        __syncthreads();                                          // offset tables ready
        fetch(0);
        for (int c0 = 0; c0 < d.Cin; c0 += CK) {
            commit();
            __syncthreads();
            if (c0 + CK < d.Cin) fetch(c0 + CK);
            mma_taps();
            __syncthreads();
        }
    } else {
    for (int c0 = 0; c0 < d.Cin; c0 += CK) {
        __syncthreads();                      // tables ready (first pass) / previous chunk's fragments consumed
        {
            // first batch of the halo tile and of the weight slices share one global-load latency
            u32x4 xv[4], wv[SB];
            int xd[4], wd[SB];
            load_x(c0, tid, xv, xd);
            load_w(c0, tid, wv, wd);
            store_x(xv, xd);
            store_w(wv, wd);
        }
        for (int e0 = tid + 1024; e0 < nxv; e0 += 1024) {
            u32x4 xv[4];
            int xd[4];
            load_x(c0, e0, xv, xd);
            store_x(xv, xd);
        }
        for (int e0 = tid + 256 * SB; e0 < nwv; e0 += 256 * SB) {
            u32x4 wv[SB];
            int wd[SB];
            load_w(c0, e0, wv, wd);
            store_w(wv, wd);
        }
        __syncthreads();
        mma_taps();
    }
    }

    // ---- epilogue: accumulators -> fp32 LDS tile -> vectors of VEC consecutive output channels
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            ot[(32 * w + (r & 3) + 8 * (r >> 2) + 4 * g) * OS + n * 32 + i] = acc[n][r];
    __syncthreads();
    const T* mask = (const T*)d.mask_src;
    const T* res = (const T*)d.res;
    const T* res2 = (const T*)d.res2;
    T* out = (T*)d.out;
    const bool ovec = (d.Cout % VEC) == 0;
    const size_t img = (size_t)b * d.Hout * d.Wout;
    for (int e = tid; e < 128 * BNV; e += 256) {
        const int m = e / BNV, vcol = (e - m * BNV) * VEC;
        const int po = out_off[m], co = co0 + vcol;
        if (po < 0 || co >= d.Cout) continue;
        const size_t o = (img + po) * d.Cout + co;
        float v[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) v[q] = ot[m * OS + vcol + q] + ((d.bias && co + q < d.Cout) ? d.bias[co + q] : 0.f);
        alignas(16) T mk[VEC], r1[VEC], r2[VEC], ov[VEC];
        if (ovec) {
            if (mask) *(u32x4*)mk = *(const u32x4*)(mask + o);
            if (res) *(u32x4*)r1 = *(const u32x4*)(res + o);
            if (res2) *(u32x4*)r2 = *(const u32x4*)(res2 + o);
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const bool in = co + q < d.Cout;
                if (mask) mk[q] = in ? mask[o + q] : (T)0;
                if (res) r1[q] = in ? res[o + q] : (T)0;
                if (res2) r2[q] = in ? res2[o + q] : (T)0;
            }
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float x = v[q];
            if (mask) x = x * (Elt<T>::ld(&mk[q]) > 0.f ? 1.f : d.mask_slope);
            if (res) x = x + Elt<T>::ld(&r1[q]);
            if (res2) x = Elt<T>::ld(&r2[q]) + x;
            if (d.out_div != 1.f) x = x / d.out_div;
            if (d.out_slope != 1.f) x = x > 0.f ? x : x * d.out_slope;
            Elt<T>::st(&ov[q], x);
        }
        if (ovec) {
            *(u32x4*)(out + o) = *(const u32x4*)ov;
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q)
                if (co + q < d.Cout) out[o + q] = ov[q];
        }
    }
}


template <typename T, int NT, int CKM, int SB>
__global__ __launch_bounds__(256, 2) void conv_gather2_kernel(msmc_conv_desc d, CvGeom G) {
    cv2_body<T, NT, CKM, SB>(d, G, blockIdx.x, blockIdx.y);
}

// Grouped launch: up to MSMC_GROUP_MAX independent convolutions (same kernel instantiation, any geometry) share one
// grid -- the three parallel ResBlocks of a generator stage, the same layer of the five period / six resolution
// sub-discriminators.  Each of them alone is a small grid (tens to a few hundred workgroups) that leaves most of
// the 256 CUs idle at its head and tail; hipGraph branches do not overlap on this stack, one grid does.
#define MSMC_GROUP_MAX 6
struct CvGroupArgs {
    int n;
    int first[MSMC_GROUP_MAX + 1];      // first flattened block of member k (first[n] = total)
    int nx[MSMC_GROUP_MAX];             // blocks along x of member k (flattened id = x + nx * y)
    msmc_conv_desc d[MSMC_GROUP_MAX];
    CvGeom G[MSMC_GROUP_MAX];
};
MSMC_DEV int cv_group_member(const int* first, int n) {
    int k = 0;
    while (k + 1 < n && (int)blockIdx.x >= first[k + 1]) ++k;
    return k;
}
template <typename T, int NT, int CKM, int SB>
__global__ __launch_bounds__(256, 2) void conv_gather2_group_kernel(CvGroupArgs a) {
    const int k = cv_group_member(a.first, a.n);
    const int id = blockIdx.x - a.first[k];
    cv2_body<T, NT, CKM, SB>(a.d[k], a.G[k], id % a.nx[k], id / a.nx[k]);
}

static int cv_geometry(const msmc_conv_desc* d, CvGeom* G, int elt_bytes, int XS, int BN, size_t* lds,
                       int bm = CV_BM) {
    if (d->ntaps <= 0 || d->ntaps > MSMC_CONV_MAX_TAPS) return MSMC_E_SHAPE;
    int dyMin = d->tap_dy[0], dyMax = d->tap_dy[0], dxMin = d->tap_dx[0], dxMax = d->tap_dx[0];
    for (int t = 1; t < d->ntaps; ++t) {
        if (d->tap_dy[t] < dyMin) dyMin = d->tap_dy[t];
        if (d->tap_dy[t] > dyMax) dyMax = d->tap_dy[t];
        if (d->tap_dx[t] < dxMin) dxMin = d->tap_dx[t];
        if (d->tap_dx[t] > dxMax) dxMax = d->tap_dx[t];
    }
    int TH, TW;
    if (d->QH == 1) { TH = 1; TW = bm; }
    else if (d->QW <= 16) { TW = d->QW; TH = bm / TW; if (TH < 1) TH = 1; }
    else { TW = 16; TH = bm / 16; }
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

#include "gather3.inc"
#include "gather4.inc"
#include "gemm1.inc"
#include "gather5.inc"
#include "gather7.inc"
#include "gather6.inc"

template <typename T, int NT, int MT>
static int cv_try_pipe(const msmc_conv_desc* d, msmc_stream stream, bool* done) {
    constexpr int XS = Elt<T>::CK + Elt<T>::VEC, CKV = Elt<T>::CK / Elt<T>::VEC;
    CvGeom G;
    size_t lds;
    *done = false;
    int rc = cv_geometry(d, &G, sizeof(T), XS, 32 * NT, &lds, CV_BM * MT);
    if (rc) return rc;
    if (lds > 160 * 1024) return 0;
    if (G.IH * G.IW * CKV > 256 * CV_XLD || d->ntaps * 32 * NT * CKV > 256 * CV_WLD) return 0;
    dim3 grid((unsigned)(G.tilesX * G.tilesY * d->B), (unsigned)((d->Cout + 32 * NT - 1) / (32 * NT)));
    rc = msmc_allow_lds((const void*)conv_gather_pipe_kernel<T, NT, MT>, (int)lds);
    if (rc) return rc;
    MSMC_LAUNCH((conv_gather_pipe_kernel<T, NT, MT>), grid, dim3(256), lds, (msmc_stream_t)stream, *d, G);
    msmc_conv_last = msmc_prof_name(msmc_kname("conv_gather_pipe_kernel", EltName<T>::v, NT, MT));
    *done = true;
    return msmc_check_launch();
}

// Kernel choice of the second-generation gather for one descriptor (shared by the single and the grouped launch).
struct Cv2Plan {
    int applies;            // 1: second generation; 0: first-generation dispatch
    int nt, ckm, sb;
    CvGeom G;
    size_t lds;
    unsigned gx, gy;
};
template <typename T>
static int cv2_plan(const msmc_conv_desc* d, Cv2Plan* pl, int* narrow_nt) {
    constexpr int XS = Elt<T>::CK + Elt<T>::VEC;
    pl->applies = 0;
    int NT = d->Cout > 32 ? 2 : 1;
    if (NT == 2 && msmc_conv_narrow_when_small) {
        // few output pixels: 32-channel N tiles double the workgroup count (two co-resident workgroups per CU
        // hide each other's global-load latency, which dominates at this size)
        const long mt = ((long)d->QH * d->QW + CV_BM - 1) / CV_BM * d->B;
        const long wide = mt * ((d->Cout + 63) / 64);
        if (wide < MSMC_NUM_CU || (wide < 2 * MSMC_NUM_CU && d->Cin >= 256)) NT = 1;
    }
    *narrow_nt = NT;
    // second generation for shallow reductions (fewer than 4 channel chunks); deep ones keep the register-prefetching
    // pipelined kernel unless a variant says otherwise: measured per layer on MI355X
    const int gen = d->variant > 0 ? d->variant : msmc_gather_generation;
    const bool sb8 = gen == 4 || gen == 5 || (d->variant == 0 && (long)d->ntaps * 32 * NT * (Elt<T>::CK / Elt<T>::VEC) > 1024);
    const bool ck1 = gen == 3 || gen == 5 || gen == 7;
    const bool want16 = gen == 6 || gen == 7;                  // whole chunk in flight + prefetch (small grids)
    bool shallow = d->Cin < 4 * Elt<T>::CK || (d->Cin % Elt<T>::VEC) != 0 || gen >= 3 || d->variant == 2;
    if (want16 && (d->Cin % Elt<T>::VEC) != 0) return MSMC_E_SHAPE;
    if (!shallow && gen == 2) {
        // heuristic without a tuned variant: deep reductions go to the pipelined kernel only where it would use
        // 256-point (or wider) M tiles, i.e. where it amortises the weight staging over a large grid
        const long points = (long)d->B * d->QH * d->QW;
        const long ntile = (d->Cout + 32 * NT - 1) / (32 * NT);
        shallow = (points / (CV_BM * 2)) * ntile < 2 * MSMC_NUM_CU;
    }
    if (!(gen >= 2 && shallow && (long)d->Hin * d->Win * d->Cin < (1L << 31) && (long)d->Hout * d->Wout < (1L << 31)))
        return 0;
    size_t unused;
    int rc = cv_geometry(d, &pl->G, sizeof(T), XS, 32, &unused);
    if (rc) return rc;
    const long npix = (long)pl->G.IH * pl->G.IW;
    const size_t tables = (((size_t)(npix + 128 + 16) * sizeof(int)) + 15) & ~(size_t)15;
    auto lds_of = [&](int nt, int ckm) {
        const size_t xs = (size_t)Elt<T>::CK * ckm + Elt<T>::VEC;
        const size_t stage = ((size_t)npix + (size_t)d->ntaps * 32 * nt) * xs * sizeof(T);
        const size_t epi = (size_t)128 * (32 * nt + 4) * sizeof(float);
        return tables + (stage > epi ? stage : epi);
    };
    int nt = NT;
    int ckm = (d->Cin >= 2 * Elt<T>::CK && (d->Cin % Elt<T>::VEC) == 0 && lds_of(nt, 2) <= 64 * 1024 && !ck1) ? 2 : 1;
    if (lds_of(nt, ckm) > 160 * 1024 && nt == 2) nt = 1;
    const size_t lds2 = lds_of(nt, ckm);
    const long chunk_vectors = (npix + (long)d->ntaps * 32 * nt) * (Elt<T>::CK * ckm / Elt<T>::VEC);
    const bool sb16 = want16 && chunk_vectors <= 16L * 256 && (d->Cin % Elt<T>::VEC) == 0;
    if (want16 && !sb16) return MSMC_E_SHAPE;                  // the tuner skips candidates that do not apply
    if (lds2 > 160 * 1024) return 0;
    pl->applies = 1;
    pl->nt = nt;
    pl->ckm = ckm;
    pl->sb = sb16 ? 16 : sb8 ? 8 : 4;
    pl->lds = lds2;
    pl->gx = (unsigned)(pl->G.tilesX * pl->G.tilesY * d->B);
    pl->gy = (unsigned)((d->Cout + 32 * nt - 1) / (32 * nt));
    return 0;
}

// the same plan with the tile / chunk / batch parameters imposed (a grouped launch runs ONE instantiation: members
// adopt the parameters of the member with the largest grid when they can)
template <typename T>
static int cv2_plan_forced(const msmc_conv_desc* d, Cv2Plan* pl, int nt, int ckm, int sb) {
    constexpr int XS = Elt<T>::CK + Elt<T>::VEC;
    pl->applies = 0;
    if ((long)d->Hin * d->Win * d->Cin >= (1L << 31) || (long)d->Hout * d->Wout >= (1L << 31)) return 0;
    if (ckm == 2 && (d->Cin < 2 * Elt<T>::CK || (d->Cin % Elt<T>::VEC) != 0)) return 0;
    size_t unused;
    int rc = cv_geometry(d, &pl->G, sizeof(T), XS, 32, &unused);
    if (rc) return rc;
    const long npix = (long)pl->G.IH * pl->G.IW;
    const size_t tables = (((size_t)(npix + 128 + 16) * sizeof(int)) + 15) & ~(size_t)15;
    const size_t xs = (size_t)Elt<T>::CK * ckm + Elt<T>::VEC;
    const size_t stage = ((size_t)npix + (size_t)d->ntaps * 32 * nt) * xs * sizeof(T);
    const size_t epi = (size_t)128 * (32 * nt + 4) * sizeof(float);
    const size_t lds = tables + (stage > epi ? stage : epi);
    if (lds > 160 * 1024) return 0;
    if (sb == 16) {
        const long chunk_vectors = (npix + (long)d->ntaps * 32 * nt) * (Elt<T>::CK * ckm / Elt<T>::VEC);
        if (chunk_vectors > 16L * 256 || (d->Cin % Elt<T>::VEC) != 0) return 0;
    }
    pl->applies = 1;
    pl->nt = nt;
    pl->ckm = ckm;
    pl->sb = sb;
    pl->lds = lds;
    pl->gx = (unsigned)(pl->G.tilesX * pl->G.tilesY * d->B);
    pl->gy = (unsigned)((d->Cout + 32 * nt - 1) / (32 * nt));
    return 0;
}

// launch one second-generation kernel instantiation: SINGLE (desc, geometry) or GROUP (CvGroupArgs)
#define CV2_INST(T_, NT_, CKM_, SB_, GROUP_, ...)                                                                  \
    do {                                                                                                           \
        if (GROUP_) {                                                                                              \
            rc = msmc_allow_lds((const void*)conv_gather2_group_kernel<T_, NT_, CKM_, SB_>, (int)lds);             \
            if (rc) return rc;                                                                                     \
            MSMC_LAUNCH((conv_gather2_group_kernel<T_, NT_, CKM_, SB_>), grid, dim3(256), lds,                     \
                        (msmc_stream_t)stream, *group);                                                            \
        } else {                                                                                                   \
            rc = msmc_allow_lds((const void*)conv_gather2_kernel<T_, NT_, CKM_, SB_>, (int)lds);                   \
            if (rc) return rc;                                                                                     \
            MSMC_LAUNCH((conv_gather2_kernel<T_, NT_, CKM_, SB_>), grid, dim3(256), lds, (msmc_stream_t)stream,    \
                        *d, *G);                                                                                   \
        }                                                                                                          \
    } while (0)
template <typename T>
static int cv2_dispatch(int nt, int ckm, int sb, dim3 grid, size_t lds, msmc_stream stream, const msmc_conv_desc* d,
                        const CvGeom* G, const CvGroupArgs* group) {
    int rc;
    const bool grp = group != nullptr;
#define CV2_SB(NT_, CKM_)                                                                                          \
    do {                                                                                                           \
        if (sb == 16) CV2_INST(T, NT_, CKM_, 16, grp);                                                             \
        else if (sb == 8) CV2_INST(T, NT_, CKM_, 8, grp);                                                          \
        else CV2_INST(T, NT_, CKM_, 4, grp);                                                                       \
    } while (0)
    if (nt == 2 && ckm == 2) CV2_SB(2, 2);
    else if (nt == 2) CV2_SB(2, 1);
    else if (ckm == 2) CV2_SB(1, 2);
    else CV2_SB(1, 1);
#undef CV2_SB
    msmc_conv_last = msmc_prof_name(msmc_kname2(grp ? "conv_gather2_group_kernel" : "conv_gather2_kernel", EltName<T>::v, nt, ckm, sb));
    return msmc_check_launch();
}

template <typename T>
static int cv_launch(const msmc_conv_desc* d, msmc_stream stream) {
    constexpr int XS = Elt<T>::CK + Elt<T>::VEC;
    Cv2Plan pl;
    int NT;
    int rc0 = cv2_plan<T>(d, &pl, &NT);
    if (rc0) return rc0;
    if (pl.applies)
        return cv2_dispatch<T>(pl.nt, pl.ckm, pl.sb, dim3(pl.gx, pl.gy), pl.lds, stream, d, &pl.G, nullptr);
    // the pipelined kernel pays a slot-table prologue: worth it from ~4 channel chunks on (measured per layer)
    const bool deep = d->Cin >= 4 * Elt<T>::CK || msmc_conv_pipeline_enabled >= 2;
    if ((d->Cin % Elt<T>::VEC) == 0 && msmc_conv_pipeline_enabled && deep) {
        // widest M tile that still leaves >= ~2 workgroups per CU
        const long points = (long)d->B * d->QH * d->QW;
        const long ntile = (d->Cout + 32 * NT - 1) / (32 * NT);
        int MT = 4;
        if (msmc_conv_pipeline_enabled == 2 || msmc_conv_pipeline_enabled == 4) MT = msmc_conv_pipeline_enabled;   // tests
        else while (MT > 1 && (points / (CV_BM * MT)) * ntile < 2 * MSMC_NUM_CU) MT >>= 1;
        bool done = false;
        int rc = 0;
        for (; MT >= 1 && !done; MT >>= 1) {
            if (NT == 2) {
                if (MT == 4) rc = cv_try_pipe<T, 2, 4>(d, stream, &done);
                else if (MT == 2) rc = cv_try_pipe<T, 2, 2>(d, stream, &done);
                else rc = cv_try_pipe<T, 2, 1>(d, stream, &done);
            } else {
                if (MT == 4) rc = cv_try_pipe<T, 1, 4>(d, stream, &done);
                else if (MT == 2) rc = cv_try_pipe<T, 1, 2>(d, stream, &done);
                else rc = cv_try_pipe<T, 1, 1>(d, stream, &done);
            }
            if (rc) return rc;
        }
        if (done) return 0;
    }
    CvGeom G;
    size_t lds;
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
        msmc_conv_last = msmc_prof_name(msmc_kname("conv_gather_kernel", EltName<T>::v, 2, -1));
    } else {
        rc = msmc_allow_lds((const void*)conv_gather_kernel<T, 1>, (int)lds);
        if (rc) return rc;
        MSMC_LAUNCH((conv_gather_kernel<T, 1>), grid, dim3(256), lds, (msmc_stream_t)stream, *d, G);
        msmc_conv_last = msmc_prof_name(msmc_kname("conv_gather_kernel", EltName<T>::v, 1, -1));
    }
    return msmc_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Direct (matrix-core-free) kernels for the layers an MFMA tile cannot fill: the first discriminator layers
// (2 -> 4 .. 8 -> 16 channels), the score layers (C -> 1) and their data gradients (1 -> C).  They are a few MB of
// memory traffic each; a 128 x 32 MFMA tile spends its time staging 64-byte channel chunks that are 87-97 % padding.
//   small: one work-item per lattice point, all (<= 16) output channels in registers, weights as floats in LDS
//   dot  : one wave per lattice point, lanes over 16-byte channel vectors, butterfly reduction      (Cout == 1)
//   outer: one work-item per (lattice point, 16-byte vector of output channels)                      (Cin == 1)
// Same descriptor semantics (lattice, taps, padding rule, input activation, epilogue) as conv_gather_kernel.
// ------------------------------------------------------------------------------------------------
struct DirPoint {
    int b, qy, qx;
    size_t out;                 // element offset of the output pixel's channel 0
};
MSMC_DEV DirPoint dir_point(const msmc_conv_desc& d, long p) {
    DirPoint r;
    const int per = d.QH * d.QW;
    r.b = (int)(p / per);
    const int rem = (int)(p - (long)r.b * per);
    r.qy = rem / d.QW;
    r.qx = rem - r.qy * d.QW;
    r.out = (((size_t)r.b * d.Hout + (d.oy0 + r.qy * d.osy)) * d.Wout + (d.ox0 + r.qx * d.osx)) * d.Cout;
    return r;
}
// element offset of input pixel (channel 0) for tap t of a point, or -1 when the tap reads zero padding
MSMC_DEV long dir_in(const msmc_conv_desc& d, const DirPoint& pt, int t) {
    int iy = pt.qy * d.isy + d.iy0 + d.tap_dy[t], ix = pt.qx * d.isx + d.ix0 + d.tap_dx[t];
    if (d.pad_mode == 1) {
        iy = reflect_index(iy, d.Hin);
        ix = reflect_index(ix, d.Win);
    } else if (iy < 0 || iy >= d.Hin || ix < 0 || ix >= d.Win) {
        return -1;
    }
    return (((long)pt.b * d.Hin + iy) * d.Win + ix) * d.Cin;
}
template <typename T>
MSMC_DEV float dir_epilogue(const msmc_conv_desc& d, float v, size_t o, int co) {
    if (d.bias) v = v + d.bias[co];
    if (d.mask_src) v = v * (Elt<T>::ld((const T*)d.mask_src + o) > 0.f ? 1.f : d.mask_slope);
    if (d.res) v = v + Elt<T>::ld((const T*)d.res + o);
    if (d.res2) v = Elt<T>::ld((const T*)d.res2 + o) + v;
    if (d.out_div != 1.f) v = v / d.out_div;
    if (d.out_slope != 1.f) v = v > 0.f ? v : v * d.out_slope;
    return v;
}
MSMC_DEV float dir_act(float f, float slope) { return (slope == 1.f || f > 0.f) ? f : f * slope; }

// A run of N consecutive output channels (N * sizeof(T) a multiple of 8 bytes, the run aligned to its size): the optional operands
// arrive as ONE vector load each and the result leaves as one vector store per 16 bytes -- written per element (dir_epilogue in a
// loop, Elt::st per channel) every work-item issued N two-byte loads per operand and N two-byte stores, each a memory request
// of its own (round 6: the thin first / last layers of the discriminators ran at 7-12 % of the HBM roofline).  Same arithmetic, in
// the same order, as dir_epilogue.
template <typename T, int N>
MSMC_DEV void dir_epilogue_run(const msmc_conv_desc& d, float (&v)[N], const size_t o, const int co0) {
    constexpr int BYTES = N * (int)sizeof(T);
    static_assert(BYTES % 8 == 0, "vector run");
    alignas(16) T mk[N], r1[N], r2[N], ov[N];
    auto ldrun = [&](const T* src, T (&dst)[N]) {
        if constexpr (BYTES % 16 == 0) {
#pragma unroll
            for (int q = 0; q < BYTES / 16; ++q) ((u32x4*)dst)[q] = ((const u32x4*)(src + o))[q];
        } else {
            *(u32x2*)dst = *(const u32x2*)(src + o);
        }
    };
    if (d.bias) {
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = v[q] + d.bias[co0 + q];
    }
    if (d.mask_src) {
        ldrun((const T*)d.mask_src, mk);
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = v[q] * (Elt<T>::ld(&mk[q]) > 0.f ? 1.f : d.mask_slope);
    }
    if (d.res) {
        ldrun((const T*)d.res, r1);
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = v[q] + Elt<T>::ld(&r1[q]);
    }
    if (d.res2) {
        ldrun((const T*)d.res2, r2);
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = Elt<T>::ld(&r2[q]) + v[q];
    }
    if (d.out_div != 1.f) {
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = v[q] / d.out_div;
    }
    if (d.out_slope != 1.f) {
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * d.out_slope;
    }
#pragma unroll
    for (int q = 0; q < N; ++q) Elt<T>::st(&ov[q], v[q]);
    T* out = (T*)d.out + o;
    if constexpr (BYTES % 16 == 0) {
#pragma unroll
        for (int q = 0; q < BYTES / 16; ++q) ((u32x4*)out)[q] = ((const u32x4*)ov)[q];
    } else {
        *(u32x2*)out = *(const u32x2*)ov;
    }
}

template <typename T, int CI, int CO>
MSMC_DEV void dir_small_body(const msmc_conv_desc& d, const long npoints, const int block, const int nblocks) {
    MSMC_DYN_LDS(smem);
    float* wl = (float*)smem;                       // [ntaps][CI][CO], zero beyond the real channels
    for (int e = threadIdx.x; e < d.ntaps * CI * CO; e += 256) {
        const int co = e % CO, ci = (e / CO) % CI, t = e / (CO * CI);
        float v = 0.f;
        if (co < d.Cout && ci < d.Cin) v = Elt<T>::ld((const T*)d.w + ((size_t)d.tap_w[t] * d.Cout + co) * d.Cin + ci);
        wl[e] = v;
    }
    __syncthreads();
    const T* x = (const T*)d.x;
    T* out = (T*)d.out;
    for (long p = (long)block * 256 + threadIdx.x; p < npoints; p += (long)nblocks * 256) {
        const DirPoint pt = dir_point(d, p);
        float acc[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = 0.f;
        // (the input vectors of three taps are requested together -- taps outside the image re-read the point's first pixel and are
        //  skipped in the sum -- instead of one memory round trip per tap)
        for (int t0 = 0; t0 < d.ntaps; t0 += 3) {
            alignas(16) T xv[3][CI];
            bool on[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const long off_ = t0 + u < d.ntaps ? dir_in(d, pt, t0 + u) : -1;
                on[u] = off_ >= 0;
                const long off = on[u] ? off_ : 0;
                if (CI * sizeof(T) == 16 && d.Cin == CI) {
                    *(u32x4*)xv[u] = *(const u32x4*)(x + off);
                } else if (CI * sizeof(T) == 8 && d.Cin == CI) {
                    *(u32x2*)xv[u] = *(const u32x2*)(x + off);
                } else if (CI * sizeof(T) == 4 && d.Cin == CI) {
                    *(unsigned int*)xv[u] = *(const unsigned int*)(x + off);
                } else {
#pragma unroll
                    for (int ci = 0; ci < CI; ++ci) xv[u][ci] = ci < d.Cin ? x[off + ci] : (T)0;
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (!on[u]) continue;
                const float* wt = wl + (t0 + u) * CI * CO;
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
                    const float xf = dir_act(Elt<T>::ld(&xv[u][ci]), d.in_slope);
#pragma unroll
                    for (int co = 0; co < CO; ++co) acc[co] = fmaf(wt[ci * CO + co], xf, acc[co]);
                }
            }
        }
        if constexpr ((CO * sizeof(T)) % 8 == 0) {
            if (d.Cout == CO) {                     // (the whole channel run of the point: vector operands, vector store)
                dir_epilogue_run<T, CO>(d, acc, pt.out, 0);
                continue;
            }
        }
#pragma unroll
        for (int co = 0; co < CO; ++co)
            if (co < d.Cout) Elt<T>::st(out + pt.out + co, dir_epilogue<T>(d, acc[co], pt.out + co, co));
    }
}

struct DirGroupArgs {
    int n;
    int first[MSMC_GROUP_MAX + 1];
    long items[MSMC_GROUP_MAX];         // lattice points (small, dot) or point x vector items (outer) of member k
    msmc_conv_desc d[MSMC_GROUP_MAX];
};
template <typename T, int CI, int CO>
__global__ __launch_bounds__(256) void conv_direct_small_kernel(msmc_conv_desc d, long npoints) {
    dir_small_body<T, CI, CO>(d, npoints, blockIdx.x, gridDim.x);
}
template <typename T, int CI, int CO>
__global__ __launch_bounds__(256) void conv_direct_small_group_kernel(DirGroupArgs a) {
    const int k = cv_group_member(a.first, a.n);
    dir_small_body<T, CI, CO>(a.d[k], a.items[k], blockIdx.x - a.first[k], a.first[k + 1] - a.first[k]);
}

template <typename T>
MSMC_DEV void dir_dot_body(const msmc_conv_desc& d, const long npoints, const int block, const int nblocks) {
    MSMC_DYN_LDS(smem);
    constexpr int VEC = Elt<T>::VEC;
    T* wl = (T*)smem;                               // [ntaps][Cin]
    const int nvec = d.Cin / VEC;
    for (int e = threadIdx.x; e < d.ntaps * nvec; e += 256) {
        const int t = e / nvec, v = e - t * nvec;
        *(u32x4*)(wl + (size_t)t * d.Cin + v * VEC) = *(const u32x4*)((const T*)d.w + (size_t)d.tap_w[t] * d.Cin + v * VEC);
    }
    __syncthreads();
    const T* x = (const T*)d.x;
    T* out = (T*)d.out;
    const int lane = threadIdx.x & 63;
    const long wave = (long)block * 4 + (threadIdx.x >> 6), nwaves = (long)nblocks * 4;
    for (long p = wave; p < npoints; p += nwaves) {
        const DirPoint pt = dir_point(d, p);
        float acc = 0.f;
        for (int t = 0; t < d.ntaps; ++t) {
            const long off = dir_in(d, pt, t);
            if (off < 0) continue;
            for (int v = lane; v < nvec; v += 64) {
                alignas(16) T xv[VEC], wv[VEC];
                *(u32x4*)xv = *(const u32x4*)(x + off + v * VEC);
                *(u32x4*)wv = *(const u32x4*)(wl + (size_t)t * d.Cin + v * VEC);
#pragma unroll
                for (int q = 0; q < VEC; ++q)
                    acc = fmaf(Elt<T>::ld(&wv[q]), dir_act(Elt<T>::ld(&xv[q]), d.in_slope), acc);
            }
        }
        for (int m = 1; m < 64; m <<= 1) acc = acc + wave_xor(acc, m);
        if (lane == 0) Elt<T>::st(out + pt.out, dir_epilogue<T>(d, acc, pt.out, 0));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void conv_direct_dot_kernel(msmc_conv_desc d, long npoints) {
    dir_dot_body<T>(d, npoints, blockIdx.x, gridDim.x);
}
template <typename T>
__global__ __launch_bounds__(256) void conv_direct_dot_group_kernel(DirGroupArgs a) {
    const int k = cv_group_member(a.first, a.n);
    dir_dot_body<T>(a.d[k], a.items[k], blockIdx.x - a.first[k], a.first[k + 1] - a.first[k]);
}

template <typename T>
MSMC_DEV void dir_outer_body(const msmc_conv_desc& d, const long nitems, const int block, const int nblocks) {
    MSMC_DYN_LDS(smem);
    constexpr int VEC = Elt<T>::VEC;
    T* wl = (T*)smem;                               // [ntaps][Cout]   (Cin == 1)
    const int nvec = d.Cout / VEC;
    for (int e = threadIdx.x; e < d.ntaps * nvec; e += 256) {
        const int t = e / nvec, v = e - t * nvec;
        *(u32x4*)(wl + (size_t)t * d.Cout + v * VEC) = *(const u32x4*)((const T*)d.w + (size_t)d.tap_w[t] * d.Cout + v * VEC);
    }
    __syncthreads();
    const T* x = (const T*)d.x;
    T* out = (T*)d.out;
    for (long it = (long)block * 256 + threadIdx.x; it < nitems; it += (long)nblocks * 256) {
        const long p = it / nvec;
        const int v = (int)(it - p * nvec);
        const DirPoint pt = dir_point(d, p);
        float acc[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
        // the input values of four taps at a time: requested together (taps outside the image re-read element 0 and are skipped
        // in the sum, as before) -- one memory round trip per four taps instead of one per tap
        for (int t0 = 0; t0 < d.ntaps; t0 += 4) {
            float xf[4];
            bool on[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long off = t0 + u < d.ntaps ? dir_in(d, pt, t0 + u) : -1;
                on[u] = off >= 0;
                xf[u] = Elt<T>::ld(x + (on[u] ? off : 0));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!on[u]) continue;
                const float xa = dir_act(xf[u], d.in_slope);
                alignas(16) T wv[VEC];
                *(u32x4*)wv = *(const u32x4*)(wl + (size_t)(t0 + u) * d.Cout + v * VEC);
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] = fmaf(Elt<T>::ld(&wv[q]), xa, acc[q]);
            }
        }
        dir_epilogue_run<T, VEC>(d, acc, pt.out + (size_t)v * VEC, v * VEC);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void conv_direct_outer_kernel(msmc_conv_desc d, long nitems) {
    dir_outer_body<T>(d, nitems, blockIdx.x, gridDim.x);
}
template <typename T>
__global__ __launch_bounds__(256) void conv_direct_outer_group_kernel(DirGroupArgs a) {
    const int k = cv_group_member(a.first, a.n);
    dir_outer_body<T>(a.d[k], a.items[k], blockIdx.x - a.first[k], a.first[k + 1] - a.first[k]);
}

// returns 1 when a direct kernel was launched, 0 when none applies, < 0 on error
template <typename T>
static int cv_direct_launch(const msmc_conv_desc* d, msmc_stream stream) {
    constexpr int VEC = Elt<T>::VEC;
    const long npoints = (long)d->B * d->QH * d->QW;
    auto blocks = [](long items) {
        long b = (items + 255) / 256;
        const long cap = 16L * MSMC_NUM_CU;
        return (unsigned)(b < 1 ? 1 : b > cap ? cap : b);
    };
    int rc;
    if (d->Cin <= 8 && d->Cout <= 16 && !(d->Cin == 1 && d->Cout % VEC == 0 && d->Cout >= 4 * VEC)) {
        const int CI = d->Cin <= 1 ? 1 : d->Cin <= 2 ? 2 : d->Cin <= 4 ? 4 : 8;
        const int CO = d->Cout <= 1 ? 1 : d->Cout <= 4 ? 4 : d->Cout <= 8 ? 8 : 16;
        const size_t lds = (size_t)d->ntaps * CI * CO * sizeof(float);
#define DIR_GO(CI_, CO_)                                                                                            \
    do {                                                                                                            \
        rc = msmc_allow_lds((const void*)conv_direct_small_kernel<T, CI_, CO_>, (int)lds);                          \
        if (rc) return rc;                                                                                          \
        MSMC_LAUNCH((conv_direct_small_kernel<T, CI_, CO_>), dim3(blocks(npoints)), dim3(256), lds,                 \
                    (msmc_stream_t)stream, *d, npoints);                                                            \
    } while (0)
#define DIR_CO(CI_)                                                                                                 \
    do {                                                                                                            \
        if (CO == 1) DIR_GO(CI_, 1);                                                                                \
        else if (CO == 4) DIR_GO(CI_, 4);                                                                           \
        else if (CO == 8) DIR_GO(CI_, 8);                                                                           \
        else DIR_GO(CI_, 16);                                                                                       \
    } while (0)
        if (CI == 1) DIR_CO(1);
        else if (CI == 2) DIR_CO(2);
        else if (CI == 4) DIR_CO(4);
        else DIR_CO(8);
#undef DIR_CO
#undef DIR_GO
        msmc_conv_last = msmc_prof_name(msmc_kname2("conv_direct_small_kernel", EltName<T>::v, CI, CO, 0));
        rc = msmc_check_launch();
        return rc ? rc : 1;
    }
    if (d->Cout == 1 && d->Cin % VEC == 0 && d->Cin >= 8 * VEC) {
        const size_t lds = (size_t)d->ntaps * d->Cin * sizeof(T);
        if (lds > 160 * 1024) return 0;
        rc = msmc_allow_lds((const void*)conv_direct_dot_kernel<T>, (int)lds);
        if (rc) return rc;
        long b = (npoints + 3) / 4;
        if (b > 8L * MSMC_NUM_CU) b = 8L * MSMC_NUM_CU;
        MSMC_LAUNCH((conv_direct_dot_kernel<T>), dim3((unsigned)(b < 1 ? 1 : b)), dim3(256), lds, (msmc_stream_t)stream, *d,
                    npoints);
        msmc_conv_last = msmc_prof_name(msmc_kname("conv_direct_dot_kernel", EltName<T>::v, 0, -1));
        rc = msmc_check_launch();
        return rc ? rc : 1;
    }
    if (d->Cin == 1 && d->Cout % VEC == 0) {
        const size_t lds = (size_t)d->ntaps * d->Cout * sizeof(T);
        if (lds > 160 * 1024) return 0;
        const long nitems = npoints * (d->Cout / VEC);
        rc = msmc_allow_lds((const void*)conv_direct_outer_kernel<T>, (int)lds);
        if (rc) return rc;
        MSMC_LAUNCH((conv_direct_outer_kernel<T>), dim3(blocks(nitems)), dim3(256), lds, (msmc_stream_t)stream, *d, nitems);
        msmc_conv_last = msmc_prof_name(msmc_kname("conv_direct_outer_kernel", EltName<T>::v, 0, -1));
        rc = msmc_check_launch();
        return rc ? rc : 1;
    }
    return 0;
}

template <typename T>
static int cv_ks_launch(const msmc_conv_desc* d, msmc_stream stream);

extern "C" int msmc_conv_gather(const msmc_conv_desc* d, msmc_stream stream) {
    if (!d || d->B <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->QH <= 0 || d->QW <= 0) return MSMC_E_SHAPE;
    ++msmc_conv_launches;
    // variant 8 = direct kernels (E_SHAPE when none applies); without a tuned variant they are the default for the
    // layers they cover (generation 1 keeps the MFMA kernels everywhere: A/B tests)
    if (d->variant == 8 || (d->variant == 0 && msmc_gather_generation >= 2)) {
        int rc = d->dtype == 0 ? cv_direct_launch<float>(d, stream)
                 : d->dtype == 1 ? cv_direct_launch<unsigned short>(d, stream) : MSMC_E_SHAPE;
        if (rc != 0) return rc < 0 ? rc : 0;
        if (d->variant == 8) return MSMC_E_SHAPE;
    }
    if (cv3_is_variant(d->variant)) {           // third generation (bf16): E_SHAPE when the configuration does not apply
        const int rc = cv3_launch(d, stream);
        return rc < 0 ? rc : rc == 1 ? 0 : MSMC_E_SHAPE;
    }
    if (cv4_is_variant(d->variant)) {           // persistent thin-layer kernel (bf16): E_SHAPE outside its scope
        const int rc = cv4_launch(d, stream);
        return rc < 0 ? rc : rc == 1 ? 0 : MSMC_E_SHAPE;
    }
    if (g1_is_variant(d->variant)) {            // 1-tap layers as a plain channel GEMM (bf16): E_SHAPE outside its scope
        const int rc = g1_launch(d, stream);
        return rc < 0 ? rc : rc == 1 ? 0 : MSMC_E_SHAPE;
    }
    if (cv5_is_variant(d->variant)) {           // fifth generation (bf16, sixteen waves, staged taps): E_SHAPE outside its scope
        const int rc = cv5_launch(d, stream);
        return rc < 0 ? rc : rc == 1 ? 0 : MSMC_E_SHAPE;
    }
    if (cv7_is_variant(d->variant)) {           // seventh generation (bf16, eight waves of 64 x 64, two workgroups per CU): E_SHAPE outside its scope
        const int rc = cv7_launch(d, stream);
        return rc < 0 ? rc : rc == 1 ? 0 : MSMC_E_SHAPE;
    }
    if (cv6_is_variant(d->variant)) {           // thin-channel kernel (bf16, Cin 8 / 16 / 32, fragments straight from global memory)
        const int rc = cv6_launch(d, stream);
        return rc < 0 ? rc : rc == 1 ? 0 : MSMC_E_SHAPE;
    }
    if (d->variant == 9) {                      // wave-split deep reduction (E_SHAPE when it does not apply)
        int rc = d->dtype == 0 ? cv_ks_launch<float>(d, stream)
                 : d->dtype == 1 ? cv_ks_launch<unsigned short>(d, stream) : MSMC_E_SHAPE;
        return rc < 0 ? rc : rc == 1 ? 0 : MSMC_E_SHAPE;
    }
    if (d->dtype == 0) return cv_launch<float>(d, stream);
    if (d->dtype == 1) return cv_launch<unsigned short>(d, stream);
    return MSMC_E_SHAPE;
}

// ------------------------------------------------------------------------------------------------
// Deep reductions on small grids (FFT-block convolutions at T/4, conv_pre, the deep MPD / MRD layers): the 128-point
// kernels leave most CUs idle and walk 16-38 dependent channel-chunk round trips.  Here a workgroup owns 32 lattice
// points x BN channels and its four waves split the channel chunks (wave w takes chunks w, w+4, ...), each staging
// into its own LDS region with wave-level hand-offs only; the four partial tiles meet in LDS before the epilogue.
// Four times the workgroups, a quarter of the chain.
// ------------------------------------------------------------------------------------------------
template <typename T, int NT, int CKM>
MSMC_DEV void cvks_body(const msmc_conv_desc& d, const CvGeom& G, const int region_bytes, const int block_x, const int block_y) {
    MSMC_DYN_LDS(smem);
    constexpr int VEC = Elt<T>::VEC, CK = Elt<T>::CK * CKM, CKV = CK / VEC, XS = CK + VEC, BN = 32 * NT, OS = BN + 4;
    constexpr int BNV = BN / VEC, SB = 8;
    const int npix = G.IH * G.IW, IW = G.IW;
    int* in_off = (int*)smem;                                   // [npix]
    int* out_off = in_off + npix;                               // [32]
    int* tapw = out_off + 32;                                   // [16]
    char* regions = smem + (((size_t)(npix + 32 + 16) * sizeof(int) + 15) & ~(size_t)15);
    const int tid = threadIdx.x, w = wave_uniform(tid >> 6), lane = tid & 63, i = lane & 31, g = lane >> 5;
    T* xt = (T*)(regions + (size_t)w * region_bytes);           // this wave's [npix][XS]
    T* wt = xt + (size_t)npix * XS;                             //             [ntaps][BN][XS]
    int bt = block_x;
    const int tx_ = bt % G.tilesX;
    bt /= G.tilesX;
    const int ty_ = bt % G.tilesY;
    const int b = bt / G.tilesY;
    const int co0 = block_y * BN;
    const int qy0 = ty_ * G.TH, qx0 = tx_ * G.TW;
    const int iyBase = qy0 * d.isy + d.iy0 + G.dyMin, ixBase = qx0 * d.isx + d.ix0 + G.dxMin;
    for (int pi = tid; pi < npix; pi += 256) {
        const int ry = pi / IW, rx = pi - ry * IW;
        int iy = iyBase + ry, ix = ixBase + rx;
        bool inside = true;
        if (d.pad_mode == 1) {
            iy = reflect_index(iy, d.Hin);
            ix = reflect_index(ix, d.Win);
        } else {
            inside = (iy >= 0) && (iy < d.Hin) && (ix >= 0) && (ix < d.Win);
        }
        in_off[pi] = inside ? (iy * d.Win + ix) * d.Cin : -1;
    }
    if (tid < 32) {
        const int mty = tid / G.TW, mtx = tid - mty * G.TW;
        const int qy = qy0 + mty, qx = qx0 + mtx;
        const bool valid = mty < G.TH && qy < d.QH && qx < d.QW;
        out_off[tid] = valid ? (d.oy0 + qy * d.osy) * d.Wout + (d.ox0 + qx * d.osx) : -1;
    }
#pragma unroll
    for (int t = 0; t < MSMC_CONV_MAX_TAPS; ++t)
        if (tid == 64 + t) tapw[t] = t < d.ntaps ? d.tap_w[t] : 0;
    int arow;
    {
        const int mty = i / G.TW, mtx = i - mty * G.TW;
        arow = (mty < G.TH) ? (mty * d.isy) * IW + mtx * d.isx : 0;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const T* xb = (const T*)d.x + (size_t)b * d.Hin * d.Win * d.Cin;
    const T* wg = (const T*)d.w;
    const float slope = d.in_slope;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int nxv = npix * CKV, nwv = d.ntaps * BN * CKV;
    __syncthreads();                                            // offset tables ready

    for (int c0 = w * CK; c0 < d.Cin; c0 += 4 * CK) {
        for (int e0 = lane; e0 < nxv; e0 += 64 * SB) {
            u32x4 vals[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int e = e0 + 64 * u;
                vals[u] = zero4;
                if (e < nxv) {
                    const int pi = e / CKV;
                    const int off = in_off[pi], c = c0 + (e - pi * CKV) * VEC;
                    if (off >= 0 && c < d.Cin) vals[u] = *(const u32x4*)(xb + off + c);
                }
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int e = e0 + 64 * u;
                if (e >= nxv) continue;
                const int pi = e / CKV;
                if (slope != 1.f) {
                    alignas(16) T tmp[VEC];
                    *(u32x4*)tmp = vals[u];
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        float f = Elt<T>::ld(&tmp[q]);
                        f = f > 0.f ? f : f * slope;
                        Elt<T>::st(&tmp[q], f);
                    }
                    vals[u] = *(const u32x4*)tmp;
                }
                *(u32x4*)(xt + (size_t)pi * XS + (e - pi * CKV) * VEC) = vals[u];
            }
        }
        for (int e0 = lane; e0 < nwv; e0 += 64 * SB) {
            u32x4 vals[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int e = e0 + 64 * u;
                vals[u] = zero4;
                if (e < nwv) {
                    const int row = e / CKV;
                    const int t = row / BN, co = co0 + (row - t * BN), c = c0 + (e - row * CKV) * VEC;
                    if (co < d.Cout && c < d.Cin) vals[u] = *(const u32x4*)(wg + ((size_t)tapw[t] * d.Cout + co) * d.Cin + c);
                }
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int e = e0 + 64 * u;
                if (e >= nwv) continue;
                const int row = e / CKV;
                *(u32x4*)(wt + (size_t)row * XS + (e - row * CKV) * VEC) = vals[u];
            }
        }
        wave_sync();                                            // this wave's tiles are complete
        for (int t = 0; t < d.ntaps; ++t) {
            const T* ap = xt + (size_t)(arow + (d.tap_dy[t] - G.dyMin) * IW + (d.tap_dx[t] - G.dxMin)) * XS;
            const T* bp = wt + (size_t)(t * BN + i) * XS;
#pragma unroll
            for (int ks = 0; ks < CK / 16; ++ks) mma_chunk16<NT>(ap + ks * 16, bp + ks * 16, 32 * XS, g, acc);
        }
        wave_sync();                                            // fragments consumed before the next chunk lands
    }
    // ---- the four partial tiles meet in LDS (each wave parks its own in its own region), then the usual epilogue
    float* pt = (float*)(regions + (size_t)w * region_bytes);   // [32][OS]
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) pt[((r & 3) + 8 * (r >> 2) + 4 * g) * OS + n * 32 + i] = acc[n][r];
    __syncthreads();
    const T* mask = (const T*)d.mask_src;
    const T* res = (const T*)d.res;
    const T* res2 = (const T*)d.res2;
    T* out = (T*)d.out;
    const size_t img = (size_t)b * d.Hout * d.Wout;
    const int fstride = region_bytes / (int)sizeof(float);
    const float* p0 = (const float*)regions;
    for (int e = tid; e < 32 * BNV; e += 256) {
        const int m = e / BNV, vcol = (e - m * BNV) * VEC;
        const int po = out_off[m], co = co0 + vcol;
        if (po < 0 || co >= d.Cout) continue;
        const size_t o = (img + po) * d.Cout + co;
        alignas(16) T mk[VEC], r1[VEC], r2[VEC], ov[VEC];
        if (mask) *(u32x4*)mk = *(const u32x4*)(mask + o);
        if (res) *(u32x4*)r1 = *(const u32x4*)(res + o);
        if (res2) *(u32x4*)r2 = *(const u32x4*)(res2 + o);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int idx = m * OS + vcol + q;
            float x = ((p0[idx] + p0[fstride + idx]) + p0[2 * fstride + idx]) + p0[3 * fstride + idx];
            if (d.bias) x = x + d.bias[co + q];
            if (mask) x = x * (Elt<T>::ld(&mk[q]) > 0.f ? 1.f : d.mask_slope);
            if (res) x = x + Elt<T>::ld(&r1[q]);
            if (res2) x = Elt<T>::ld(&r2[q]) + x;
            if (d.out_div != 1.f) x = x / d.out_div;
            if (d.out_slope != 1.f) x = x > 0.f ? x : x * d.out_slope;
            Elt<T>::st(&ov[q], x);
        }
        *(u32x4*)(out + o) = *(const u32x4*)ov;
    }
}

// returns 1 when launched, 0 when the kernel does not apply
template <typename T, int NT, int CKM>
__global__ __launch_bounds__(256) void conv_gather_ks_kernel(msmc_conv_desc d, CvGeom G, int region_bytes) {
    cvks_body<T, NT, CKM>(d, G, region_bytes, blockIdx.x, blockIdx.y);
}
// the wave-split members of a grouped call on one grid (the phases of a transposed convolution, of a strided layer's data
// gradient: five to thirteen launches of tens of workgroups each otherwise)
struct CvKsGroupArgs {
    CvGroupArgs a;
    int reg[MSMC_GROUP_MAX];
};
template <typename T, int NT, int CKM>
__global__ __launch_bounds__(256) void conv_gather_ks_group_kernel(CvKsGroupArgs ga) {
    const int k = cv_group_member(ga.a.first, ga.a.n);
    const int id = blockIdx.x - ga.a.first[k];
    cvks_body<T, NT, CKM>(ga.a.d[k], ga.a.G[k], ga.reg[k], id % ga.a.nx[k], id / ga.a.nx[k]);
}

struct CvKsPlan {
    int applies, nt, ckm;
    CvGeom G;
    size_t reg, lds;
    unsigned gx, gy;
};
template <typename T>
static int cv_ks_plan(const msmc_conv_desc* d, CvKsPlan* pl) {
    constexpr int VEC = Elt<T>::VEC;
    pl->applies = 0;
    if ((d->Cin % VEC) != 0 || (d->Cout % VEC) != 0 || d->Cin < 8 * Elt<T>::CK) return 0;
    if ((long)d->Hin * d->Win * d->Cin >= (1L << 31) || (long)d->Hout * d->Wout >= (1L << 31)) return 0;
    size_t unused;
    int rc = cv_geometry(d, &pl->G, sizeof(T), 0, 0, &unused, 32);
    if (rc) return rc;
    const long npix = (long)pl->G.IH * pl->G.IW;
    const int nt = d->Cout > 32 ? 2 : 1;
    const size_t tables = (((size_t)(npix + 32 + 16) * sizeof(int)) + 15) & ~(size_t)15;
    auto region = [&](int ckm) {
        const size_t xs = (size_t)Elt<T>::CK * ckm + VEC;
        size_t r = ((size_t)npix + (size_t)d->ntaps * 32 * nt) * xs * sizeof(T);
        const size_t part = (size_t)32 * (32 * nt + 4) * sizeof(float);
        if (r < part) r = part;
        return (r + 15) & ~(size_t)15;
    };
    const int ckm = (tables + 4 * region(2) <= 150 * 1024) ? 2 : 1;
    pl->reg = region(ckm);
    pl->lds = tables + 4 * pl->reg;
    if (pl->lds > 160 * 1024) return 0;
    pl->nt = nt;
    pl->ckm = ckm;
    pl->gx = (unsigned)(pl->G.tilesX * pl->G.tilesY * d->B);
    pl->gy = (unsigned)((d->Cout + 32 * nt - 1) / (32 * nt));
    pl->applies = 1;
    return 0;
}
// SINGLE (d, G) or GROUP launch of one wave-split configuration
template <typename T>
static int cv_ks_dispatch(const CvKsPlan& pl, dim3 grid, size_t lds, msmc_stream stream, const msmc_conv_desc* d,
                          const CvKsGroupArgs* group) {
    int rc;
#define KS_GO(NT_, CKM_)                                                                                            \
    do {                                                                                                            \
        if (group) {                                                                                                \
            rc = msmc_allow_lds((const void*)conv_gather_ks_group_kernel<T, NT_, CKM_>, (int)lds);                  \
            if (rc) return rc;                                                                                      \
            MSMC_LAUNCH((conv_gather_ks_group_kernel<T, NT_, CKM_>), grid, dim3(256), lds, (msmc_stream_t)stream,   \
                        *group);                                                                                    \
        } else {                                                                                                    \
            rc = msmc_allow_lds((const void*)conv_gather_ks_kernel<T, NT_, CKM_>, (int)lds);                        \
            if (rc) return rc;                                                                                      \
            MSMC_LAUNCH((conv_gather_ks_kernel<T, NT_, CKM_>), grid, dim3(256), lds, (msmc_stream_t)stream, *d,     \
                        pl.G, (int)pl.reg);                                                                         \
        }                                                                                                           \
    } while (0)
    if (pl.nt == 2 && pl.ckm == 2) KS_GO(2, 2);
    else if (pl.nt == 2) KS_GO(2, 1);
    else if (pl.ckm == 2) KS_GO(1, 2);
    else KS_GO(1, 1);
#undef KS_GO
    msmc_conv_last = msmc_prof_name(msmc_kname(group ? "conv_gather_ks_group_kernel" : "conv_gather_ks_kernel", EltName<T>::v,
                                               pl.nt, pl.ckm));
    return msmc_check_launch();
}
template <typename T>
static int cv_ks_launch(const msmc_conv_desc* d, msmc_stream stream) {
    CvKsPlan pl;
    int rc = cv_ks_plan<T>(d, &pl);
    if (rc) return rc;
    if (!pl.applies) return 0;
    rc = cv_ks_dispatch<T>(pl, dim3(pl.gx, pl.gy), pl.lds, stream, d, nullptr);
    return rc ? rc : 1;
}
// variant-9 members of a grouped call: one grid per (column tiles, chunk width) configuration.  done[i] = launched here.
template <typename T>
static int cv_ks_group_launch(const msmc_conv_desc* descs, int n, msmc_stream stream, bool* done) {
    CvKsPlan pk[MSMC_GROUP_LIMIT];
    bool todo[MSMC_GROUP_LIMIT];
    int count = 0;
    for (int i = 0; i < n; ++i) {
        done[i] = todo[i] = false;
        if (descs[i].variant != 9) continue;
        int rc = cv_ks_plan<T>(&descs[i], &pk[i]);
        if (rc) return rc;
        if (!pk[i].applies) return MSMC_E_SHAPE;
        todo[i] = true;
        ++count;
    }
    if (count < 2) return 0;                                    // (a lone member: the single launch below)
    for (int i = 0; i < n; ++i) {
        if (!todo[i]) continue;
        CvKsGroupArgs ga;
        ga.a.n = 0;
        int blocks = 0;
        size_t lds = 0;
        for (int j = i; j < n && ga.a.n < MSMC_GROUP_MAX; ++j) {
            if (!todo[j] || pk[j].nt != pk[i].nt || pk[j].ckm != pk[i].ckm) continue;
            ga.a.first[ga.a.n] = blocks;
            ga.a.nx[ga.a.n] = (int)pk[j].gx;
            ga.a.d[ga.a.n] = descs[j];
            ga.a.G[ga.a.n] = pk[j].G;
            ga.reg[ga.a.n] = (int)pk[j].reg;
            blocks += (int)(pk[j].gx * pk[j].gy);
            if (pk[j].lds > lds) lds = pk[j].lds;
            todo[j] = false;
            done[j] = true;
            ++ga.a.n;
        }
        ga.a.first[ga.a.n] = blocks;
        ++msmc_conv_launches;
        int rc = ga.a.n == 1 ? cv_ks_dispatch<T>(pk[i], dim3(pk[i].gx, pk[i].gy), pk[i].lds, stream, &ga.a.d[0], nullptr)
                             : cv_ks_dispatch<T>(pk[i], dim3((unsigned)blocks), lds, stream, nullptr, &ga);
        if (rc) return rc;
    }
    return 0;
}

// which direct kernel would take this descriptor: 0 none, 1 small, 2 dot, 3 outer (mirrors cv_direct_launch)
static int cv_direct_kind(const msmc_conv_desc* d) {
    const int VEC = d->dtype == 0 ? 4 : 8;
    if (d->Cin <= 8 && d->Cout <= 16 && !(d->Cin == 1 && d->Cout % VEC == 0 && d->Cout >= 4 * VEC)) return 1;
    if (d->Cout == 1 && d->Cin % VEC == 0 && d->Cin >= 8 * VEC)
        return (size_t)d->ntaps * d->Cin * (d->dtype == 0 ? 4 : 2) <= 160 * 1024 ? 2 : 0;
    if (d->Cin == 1 && d->Cout % VEC == 0) return (size_t)d->ntaps * d->Cout * (d->dtype == 0 ? 4 : 2) <= 160 * 1024 ? 3 : 0;
    return 0;
}
static bool cv_takes_direct(const msmc_conv_desc* d) {
    return (d->variant == 8 || (d->variant == 0 && msmc_gather_generation >= 2)) && cv_direct_kind(d) != 0;
}

struct DirKey {
    int kind, ci, co;
};
static DirKey cv_direct_key(const msmc_conv_desc* d) {
    DirKey k = {cv_direct_kind(d), 0, 0};
    if (k.kind == 1) {
        k.ci = d->Cin <= 1 ? 1 : d->Cin <= 2 ? 2 : d->Cin <= 4 ? 4 : 8;
        k.co = d->Cout <= 1 ? 1 : d->Cout <= 4 ? 4 : d->Cout <= 8 ? 8 : 16;
    }
    return k;
}

// direct-kernel members with one key (same kernel instantiation) as one grid
template <typename T>
static int cv_direct_group_launch(const msmc_conv_desc* const* members, int m, DirKey key, msmc_stream stream) {
    constexpr int VEC = Elt<T>::VEC;
    DirGroupArgs a;
    a.n = m;
    int blocks = 0;
    size_t lds = 0;
    for (int k = 0; k < m; ++k) {
        const msmc_conv_desc* d = members[k];
        const long npoints = (long)d->B * d->QH * d->QW;
        long items = npoints, nb;
        size_t l;
        if (key.kind == 1) {
            nb = (npoints + 255) / 256;
            if (nb > 8L * MSMC_NUM_CU) nb = 8L * MSMC_NUM_CU;
            l = (size_t)d->ntaps * key.ci * key.co * sizeof(float);
        } else if (key.kind == 2) {
            nb = (npoints + 3) / 4;
            if (nb > 4L * MSMC_NUM_CU) nb = 4L * MSMC_NUM_CU;
            l = (size_t)d->ntaps * d->Cin * sizeof(T);
        } else {
            items = npoints * (d->Cout / VEC);
            nb = (items + 255) / 256;
            if (nb > 8L * MSMC_NUM_CU) nb = 8L * MSMC_NUM_CU;
            l = (size_t)d->ntaps * d->Cout * sizeof(T);
        }
        if (nb < 1) nb = 1;
        a.first[k] = blocks;
        a.items[k] = items;
        a.d[k] = *d;
        blocks += (int)nb;
        if (l > lds) lds = l;
    }
    a.first[m] = blocks;
    int rc;
    const dim3 grid((unsigned)blocks);
    if (key.kind == 1) {
#define DIRG_GO(CI_, CO_)                                                                                           \
    do {                                                                                                            \
        rc = msmc_allow_lds((const void*)conv_direct_small_group_kernel<T, CI_, CO_>, (int)lds);                    \
        if (rc) return rc;                                                                                          \
        MSMC_LAUNCH((conv_direct_small_group_kernel<T, CI_, CO_>), grid, dim3(256), lds, (msmc_stream_t)stream, a); \
    } while (0)
#define DIRG_CO(CI_)                                                                                                \
    do {                                                                                                            \
        if (key.co == 1) DIRG_GO(CI_, 1);                                                                           \
        else if (key.co == 4) DIRG_GO(CI_, 4);                                                                      \
        else if (key.co == 8) DIRG_GO(CI_, 8);                                                                      \
        else DIRG_GO(CI_, 16);                                                                                      \
    } while (0)
        if (key.ci == 1) DIRG_CO(1);
        else if (key.ci == 2) DIRG_CO(2);
        else if (key.ci == 4) DIRG_CO(4);
        else DIRG_CO(8);
#undef DIRG_CO
#undef DIRG_GO
        msmc_conv_last = msmc_prof_name(msmc_kname2("conv_direct_small_group_kernel", EltName<T>::v, key.ci, key.co, 0));
    } else if (key.kind == 2) {
        rc = msmc_allow_lds((const void*)conv_direct_dot_group_kernel<T>, (int)lds);
        if (rc) return rc;
        MSMC_LAUNCH((conv_direct_dot_group_kernel<T>), grid, dim3(256), lds, (msmc_stream_t)stream, a);
        msmc_conv_last = msmc_prof_name(msmc_kname("conv_direct_dot_group_kernel", EltName<T>::v, 0, -1));
    } else {
        rc = msmc_allow_lds((const void*)conv_direct_outer_group_kernel<T>, (int)lds);
        if (rc) return rc;
        MSMC_LAUNCH((conv_direct_outer_group_kernel<T>), grid, dim3(256), lds, (msmc_stream_t)stream, a);
        msmc_conv_last = msmc_prof_name(msmc_kname("conv_direct_outer_group_kernel", EltName<T>::v, 0, -1));
    }
    return msmc_check_launch();
}

static int msmc_conv_grouping = 1;              // 0: grouped entry points launch their members one by one (A/B)
extern "C" void msmc_conv_set_grouping(int on) { msmc_conv_grouping = on; }

template <typename T>
static int cv_group_launch(const msmc_conv_desc* descs, int n, msmc_stream stream) {
    Cv2Plan plans[MSMC_GROUP_LIMIT];
    bool pending[MSMC_GROUP_LIMIT], direct[MSMC_GROUP_LIMIT];
    // third-generation members: one grid per configuration (tile shape, chunk, halo registers)
    Cv3Plan p3[MSMC_GROUP_LIMIT];
    bool gen3[MSMC_GROUP_LIMIT];
    for (int i = 0; i < n; ++i) {
        gen3[i] = false;
        if (!cv3_is_variant(descs[i].variant)) continue;
        int rc = cv3_plan(&descs[i], descs[i].variant, &p3[i]);
        if (rc) return rc;
        if (!p3[i].applies) return MSMC_E_SHAPE;
        gen3[i] = true;
    }
    for (int i = 0; i < n; ++i) {
        if (!gen3[i]) continue;
        CvGroupArgs a;
        a.n = 0;
        int blocks = 0;
        size_t lds = 0;
        for (int j = i; j < n && a.n < MSMC_GROUP_MAX; ++j) {
            if (!gen3[j] || p3[j].wm != p3[i].wm || p3[j].ntw != p3[i].ntw || p3[j].ckm != p3[i].ckm || p3[j].xv != p3[i].xv ||
                p3[j].xdma != p3[i].xdma)
                continue;
            a.first[a.n] = blocks;
            a.nx[a.n] = (int)p3[j].gx;
            a.d[a.n] = descs[j];
            a.G[a.n] = p3[j].G;
            blocks += (int)(p3[j].gx * p3[j].gy);
            if (p3[j].lds > lds) lds = p3[j].lds;
            gen3[j] = false;
            ++a.n;
        }
        a.first[a.n] = blocks;
        ++msmc_conv_launches;
        int rc = a.n == 1 ? cv3_dispatch(p3[i], dim3(p3[i].gx, p3[i].gy), p3[i].lds, stream, &a.d[0], &a.G[0], nullptr)
                          : cv3_dispatch(p3[i], dim3((unsigned)blocks), lds, stream, nullptr, nullptr, &a);
        if (rc) return rc;
    }
    bool done4[MSMC_GROUP_LIMIT];                               // persistent thin-layer members on one grid (off by default)
    {
        const int rc = cv4_group_launch(descs, n, stream, done4);
        if (rc) return rc;
    }
    {
        bool done5[MSMC_GROUP_LIMIT];                           // fifth-generation members: one grid per configuration
        const int rc = cv5_group_launch(descs, n, stream, done5);
        if (rc) return rc;
        for (int i = 0; i < n; ++i) done4[i] = done4[i] || done5[i];
    }
    {
        bool done7[MSMC_GROUP_LIMIT];                           // seventh-generation members: one grid per configuration
        const int rc = cv7_group_launch(descs, n, stream, done7);
        if (rc) return rc;
        for (int i = 0; i < n; ++i) done4[i] = done4[i] || done7[i];
    }
    {
        bool done6[MSMC_GROUP_LIMIT];                           // thin-channel members (variant 50): one grid per channel count
        const int rc = cv6_group_launch(descs, n, stream, done6);
        if (rc) return rc;
        for (int i = 0; i < n; ++i) done4[i] = done4[i] || done6[i];
    }
    {
        bool dones[MSMC_GROUP_LIMIT];                           // split-bf16 constant-matrix GEMMs (variants 36 / 37): one grid per tile width
        const int rc = g1s_group_launch(descs, n, stream, dones);
        if (rc) return rc;
        for (int i = 0; i < n; ++i) done4[i] = done4[i] || dones[i];
    }
    {
        bool doneks[MSMC_GROUP_LIMIT];                          // wave-split members (variant 9): one grid per configuration
        const int rc = cv_ks_group_launch<T>(descs, n, stream, doneks);
        if (rc) return rc;
        for (int i = 0; i < n; ++i) done4[i] = done4[i] || doneks[i];
    }
    for (int i = 0; i < n; ++i) {
        pending[i] = direct[i] = false;
        const msmc_conv_desc* d = &descs[i];
        if (cv3_is_variant(d->variant) || done4[i]) continue;
        int nt_unused;
        const bool own_grid = d->variant == 9 || cv4_is_variant(d->variant) || g1_is_variant(d->variant);
        int rc = (cv_takes_direct(d) || own_grid) ? 0 : cv2_plan<T>(d, &plans[i], &nt_unused);
        if (rc) return rc;
        if (own_grid) {                                         // (one launch each: they fill the chip on their own)
            rc = msmc_conv_gather(d, stream);
            if (rc) return rc;
        } else if (cv_takes_direct(d)) {
            direct[i] = true;
        } else if (!plans[i].applies) {                         // first-generation kernels: one launch each
            rc = msmc_conv_gather(d, stream);
            if (rc) return rc;
        } else {
            pending[i] = true;
        }
    }
    for (int i = 0; i < n; ++i) {                               // direct kernels: one grid per kernel instantiation
        if (!direct[i]) continue;
        const DirKey key = cv_direct_key(&descs[i]);
        const msmc_conv_desc* members[MSMC_GROUP_MAX];
        int m = 0;
        for (int j = i; j < n && m < MSMC_GROUP_MAX; ++j) {
            if (!direct[j]) continue;
            const DirKey kj = cv_direct_key(&descs[j]);
            if (kj.kind != key.kind || kj.ci != key.ci || kj.co != key.co) continue;
            members[m++] = &descs[j];
            direct[j] = false;
        }
        int rc;
        if (m == 1) {
            rc = msmc_conv_gather(members[0], stream);
        } else {
            ++msmc_conv_launches;
            rc = cv_direct_group_launch<T>(members, m, key, stream);
        }
        if (rc) return rc;
    }
    for (;;) {
        // leader = pending member with the largest grid; the others adopt its kernel parameters when they can
        int i = -1;
        for (int j = 0; j < n; ++j)
            if (pending[j] && (i < 0 || plans[j].gx * plans[j].gy > plans[i].gx * plans[i].gy)) i = j;
        if (i < 0) break;
        CvGroupArgs a;
        a.n = 0;
        int blocks = 0;
        size_t lds = 0;
        for (int jj = 0; jj < n && a.n < MSMC_GROUP_MAX; ++jj) {
            const int j = jj == 0 ? i : (jj <= i ? jj - 1 : jj);          // leader first, then the rest in order
            if (!pending[j]) continue;
            if (plans[j].nt != plans[i].nt || plans[j].ckm != plans[i].ckm || plans[j].sb != plans[i].sb) {
                Cv2Plan alt;
                int rc = cv2_plan_forced<T>(&descs[j], &alt, plans[i].nt, plans[i].ckm, plans[i].sb);
                if (rc) return rc;
                if (!alt.applies) continue;
                plans[j] = alt;
            }
            a.first[a.n] = blocks;
            a.nx[a.n] = (int)plans[j].gx;
            a.d[a.n] = descs[j];
            a.G[a.n] = plans[j].G;
            blocks += (int)(plans[j].gx * plans[j].gy);
            if (plans[j].lds > lds) lds = plans[j].lds;
            pending[j] = false;
            ++a.n;
        }
        a.first[a.n] = blocks;
        ++msmc_conv_launches;
        int rc;
        if (a.n == 1)
            rc = cv2_dispatch<T>(plans[i].nt, plans[i].ckm, plans[i].sb, dim3(plans[i].gx, plans[i].gy), plans[i].lds, stream,
                                 &a.d[0], &a.G[0], nullptr);
        else
            rc = cv2_dispatch<T>(plans[i].nt, plans[i].ckm, plans[i].sb, dim3((unsigned)blocks), lds, stream, nullptr, nullptr,
                                 &a);
        if (rc) return rc;
    }
    return 0;
}

// n independent convolutions (msmc_conv_gather semantics each) issued as few launches as their kernel choices allow
extern "C" int msmc_conv_gather_group(const msmc_conv_desc* descs, int n, msmc_stream stream) {
    if (!descs || n <= 0 || n > MSMC_GROUP_LIMIT) return MSMC_E_SHAPE;
    bool same = true;
    for (int i = 0; i < n; ++i) {
        const msmc_conv_desc* d = &descs[i];
        if (d->B <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->QH <= 0 || d->QW <= 0) return MSMC_E_SHAPE;
        same = same && d->dtype == descs[0].dtype;
    }
    if (!msmc_conv_grouping || !same || n == 1) {
        for (int i = 0; i < n; ++i) {
            int rc = msmc_conv_gather(&descs[i], stream);
            if (rc) return rc;
        }
        return 0;
    }
    if (descs[0].dtype == 0) return cv_group_launch<float>(descs, n, stream);
    if (descs[0].dtype == 1) return cv_group_launch<unsigned short>(descs, n, stream);
    return MSMC_E_SHAPE;
}

// ================================================================================================
// weight gradient
// ================================================================================================
// dW[t][co][ci] += sum over lattice points p of g[p][co] * act(x[in(p,t)][ci]).  The reduction runs over
// pixels, so the MFMA K dimension is the pixel axis while LDS holds both operands in their natural
// channels-last layout ([pixel][channel], staged with the same halo-tile code as the forward kernel):
//   bf16: fragments come from ds_read_b64_tr_b16 (hardware transpose), every lane addressing the
//         pixel row it is responsible for -- a tap is again a pure row offset;
//   fp32: v_mfma_f32_32x32x2_f32 takes one element per lane, read straight from the tile.
// One workgroup owns a 64(co) x 64(ci) tile of every tap (4 waves x 32x32 fragments x TAPS accumulators)
// and walks a range of 128-point lattice tiles; partial sums meet in fp32 atomics.
#define WG_TM 128      // lattice points per tile (upper bound; smaller tiles when the halo would not fit LDS)

template <typename T>
MSMC_DEV void wg_stage_x(T* xt, int XS, const msmc_conv_desc& d, const CvGeom& G, const T* xb, int c0, int iyBase,
                         int ixBase, int tid) {
    constexpr int VEC = Elt<T>::VEC, CKV = 64 / VEC;
    const int npix = G.IH * G.IW;
    const bool vec_ok = (d.Cin % VEC) == 0;
    for (int e = tid; e < npix * CKV; e += 256) {
        const int pi = e / CKV, v = e - pi * CKV;
        const int ry = pi / G.IW, rx = pi - ry * G.IW;
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
            if (d.in_slope != 1.f) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    float f = Elt<T>::ld(&vals[q]);
                    f = f > 0.f ? f : f * d.in_slope;
                    Elt<T>::st(&vals[q], f);
                }
            }
        }
        *(u32x4*)(xt + (size_t)pi * XS + v * VEC) = *(const u32x4*)vals;
    }
}

template <typename T>
MSMC_DEV void wg_stage_g(T* gt, int XS, const msmc_conv_desc& d, const CvGeom& G, const T* gb, int c0, int qy0, int qx0,
                         int tid, int TM) {
    constexpr int VEC = Elt<T>::VEC, CKV = 64 / VEC;
    const bool vec_ok = (d.Cout % VEC) == 0;
    for (int e = tid; e < TM * CKV; e += 256) {
        const int m = e / CKV, v = e - m * CKV;
        const int mty = m / G.TW, mtx = m - mty * G.TW;
        const int qy = qy0 + mty, qx = qx0 + mtx;
        const int c = c0 + v * VEC;
        alignas(16) T vals[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) vals[q] = 0;
        if (mty < G.TH && qy < d.QH && qx < d.QW && c < d.Cout) {
            const int oy = d.oy0 + qy * d.osy, ox = d.ox0 + qx * d.osx;
            const T* src = gb + ((size_t)oy * d.Wout + ox) * d.Cout + c;
            if (vec_ok) {
                *(u32x4*)vals = *(const u32x4*)src;
            } else {
#pragma unroll
                for (int q = 0; q < VEC; ++q)
                    if (c + q < d.Cout) vals[q] = src[q];
            }
            if (d.mask_slope != 1.f) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    float f = Elt<T>::ld(&vals[q]);
                    f = f > 0.f ? f : f * d.mask_slope;
                    Elt<T>::st(&vals[q], f);
                }
            }
        }
        *(u32x4*)(gt + (size_t)m * XS + v * VEC) = *(const u32x4*)vals;
    }
}

// One tap, one 128-point tile: acc += G^T . X_t   (per wave: 32 co x 32 ci)
MSMC_DEV f32x16 wg_tap(const float* gt, const float* xt, int XS, const int* rowtab, int tapoff, int acol, int bcol, int g,
                       f32x16 acc, int TM) {
    for (int s = 0; s < TM / 2; ++s) {
        const int m = 2 * s + g;
        acc = mfma_f32_32x32x2(gt[(size_t)m * XS + acol], xt[(size_t)(rowtab[m] + tapoff) * XS + bcol], acc);
    }
    return acc;
}
MSMC_DEV bf16x8 wg_frag(const unsigned short* tile, int XS, int row0, int row1, int col) {
    u16x4 lo = lds_read_tr16(tile + (size_t)row0 * XS + col);
    u16x4 hi = lds_read_tr16(tile + (size_t)row1 * XS + col);
    u16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// staging slots per work-item of the FAST weight-gradient path: 64 channels are 8 (bf16) / 16 (fp32) 16-byte
// vectors per pixel, so the fp32 kernel needs twice the slots for the same tile
template <typename T, int TAPS> struct WgSlots {
    static constexpr int X = sizeof(T) == 2 ? (TAPS > 8 ? 6 : 12) : 12, G = sizeof(T) == 2 ? 4 : 8;
};

template <typename T, int TAPS, bool FAST>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(msmc_conv_desc d, const T* __restrict__ gptr,
                                                        float* __restrict__ dw, float* __restrict__ db, CvGeom G,
                                                        int tilesPerWg, int totalTiles, int TM) {
    MSMC_DYN_LDS(smem);
    constexpr int XS = 64 + Elt<T>::VEC;
    T* xt = (T*)smem;                                   // [IH*IW][XS]
    T* gt = xt + (size_t)G.IH * G.IW * XS;              // [TM][XS]
    int* rowtab = (int*)(gt + (size_t)TM * XS);         // [TM] X-tile pixel row of lattice point m
    const int nks = TM >> 4;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane & 31, g = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
    const bool wave_live = (co0 + 32 * wm < d.Cout) && (ci0 + 32 * wn < d.Cin);
    for (int m = tid; m < TM; m += 256) {
        int mty = m / G.TW, mtx = m - mty * G.TW;
        rowtab[m] = (mty < G.TH) ? (mty * d.isy) * G.IW + mtx * d.isx : 0;
    }
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    __syncthreads();

    // bf16: lane L of each 16-lane group addresses row (L>>2) of its 4-row block, 4 channels from (L&3)*4
    const int L = lane & 15, half = (lane >> 4) & 1;
    int xrow[16], grow[16];
    if (sizeof(T) == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 16 * (r >> 1) + 8 * g + 4 * (r & 1) + (L >> 2);
            grow[r] = m < TM ? m : 0;
            xrow[r] = m < TM ? rowtab[m] : 0;
        }
    }
    const int acol_tr = 32 * wm + 16 * half + 4 * (L & 3), bcol_tr = 32 * wn + 16 * half + 4 * (L & 3);

    if (d.dw_copies > 1) {                              // privatised accumulators: copy (split index mod R)
        const int copy = blockIdx.x % d.dw_copies;
        dw += (size_t)copy * d.ntaps * d.Cout * d.Cin;
        if (db) db += (size_t)copy * d.Cout;
    }
    const bool do_bias = (db != nullptr) && (blockIdx.z == 0);
    float bias_acc = 0.f;
    const int t0 = blockIdx.x * tilesPerWg;
    int t1 = t0 + tilesPerWg;
    if (t1 > totalTiles) t1 = totalTiles;

    // FAST: every work-item owns fixed 16-byte staging slots (tile-relative coordinates computed once); the
    // loads of tile t+1 are issued before the MFMAs of tile t and written to LDS afterwards.
    constexpr int VEC = Elt<T>::VEC, CKV = 64 / VEC, WG_XLD = WgSlots<T, TAPS>::X, WG_GLD = WgSlots<T, TAPS>::G;
    int x_ry[WG_XLD], x_rx[WG_XLD], x_dst[WG_XLD], g_m[WG_GLD], g_dst[WG_GLD];
    u32x4 xreg[WG_XLD], greg[WG_GLD];
    if (FAST) {
        const int npix = G.IH * G.IW;
#pragma unroll
        for (int j = 0; j < WG_XLD; ++j) {
            const int e = tid + 256 * j;
            x_dst[j] = -1; x_ry[j] = x_rx[j] = 0;
            if (e < npix * CKV) {
                const int pi = e / CKV, v = e - pi * CKV;
                x_ry[j] = pi / G.IW;
                x_rx[j] = pi - x_ry[j] * G.IW;
                x_dst[j] = pi * XS + v * VEC;
            }
        }
#pragma unroll
        for (int j = 0; j < WG_GLD; ++j) {
            const int e = tid + 256 * j;
            g_dst[j] = -1; g_m[j] = 0;
            if (e < TM * CKV) {
                g_m[j] = e / CKV;
                g_dst[j] = g_m[j] * XS + (e - g_m[j] * CKV) * VEC;
            }
        }
    }
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto fetch = [&](int tile) {
        int bt = tile;
        const int tx_ = bt % G.tilesX;
        bt /= G.tilesX;
        const int ty_ = bt % G.tilesY;
        const int b = bt / G.tilesY;
        const int qy0 = ty_ * G.TH, qx0 = tx_ * G.TW;
        const int iyBase = qy0 * d.isy + d.iy0 + G.dyMin, ixBase = qx0 * d.isx + d.ix0 + G.dxMin;
        const T* xb = (const T*)d.x + (size_t)b * d.Hin * d.Win * d.Cin;
        const T* gb = gptr + (size_t)b * d.Hout * d.Wout * d.Cout;
#pragma unroll
        for (int j = 0; j < WG_XLD; ++j) {
            xreg[j] = zero4;
            if (x_dst[j] < 0) continue;
            int iy = iyBase + x_ry[j], ix = ixBase + x_rx[j];
            bool inside = true;
            if (d.pad_mode == 1) {
                iy = reflect_index(iy, d.Hin);
                ix = reflect_index(ix, d.Win);
            } else {
                inside = (iy >= 0) && (iy < d.Hin) && (ix >= 0) && (ix < d.Win);
            }
            const int c = ci0 + (x_dst[j] % XS);
            if (inside && c < d.Cin) xreg[j] = *(const u32x4*)(xb + ((size_t)iy * d.Win + ix) * d.Cin + c);
        }
#pragma unroll
        for (int j = 0; j < WG_GLD; ++j) {
            greg[j] = zero4;
            if (g_dst[j] < 0) continue;
            const int m = g_m[j];
            const int mty = m / G.TW, mtx = m - mty * G.TW;
            const int qy = qy0 + mty, qx = qx0 + mtx;
            const int c = co0 + (g_dst[j] % XS);
            if (mty < G.TH && qy < d.QH && qx < d.QW && c < d.Cout) {
                const int oy = d.oy0 + qy * d.osy, ox = d.ox0 + qx * d.osx;
                greg[j] = *(const u32x4*)(gb + ((size_t)oy * d.Wout + ox) * d.Cout + c);
            }
        }
    };
    auto act = [&](u32x4 v, float slope) {
        if (slope == 1.f) return v;
        alignas(16) T vals[VEC];
        *(u32x4*)vals = v;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float f = Elt<T>::ld(&vals[q]);
            f = f > 0.f ? f : f * slope;
            Elt<T>::st(&vals[q], f);
        }
        return *(const u32x4*)vals;
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < WG_XLD; ++j)
            if (x_dst[j] >= 0) *(u32x4*)(xt + x_dst[j]) = act(xreg[j], d.in_slope);
#pragma unroll
        for (int j = 0; j < WG_GLD; ++j)
            if (g_dst[j] >= 0) *(u32x4*)(gt + g_dst[j]) = act(greg[j], d.mask_slope);
    };

    if (FAST && t0 < t1) fetch(t0);
    for (int tile = t0; tile < t1; ++tile) {
        __syncthreads();
        if (FAST) {
            commit();
        } else {
            int bt = tile;
            const int tx_ = bt % G.tilesX;
            bt /= G.tilesX;
            const int ty_ = bt % G.tilesY;
            const int b = bt / G.tilesY;
            const int qy0 = ty_ * G.TH, qx0 = tx_ * G.TW;
            const int iyBase = qy0 * d.isy + d.iy0 + G.dyMin, ixBase = qx0 * d.isx + d.ix0 + G.dxMin;
            wg_stage_x<T>(xt, XS, d, G, (const T*)d.x + (size_t)b * d.Hin * d.Win * d.Cin, ci0, iyBase, ixBase, tid);
            wg_stage_g<T>(gt, XS, d, G, gptr + (size_t)b * d.Hout * d.Wout * d.Cout, co0, qy0, qx0, tid, TM);
        }
        __syncthreads();
        if (FAST && tile + 1 < t1) fetch(tile + 1);
        if (do_bias && tid < 64) {                      // bias gradient: column sums of the g tile (fused)
            float sacc = 0.f;
            for (int m = 0; m < TM; ++m) sacc = sacc + Elt<T>::ld(gt + (size_t)m * XS + tid);
            bias_acc = bias_acc + sacc;
        }
        if (!wave_live) continue;
        if (sizeof(T) == 2) {
            // K-step outer, taps inner: one A fragment (g tile) live at a time, reused by every tap
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks >= nks) continue;
                const bf16x8 af = wg_frag((const unsigned short*)gt, XS, grow[2 * ks], grow[2 * ks + 1], acol_tr);
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    if (t < d.ntaps) {
                        const int tapoff = (d.tap_dy[t] - G.dyMin) * G.IW + (d.tap_dx[t] - G.dxMin);
                        const bf16x8 bf = wg_frag((const unsigned short*)xt, XS, xrow[2 * ks] + tapoff,
                                                  xrow[2 * ks + 1] + tapoff, bcol_tr);
                        acc[t] = mfma_bf16_32x32x16(af, bf, acc[t]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                if (t < d.ntaps) {
                    const int tapoff = (d.tap_dy[t] - G.dyMin) * G.IW + (d.tap_dx[t] - G.dxMin);
                    acc[t] = wg_tap((const float*)gt, (const float*)xt, XS, rowtab, tapoff, 32 * wm + i, 32 * wn + i, g,
                                    acc[t], TM);
                }
            }
        }
    }
    if (do_bias && tid < 64 && co0 + tid < d.Cout) atomicAdd(db + co0 + tid, bias_acc);
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
static int wg_launch(const msmc_conv_desc* d, const void* g, float* dw, float* db, msmc_stream stream) {
    constexpr int XS = 64 + Elt<T>::VEC;
    CvGeom G;
    size_t lds_unused;
    int TM = WG_TM, rc;
    size_t lds;
    for (;;) {                                     // shrink the lattice tile until halo + g tile fit LDS
        rc = cv_geometry(d, &G, sizeof(T), XS, 0, &lds_unused, TM);
        if (rc) return rc;
        TM = ((G.TH * G.TW + 15) / 16) * 16;       // e.g. 11 x 11 MPD tile -> 128 rows, the tail rows are zero
        lds = ((size_t)G.IH * G.IW + TM) * XS * sizeof(T) + TM * sizeof(int);
        if (lds <= 160 * 1024) break;
        if (G.TH * G.TW <= 16) return MSMC_E_SHAPE;
        TM = (G.TH * G.TW) / 2;
    }
    const int totalTiles = G.tilesX * G.tilesY * d->B;
    const int ctiles = ((d->Cout + 63) / 64) * ((d->Cin + 63) / 64);
    // Split of the pixel reduction over workgroups: each split costs one fp32 atomic per dW element, and atomics
    // on ONE address retire serially at ~0.1 us each, so  t(n) = (tiles/n) * t_tile + n * 0.1 us  (t_tile ~3 us)
    // is minimal at n = sqrt(30 * tiles); never more workgroups than ~2 per CU.
    int nsplit = (int)(sqrt(30.0 * totalTiles * (d->dw_copies > 1 ? d->dw_copies : 1)) + 0.5);
    if (d->split_shift > 0) nsplit <<= d->split_shift;
    else if (d->split_shift < 0) nsplit >>= -d->split_shift;
    const int cap = (2 * MSMC_NUM_CU + ctiles - 1) / ctiles;
    if (msmc_wgrad_split_override > 0) nsplit = msmc_wgrad_split_override;
    else if (nsplit > cap) nsplit = cap;
    if (nsplit > totalTiles) nsplit = totalTiles;
    if (nsplit < 1) nsplit = 1;
    const int tilesPerWg = (totalTiles + nsplit - 1) / nsplit;
    nsplit = (totalTiles + tilesPerWg - 1) / tilesPerWg;
    dim3 grid((unsigned)nsplit, (unsigned)((d->Cout + 63) / 64), (unsigned)((d->Cin + 63) / 64));
    const T* gp = (const T*)g;
    constexpr int CKVh = 64 / Elt<T>::VEC;
    const int xslots = d->ntaps <= 4 ? WgSlots<T, 4>::X : d->ntaps <= 8 ? WgSlots<T, 8>::X : WgSlots<T, 12>::X;
    const bool fast = (d->Cin % Elt<T>::VEC) == 0 && (d->Cout % Elt<T>::VEC) == 0 && d->ntaps <= 12 &&
                      (long)G.IH * G.IW * CKVh <= 256L * xslots && (long)TM * CKVh <= 256L * WgSlots<T, 4>::G &&
                      msmc_conv_pipeline_enabled;
#define WG_GO(TP)                                                                                              \
    do {                                                                                                       \
        if (fast) {                                                                                            \
            rc = msmc_allow_lds((const void*)conv_wgrad_kernel<T, TP, true>, (int)lds);                        \
            if (rc) return rc;                                                                                 \
            MSMC_LAUNCH((conv_wgrad_kernel<T, TP, true>), grid, dim3(256), lds, (msmc_stream_t)stream, *d, gp, \
                        dw, db, G, tilesPerWg, totalTiles, TM);                                                \
        } else {                                                                                               \
            rc = msmc_allow_lds((const void*)conv_wgrad_kernel<T, TP, false>, (int)lds);                       \
            if (rc) return rc;                                                                                 \
            MSMC_LAUNCH((conv_wgrad_kernel<T, TP, false>), grid, dim3(256), lds, (msmc_stream_t)stream, *d, gp,\
                        dw, db, G, tilesPerWg, totalTiles, TM);                                                \
        }                                                                                                      \
    } while (0)
    if (d->ntaps <= 4) WG_GO(4);
    else if (d->ntaps <= 8) WG_GO(8);
    else if (d->ntaps <= 12) WG_GO(12);
    else WG_GO(16);
#undef WG_GO
    msmc_conv_last = msmc_prof_name(msmc_kname("conv_wgrad_kernel", EltName<T>::v,
                                               d->ntaps <= 4 ? 4 : d->ntaps <= 8 ? 8 : d->ntaps <= 12 ? 12 : 16, fast ? 1 : 0));
    return msmc_check_launch();
}


// ------------------------------------------------------------------------------------------------
// bf16 weight gradient, second generation.  Same math and LDS operand layout as conv_wgrad_kernel, but
//   * a wave owns one 32-channel block of output channels (its A fragment, read once per 16 pixels) and up
//     to TPW (input-channel block, tap) units of it -- with 32 or fewer channels the taps are spread over
//     all four waves instead of leaving three idle, and taps beyond the per-wave budget go to another
//     workgroup (grid.z), so no wave carries more than 5 accumulators and two or three workgroups fit a CU;
//   * staging vectors are as wide as the channel count allows (2..16 bytes), and the LDS rows hold only
//     the real channels: thin layers (2, 4, 8 channels) stage kilobytes, not 64-channel padded rows;
//   * the bias gradient is accumulated from the staged registers (no LDS pass);
//   * tile coordinates come from LDS tables built once per workgroup.
// ------------------------------------------------------------------------------------------------
struct Wg2Params {
    float* ws;               // third generation: per-split partial results [nsplit][ws_stride] (dW then db), NULL = none
    long ws_stride;          // floats per split region
    int direct;              // 1: this launch owns every dW element exactly once -> plain (non-atomic) accumulation
    int TM, tilesPerWg, totalTiles;
    int XSx, XSg;            // LDS row strides (elements)
    int vex, veg;            // elements per staging vector (1, 2, 4, 8)
    int shx, shg;            // log2(staging vectors per pixel)
    int TG, ntg;             // taps per workgroup, tap groups
};

template <int VE> struct WgVec;
template <> struct WgVec<8> { typedef u32x4 type; };
template <> struct WgVec<4> { typedef u32x2 type; };
template <> struct WgVec<2> { typedef unsigned int type; };
template <> struct WgVec<1> { typedef unsigned short type; };

template <int VE, bool SUM>
MSMC_DEV typename WgVec<VE>::type wg2_act(typename WgVec<VE>::type v, float slope, float (&sums)[8]) {
    typedef typename WgVec<VE>::type V;
    if (slope == 1.f && !SUM) return v;
    alignas(16) unsigned short vals[VE];
    *(V*)vals = v;
#pragma unroll
    for (int q = 0; q < VE; ++q) {
        float f = bf16_bits_to_f32(vals[q]);
        if (slope != 1.f) {
            f = f > 0.f ? f : f * slope;
            vals[q] = f32_to_bf16_bits(f);
        }
        if (SUM) sums[q] = sums[q] + f;
    }
    return *(const V*)vals;
}

// input halo tile -> LDS rows [pixel][channel]; padding rule and input activation applied here
template <int VE>
MSMC_DEV void wg2_stage_x(unsigned short* xt, const int* xmeta, const msmc_conv_desc& d, const unsigned short* xb,
                          int ci0, int iyBase, int ixBase, int npix, int sh, int XS, int tid) {
    typedef typename WgVec<VE>::type V;
    float unused[8];
    const int nvec = npix << sh, vmask = (1 << sh) - 1;
    for (int e0 = tid; e0 < nvec; e0 += 1024) {
        V vals[4];
        int dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            dst[u] = -1;
            vals[u] = V();
            if (e < nvec) {
                const int pi = e >> sh, c = (e & vmask) * VE;
                const int meta = xmeta[pi];
                int iy = iyBase + (meta >> 16), ix = ixBase + (meta & 0xffff);
                bool inside = true;
                if (d.pad_mode == 1) {
                    iy = reflect_index(iy, d.Hin);
                    ix = reflect_index(ix, d.Win);
                } else {
                    inside = (iy >= 0) && (iy < d.Hin) && (ix >= 0) && (ix < d.Win);
                }
                if (ci0 + c < d.Cin) {
                    dst[u] = pi * XS + c;
                    if (inside) vals[u] = *(const V*)(xb + ((size_t)iy * d.Win + ix) * d.Cin + ci0 + c);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0) *(V*)(xt + dst[u]) = wg2_act<VE, false>(vals[u], d.in_slope, unused);
    }
}

// output-gradient tile -> LDS rows [lattice point][channel]; per-thread column sums feed the bias gradient
template <int VE, bool SUM>
MSMC_DEV void wg2_stage_g(unsigned short* gt, const int* gmeta, const msmc_conv_desc& d, const unsigned short* gb,
                          int co0, int qy0, int qx0, int TM, int sh, int XS, int tid, float (&sums)[8]) {
    typedef typename WgVec<VE>::type V;
    const int nvec = TM << sh, vmask = (1 << sh) - 1;
    for (int e0 = tid; e0 < nvec; e0 += 1024) {
        V vals[4];
        int dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            dst[u] = -1;
            vals[u] = V();
            if (e < nvec) {
                const int m = e >> sh, c = (e & vmask) * VE;
                const int meta = gmeta[m];
                if (co0 + c < d.Cout) {
                    dst[u] = m * XS + c;
                    const int qy = qy0 + (meta >> 16), qx = qx0 + (meta & 0xffff);
                    if (meta >= 0 && qy < d.QH && qx < d.QW) {
                        const int oy = d.oy0 + qy * d.osy, ox = d.ox0 + qx * d.osx;
                        vals[u] = *(const V*)(gb + ((size_t)oy * d.Wout + ox) * d.Cout + co0 + c);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0) *(V*)(gt + dst[u]) = wg2_act<VE, SUM>(vals[u], d.mask_slope, sums);
    }
}

template <int TPW>
MSMC_DEV void wg2_body(const msmc_conv_desc& d, const unsigned short* __restrict__ gptr, float* __restrict__ dw,
                       float* __restrict__ db, const CvGeom& G, const Wg2Params& P, const int block_x, const int block_y,
                       const int block_z) {
    MSMC_DYN_LDS(smem);
    const int npix = G.IH * G.IW, TM = P.TM, XSx = P.XSx, XSg = P.XSg;
    unsigned short* xt = (unsigned short*)smem;                  // [npix][XSx]
    unsigned short* gt = xt + (((size_t)npix * XSx + 7) & ~(size_t)7);   // [TM][XSg], 16-byte aligned
    int* xmeta = (int*)(gt + (size_t)TM * XSg);                  // [npix] (ry << 16) | rx
    int* gmeta = xmeta + npix;                                   // [TM]   (mty << 16) | mtx, -1 past the tile
    const int tid = threadIdx.x, w = wave_uniform(tid >> 6), lane = tid & 63, L = lane & 15, half = (lane >> 4) & 1;
    const int g = lane >> 5;
    const int co0 = block_y * 64;
    const int ciTile = block_z / P.ntg, tg = block_z - ciTile * P.ntg;
    const int ci0 = ciTile * 64;
    for (int pi = tid; pi < npix; pi += 256) {
        const int ry = pi / G.IW;
        xmeta[pi] = (ry << 16) | (pi - ry * G.IW);
    }
    for (int m = tid; m < TM; m += 256) {
        const int mty = m / G.TW;
        gmeta[m] = (mty < G.TH) ? ((mty << 16) | (m - mty * G.TW)) : -1;
    }
    __syncthreads();

    // ---- this wave's units: output-channel block cb, then (input-channel block, tap) pairs
    const int coLeft = d.Cout - co0, ciLeft = d.Cin - ci0;
    const int n_cb = coLeft > 32 ? 2 : 1, n_ib = ciLeft > 32 ? 2 : 1;
    const int wpc = 4 / n_cb, cb = w % n_cb, slot = w / n_cb;
    const int tap0 = tg * P.TG;
    int ntl = d.ntaps - tap0;
    if (ntl > P.TG) ntl = P.TG;
    const int nunits = n_ib * ntl;
    const int cpx = ((ciLeft > 64 ? 64 : ciLeft) + 3) & ~3, cpg = ((coLeft > 64 ? 64 : coLeft) + 3) & ~3;
    int boff[TPW], utap[TPW], uib[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int u = slot + wpc * j;
        utap[j] = -1; uib[j] = 0; boff[j] = 0;
        if (u < nunits) {
            const int ib = u % n_ib, t = tap0 + u / n_ib;
            int col = 32 * ib + 16 * half + 4 * (L & 3);
            if (col > cpx - 4) col = cpx - 4;           // thin tiles: surplus lanes re-read the last real chunk
            utap[j] = t; uib[j] = ib;
            boff[j] = ((d.tap_dy[t] - G.dyMin) * G.IW + (d.tap_dx[t] - G.dxMin)) * XSx + col;
        }
    }
    int acol = 32 * cb + 16 * half + 4 * (L & 3);
    if (acol > cpg - 4) acol = cpg - 4;
    // fragment rows: lane L of each 16-lane group addresses pixel row (L >> 2) of its 4-row block
    int xrow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = 16 * (r >> 1) + 8 * g + 4 * (r & 1) + (L >> 2);
        int meta = m < TM ? gmeta[m] : -1;
        xrow[r] = meta >= 0 ? ((meta >> 16) * d.isy * G.IW + (meta & 0xffff) * d.isx) * XSx : 0;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const bool partial = P.ws != nullptr;               // plain stores into this split's workspace region
    if (partial) {
        dw = P.ws + (size_t)block_x * P.ws_stride;
        if (db) db = dw + (size_t)d.ntaps * d.Cout * d.Cin;
    } else if (d.dw_copies > 1 && !P.direct) {          // privatised accumulators: copy (split index mod R)
        const int copy = block_x % d.dw_copies;
        dw += (size_t)copy * d.ntaps * d.Cout * d.Cin;
        if (db) db += (size_t)copy * d.Cout;
    }
    const bool do_bias = (db != nullptr) && (block_z == 0);
    float bsum[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bsum[q] = 0.f;
    const int nks = TM >> 4;
    const int t0 = block_x * P.tilesPerWg;
    int t1 = t0 + P.tilesPerWg;
    if (t1 > P.totalTiles) t1 = P.totalTiles;

    for (int tile = t0; tile < t1; ++tile) {
        int bt = tile;
        const int tx_ = bt % G.tilesX;
        bt /= G.tilesX;
        const int ty_ = bt % G.tilesY;
        const int b = bt / G.tilesY;
        const int qy0 = ty_ * G.TH, qx0 = tx_ * G.TW;
        const int iyBase = qy0 * d.isy + d.iy0 + G.dyMin, ixBase = qx0 * d.isx + d.ix0 + G.dxMin;
        const unsigned short* xb = (const unsigned short*)d.x + (size_t)b * d.Hin * d.Win * d.Cin;
        const unsigned short* gb = gptr + (size_t)b * d.Hout * d.Wout * d.Cout;
        __syncthreads();
        switch (P.vex) {
            case 8: wg2_stage_x<8>(xt, xmeta, d, xb, ci0, iyBase, ixBase, npix, P.shx, XSx, tid); break;
            case 4: wg2_stage_x<4>(xt, xmeta, d, xb, ci0, iyBase, ixBase, npix, P.shx, XSx, tid); break;
            case 2: wg2_stage_x<2>(xt, xmeta, d, xb, ci0, iyBase, ixBase, npix, P.shx, XSx, tid); break;
            default: wg2_stage_x<1>(xt, xmeta, d, xb, ci0, iyBase, ixBase, npix, P.shx, XSx, tid); break;
        }
#define WG2_STAGE_G(VE_)                                                                                      \
    do {                                                                                                      \
        if (do_bias) wg2_stage_g<VE_, true>(gt, gmeta, d, gb, co0, qy0, qx0, TM, P.shg, XSg, tid, bsum);      \
        else wg2_stage_g<VE_, false>(gt, gmeta, d, gb, co0, qy0, qx0, TM, P.shg, XSg, tid, bsum);             \
    } while (0)
        switch (P.veg) {
            case 8: WG2_STAGE_G(8); break;
            case 4: WG2_STAGE_G(4); break;
            case 2: WG2_STAGE_G(2); break;
            default: WG2_STAGE_G(1); break;
        }
#undef WG2_STAGE_G
        __syncthreads();
        if (utap[0] < 0) continue;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks >= nks) continue;
            const int m0 = 16 * ks + 8 * g + (L >> 2);
            const bf16x8 af = wg_frag(gt, 1, m0 * XSg, (m0 + 4) * XSg, acol);
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (utap[j] >= 0) {
                    const bf16x8 bf = wg_frag(xt, 1, xrow[2 * ks] + boff[j], xrow[2 * ks + 1] + boff[j], 0);
                    acc[j] = mfma_bf16_32x32x16(af, bf, acc[j]);
                }
            }
        }
    }

    if (do_bias) {
        // every work-item staged the same channel vector of each pixel it touched: combine the lanes that
        // share it, then the four waves through LDS -- ONE atomic per channel and workgroup (atomics on one
        // address retire at ~10 per microsecond on MI355X, whoever issues them)
        const int nv = 1 << P.shg, ve = P.veg;
        for (int mask = nv; mask < 64; mask <<= 1)
#pragma unroll
            for (int q = 0; q < 8; ++q) bsum[q] = bsum[q] + wave_xor(bsum[q], mask);
        float* red = (float*)smem;                      // [4][64]; the tiles are dead by now
        __syncthreads();
        if (lane < nv) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < ve) red[w * 64 + lane * ve + q] = bsum[q];
        }
        __syncthreads();
        if (tid < nv * ve && co0 + tid < d.Cout) {
            const float bs = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
            if (partial) db[co0 + tid] = bs;
            else if (P.direct) db[co0 + tid] = db[co0 + tid] + bs;
            else atomicAdd(db + co0 + tid, bs);
        }
    }
    // D fragment: row (co) = 32*cb + (r&3) + 8*(r>>2) + 4*g, col (ci) = 32*ib + (lane & 31)
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        if (utap[j] < 0) continue;
        const int ci = ci0 + 32 * uib[j] + (lane & 31);
        if (ci >= d.Cin) continue;
        float* dst = dw + (size_t)d.tap_w[utap[j]] * d.Cout * d.Cin;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * cb + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (co >= d.Cout) continue;
            float* q = dst + (size_t)co * d.Cin + ci;
            if (partial) *q = acc[j][r];
            else if (P.direct) *q = *q + acc[j][r];
            else atomicAdd(q, acc[j][r]);
        }
    }
}

// second stage of the third-generation weight gradient: the splits' partial results are summed in split order
// (bit-reproducible).  Small dW with many splits (thin layers over long signals) would leave this stage with a dozen
// workgroups, so it runs in two levels there: groups of consecutive splits are summed into `groups` intermediate
// regions (stored), then the groups are added to dW / db.  A member is one such pass: nsplit source regions of
// `stride` floats -> either an intermediate region (dst_ws) or the final dw | db pair.
template <int M>
struct WgReduceArgsT {
    int n;
    int first[M + 1];                       // first block of member k
    int eblocks[M];                         // blocks per group of member k (1024 floats each)
    const float* src[M];
    long stride[M];
    long n_dw[M];
    int nsplit[M], per_group[M], n_db[M];
    float* dst_ws[M];                       // not NULL: intermediate level, group gi stores its sums at dst_ws + gi * stride
    float* dw[M];                           // final level (one group): dw[e] += sum, db[e - n_dw] += sum
    float* db[M];
};
typedef WgReduceArgsT<MSMC_GROUP_MAX> WgReduceArgs;
// the merged second stage of a whole backward pass (msmc_conv_wgrad_reduce_pending): as many members per launch as a kernel
// argument block (4 KB) carries -- the discriminator's ~40 records went out as seven launches of six members, back to back on
// the critical chain in front of its optimizer step (profiles/r06_step_timeline_start_of_round.txt: 143 us)
#define WG_PENDING_MAX 40
typedef WgReduceArgsT<WG_PENDING_MAX> WgReduceArgsBig;
static_assert(sizeof(WgReduceArgsBig) <= 4000, "kernel argument block");
template <int M>
MSMC_DEV void wgrad_reduce_body(const WgReduceArgsT<M>& a) {
    int k = 0;
    while (k + 1 < a.n && (int)blockIdx.x >= a.first[k + 1]) ++k;
    const int id = blockIdx.x - a.first[k];
    const int gi = id / a.eblocks[k], eb = id - gi * a.eblocks[k];
    const long stride = a.stride[k], n_dw = a.n_dw[k], total = n_dw + a.n_db[k];
    const int s0 = gi * a.per_group[k];
    int S = a.nsplit[k] - s0;
    if (S > a.per_group[k]) S = a.per_group[k];
    const float* ws = a.src[k] + (size_t)s0 * stride;
    const long e0 = ((long)eb * 256 + threadIdx.x) * 4;
    if (e0 >= total || S <= 0) return;
    float* mid = a.dst_ws[k] ? a.dst_ws[k] + (size_t)gi * stride : nullptr;
    // (regions are padded to a multiple of four floats: an intermediate level may run past `total` inside them)
    if ((n_dw & 3) == 0 && (mid ? e0 + 4 <= stride : e0 + 4 <= n_dw)) {
        // (the splits are added in split order -- bit-reproducible -- but LOADED eight at a time: with one load in flight per
        //  work-item a member of 64 splits was 64 dependent memory round trips, and the pass ran at ~1 TB/s)
        // (round 6: a last batch of fewer than eight goes out together as well -- splits past the end re-read the last one and
        //  are not added; most members have 2-8 splits and ran entirely in the one-at-a-time remainder loop)
        f32x4 sum = *(const f32x4*)(ws + e0);
        for (int s_ = 1; s_ < S; s_ += 8) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sj = s_ + j < S ? s_ + j : S - 1;
                v[j] = *(const f32x4*)(ws + (size_t)sj * stride + e0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s_ + j < S) sum = sum + v[j];
        }
        if (mid) { *(f32x4*)(mid + e0) = sum; return; }
        f32x4* q = (f32x4*)(a.dw[k] + e0);
        *q = *q + sum;
        return;
    }
    for (long e = e0; e < e0 + 4 && e < total; ++e) {
        float sum = ws[e];
        for (int s_ = 1; s_ < S; ++s_) sum = sum + ws[(size_t)s_ * stride + e];
        if (mid) mid[e] = sum;
        else if (e < n_dw) a.dw[k][e] = a.dw[k][e] + sum;
        else if (a.db[k]) a.db[k][e - n_dw] = a.db[k][e - n_dw] + sum;
    }
}
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(WgReduceArgs a) { wgrad_reduce_body(a); }
__global__ __launch_bounds__(256) void conv_wgrad_reduce_pending_kernel(WgReduceArgsBig a) { wgrad_reduce_body(a); }

// plan of the second stage for one weight gradient: groups == 1 -> one level
struct Wg3Reduce {
    int groups, per_group;      // intermediate regions and splits per region (the last one may hold fewer)
};
static Wg3Reduce wg3_reduce_plan(long total, int nsplit) {
    Wg3Reduce r = {1, nsplit};
    const long echunks = (total + 1023) / 1024;
    if (nsplit >= 32 && echunks < 2 * MSMC_NUM_CU) {
        int groups = (int)((2 * MSMC_NUM_CU + echunks - 1) / echunks);
        if (groups > nsplit / 8) groups = nsplit / 8;
        if (groups > 32) groups = 32;
        if (groups > 1) {
            r.per_group = (nsplit + groups - 1) / groups;
            r.groups = (nsplit + r.per_group - 1) / r.per_group;
        }
    }
    return r;
}
// append the pass of one weight gradient at `level` (0: split groups -> intermediate regions, only when the plan has
// several groups; 1: -> dw | db) to a launch; `mid` = the intermediate regions (groups * stride floats)
// Deferred second stage (msmc_conv_wgrad_defer_begin / _end, include/msmc_hip.h): while the calling thread has a sink
// armed, the weight-gradient launchers record what their second stage would add up instead of launching it; the caller
// issues the recorded reductions of a whole backward pass together (msmc_conv_wgrad_reduce_pending).
static thread_local msmc_wg_pending* wg_defer_sink = nullptr;
static thread_local int wg_defer_cap = 0, wg_defer_n = 0;
template <int M>
static void wg3_reduce_add(WgReduceArgsT<M>& a, int* blocks, const float* ws, long stride, long n_dw, int n_db, int nsplit,
                           float* mid, float* dw, float* db, int level) {
    if (!ws) return;
    if (wg_defer_sink) {
        if (level == 0) {
            if (wg_defer_n < wg_defer_cap) {
                msmc_wg_pending& p = wg_defer_sink[wg_defer_n++];
                p.ws = ws; p.mid = mid; p.dw = dw; p.db = db;
                p.stride = stride; p.n_dw = n_dw; p.n_db = n_db; p.nsplit = nsplit;
                return;
            }
        } else {
            for (int i = 0; i < wg_defer_n; ++i)
                if (wg_defer_sink[i].ws == ws) return;      // recorded at level 0: nothing to launch now
        }
    }
    const long total = n_dw + n_db;
    const Wg3Reduce r = wg3_reduce_plan(total, nsplit);
    if (level == 0 && r.groups == 1) return;
    const int k = a.n++;
    a.first[k] = *blocks;
    a.eblocks[k] = (int)((total + 1023) / 1024);
    a.stride[k] = stride; a.n_dw[k] = n_dw; a.n_db[k] = n_db;
    if (level == 0) {
        a.src[k] = ws; a.nsplit[k] = nsplit; a.per_group[k] = r.per_group;
        a.dst_ws[k] = mid; a.dw[k] = nullptr; a.db[k] = nullptr;
        *blocks += a.eblocks[k] * r.groups;
    } else {
        a.src[k] = r.groups == 1 ? ws : mid;
        a.nsplit[k] = a.per_group[k] = r.groups == 1 ? nsplit : r.groups;
        a.dst_ws[k] = nullptr; a.dw[k] = dw; a.db[k] = db;
        *blocks += a.eblocks[k];
    }
}

template <int TPW>
__global__ __launch_bounds__(256, 2) void conv_wgrad2_kernel(msmc_conv_desc d, const unsigned short* __restrict__ gptr,
                                                            float* __restrict__ dw, float* __restrict__ db, CvGeom G,
                                                            Wg2Params P) {
    wg2_body<TPW>(d, gptr, dw, db, G, P, blockIdx.x, blockIdx.y, blockIdx.z);
}

// grouped weight gradients (see conv_gather2_group_kernel): members flattened over their (split, co tile, ci tile x taps)
struct Wg2GroupArgs {
    int n;
    int first[MSMC_GROUP_MAX + 1];
    int nx[MSMC_GROUP_MAX], ny[MSMC_GROUP_MAX];
    const unsigned short* g[MSMC_GROUP_MAX];
    float* dw[MSMC_GROUP_MAX];
    float* db[MSMC_GROUP_MAX];
    msmc_conv_desc d[MSMC_GROUP_MAX];
    CvGeom G[MSMC_GROUP_MAX];
    Wg2Params P[MSMC_GROUP_MAX];
};
template <int TPW>
__global__ __launch_bounds__(256, 2) void conv_wgrad2_group_kernel(Wg2GroupArgs a) {
    const int k = cv_group_member(a.first, a.n);
    int id = blockIdx.x - a.first[k];
    const int bx = id % a.nx[k];
    id /= a.nx[k];
    wg2_body<TPW>(a.d[k], a.g[k], a.dw[k], a.db[k], a.G[k], a.P[k], bx, id % a.ny[k], id / a.ny[k]);
}

static int wg2_vec_elems(int channels, const void* base) {
    int ve = 8;                                   // largest power of two dividing the pixel pitch and the base
    while (ve > 1 && ((channels % ve) != 0 || (((size_t)base) % (2 * ve)) != 0)) ve >>= 1;
    return ve;
}
static int wg2_row_stride(int cp) { return cp == 32 ? 48 : cp + 8; }   // rows of a 4-row transpose read on disjoint banks

struct Wg2Plan {
    Wg2Params P;
    CvGeom G;
    size_t lds;
    int tpw;
    unsigned gx, gy, gz;
    size_t ws_floats;            // third generation: workspace this launch needs (0: direct accumulation, one split)
};
#define WG3_WS_CAP_FLOATS (12L * 1024 * 1024)       // 48 MiB of partial results per launch at most
static int wg2_plan(const msmc_conv_desc* d, const void* g, Wg2Plan* pl, bool gen3 = false) {
    Wg2Params& P = pl->P;
    P.ws = nullptr;
    P.ws_stride = 0;
    P.direct = 0;
    pl->ws_floats = 0;
    const int cx = d->Cin > 64 ? 64 : d->Cin, cg = d->Cout > 64 ? 64 : d->Cout;
    P.vex = wg2_vec_elems(d->Cin, d->x);
    P.veg = wg2_vec_elems(d->Cout, g);
    P.shx = 0;
    while ((P.vex << P.shx) < cx) ++P.shx;
    P.shg = 0;
    while ((P.veg << P.shg) < cg) ++P.shg;
    P.XSx = wg2_row_stride((cx + 3) & ~3);
    P.XSg = wg2_row_stride((cg + 3) & ~3);
    CvGeom& G = pl->G;
    size_t lds_unused, lds;
    int TM = WG_TM, rc;
    for (;;) {                                     // shrink the lattice tile until two workgroups fit a CU
        rc = cv_geometry(d, &G, 2, P.XSx, 0, &lds_unused, TM);
        if (rc) return rc;
        TM = ((G.TH * G.TW + 15) / 16) * 16;
        lds = ((((size_t)G.IH * G.IW * P.XSx + 7) & ~(size_t)7) + (size_t)TM * P.XSg) * 2 +
              ((size_t)G.IH * G.IW + TM) * sizeof(int);
        if (lds <= 64 * 1024 && G.IH < 32768 && G.IW < 65536) break;
        if (G.TH * G.TW <= 16) {
            if (lds <= 160 * 1024) break;
            return MSMC_E_SHAPE;
        }
        TM = (G.TH * G.TW) / 2;
    }
    P.TM = TM;
    if (lds < 1024) lds = 1024;                  // the bias reduction reuses the first KiB
    // taps per workgroup: a wave carries at most 5 accumulators (6 would spill at two waves per SIMD)
    const int ncb = d->Cout > 32 ? 2 : 1, nib = d->Cin > 32 ? 2 : 1, wpc = 4 / ncb;
    int tgmax = msmc_wgrad_tpw_cap * wpc / nib;
    if (tgmax < 1) tgmax = 1;
    P.ntg = (d->ntaps + tgmax - 1) / tgmax;
    P.TG = (d->ntaps + P.ntg - 1) / P.ntg;
    const int tpw = (P.TG * nib + wpc - 1) / wpc;
    P.totalTiles = G.tilesX * G.tilesY * d->B;
    const int cols = ((d->Cout + 63) / 64) * ((d->Cin + 63) / 64) * P.ntg;
    // pixel split: every split adds one fp32 atomic per dW element, and atomics on ONE address retire serially at
    // ~0.1 us each (measured: 4096 per address -> 430 us), so  t(n) = (tiles/n) * t_tile + n * 0.1 us  with
    // t_tile ~3 us is minimal at n = sqrt(30 * tiles), whatever the size of dW
    int nsplit = (int)(sqrt(30.0 * P.totalTiles * (d->dw_copies > 1 ? d->dw_copies : 1)) + 0.5);
    if (d->split_shift > 0) nsplit <<= d->split_shift;
    else if (d->split_shift < 0) nsplit >>= -d->split_shift;
    const int cap = (4 * MSMC_NUM_CU + cols - 1) / cols;
    const long n_dw = (long)d->ntaps * d->Cout * d->Cin;
    const long stride = ((n_dw + d->Cout + 3) / 4) * 4;
    if (gen3) {
        // third generation: no atomics.  A split costs one plain store of its partial dW and one read in the second
        // stage, so the pixel reduction is split only as far as it takes to fill the chip (~3 workgroups per CU), and
        // never beyond the workspace cap; one split accumulates straight into dW.
        nsplit = (3 * MSMC_NUM_CU + cols - 1) / cols;
        if (d->split_shift > 0) nsplit <<= d->split_shift;
        else if (d->split_shift < 0) nsplit >>= -d->split_shift;
        const long fit = WG3_WS_CAP_FLOATS / stride;            // (the intermediate regions of a two-level second stage
        if (nsplit > fit) nsplit = (int)(fit > 1 ? fit : 1);    //  are at most an eighth on top)
    }
    if (msmc_wgrad_split_override > 0) nsplit = msmc_wgrad_split_override;
    else if (!gen3 && nsplit > cap) nsplit = cap;
    if (nsplit > P.totalTiles) nsplit = P.totalTiles;
    if (nsplit < 1) nsplit = 1;
    P.tilesPerWg = (P.totalTiles + nsplit - 1) / nsplit;
    nsplit = (P.totalTiles + P.tilesPerWg - 1) / P.tilesPerWg;
    if (gen3) {
        P.direct = nsplit == 1;
        P.ws_stride = stride;
        pl->ws_floats = nsplit > 1 ? (size_t)(nsplit + wg3_reduce_plan(n_dw + d->Cout, nsplit).groups) * stride : 0;
    }
    pl->lds = lds;
    pl->tpw = tpw <= 4 ? (tpw < 1 ? 1 : tpw) : 5;
    pl->gx = (unsigned)nsplit;
    pl->gy = (unsigned)((d->Cout + 63) / 64);
    pl->gz = (unsigned)(((d->Cin + 63) / 64) * P.ntg);
    return 0;
}

static int wg3_reduce_launch(WgReduceArgs& a, int blocks, msmc_stream stream) {
    MSMC_LAUNCH(conv_wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream, a);
    ++msmc_conv_launches;
    return msmc_check_launch();
}

extern "C" void msmc_conv_wgrad_defer_begin(msmc_wg_pending* sink, int capacity) {
    wg_defer_sink = capacity > 0 ? sink : nullptr;
    wg_defer_cap = capacity;
    wg_defer_n = 0;
}
extern "C" int msmc_conv_wgrad_defer_end(void) {
    const int n = wg_defer_n;
    wg_defer_sink = nullptr;
    wg_defer_cap = wg_defer_n = 0;
    return n;
}
extern "C" int msmc_conv_wgrad_reduce_pending(const msmc_wg_pending* items, int n, msmc_stream stream) {
    if (n < 0 || (n && !items)) return MSMC_E_SHAPE;
    msmc_wg_pending* keep = wg_defer_sink;                  // (the merged launches themselves are never deferred)
    wg_defer_sink = nullptr;
    int rc = 0;
    for (int level = 0; level < 2 && !rc; ++level) {
        int i = 0;
        while (i < n && !rc) {
            WgReduceArgsBig a;
            a.n = 0;
            int blocks = 0;
            for (; i < n && a.n < WG_PENDING_MAX; ++i) {
                const msmc_wg_pending& p = items[i];
                if (level == 1) {
                    // a layer applied twice in one backward pass (D(real) and D(fake) as separate calls, rb(rb(x))) has two
                    // records with the same accumulator: their `dw += sum` are plain read-modify-writes, so they must not
                    // share a launch -- close this one, the next is ordered after it on the stream
                    bool clash = false;
                    for (int j = 0; j < a.n && !clash; ++j)
                        clash = (a.dw[j] && a.dw[j] == p.dw) || (a.db[j] && a.db[j] == p.db);
                    if (clash) break;
                }
                wg3_reduce_add(a, &blocks, p.ws, p.stride, p.n_dw, p.n_db, p.nsplit, p.mid, p.dw, p.db, level);
            }
            if (!a.n) continue;
            a.first[a.n] = blocks;
            MSMC_LAUNCH(conv_wgrad_reduce_pending_kernel, dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream, a);
            ++msmc_conv_launches;
            rc = msmc_check_launch();
        }
    }
    wg_defer_sink = keep;
    return rc;
}

static int wg2_launch(const msmc_conv_desc* d, const void* g, float* dw, float* db, msmc_stream stream,
                      float* ws = nullptr, size_t ws_floats = 0) {
    Wg2Plan pl;
    const bool gen3 = d->variant == 3;
    int rc = wg2_plan(d, g, &pl, gen3);
    if (rc) return rc;
    if (gen3 && pl.ws_floats) {
        if (!ws || ws_floats < pl.ws_floats) return MSMC_E_WORKSPACE;
        pl.P.ws = ws;
    }
    const dim3 grid(pl.gx, pl.gy, pl.gz);
    const size_t lds = pl.lds;
    const unsigned short* gp = (const unsigned short*)g;
#define WG2_GO(TP)                                                                                           \
    do {                                                                                                     \
        rc = msmc_allow_lds((const void*)conv_wgrad2_kernel<TP>, (int)lds);                                  \
        if (rc) return rc;                                                                                   \
        MSMC_LAUNCH((conv_wgrad2_kernel<TP>), grid, dim3(256), lds, (msmc_stream_t)stream, *d, gp, dw, db, pl.G, pl.P); \
    } while (0)
    if (pl.tpw == 1) WG2_GO(1);
    else if (pl.tpw == 2) WG2_GO(2);
    else if (pl.tpw == 3) WG2_GO(3);
    else if (pl.tpw == 4) WG2_GO(4);
    else WG2_GO(5);
#undef WG2_GO
    msmc_conv_last = msmc_prof_name(msmc_kname("conv_wgrad2_kernel", nullptr, pl.tpw, -1));
    rc = msmc_check_launch();
    if (rc || !pl.P.ws) return rc;
    const long n_dw = (long)d->ntaps * d->Cout * d->Cin;
    float* mid = ws + (size_t)pl.gx * pl.P.ws_stride;
    for (int level = 0; level < 2; ++level) {
        WgReduceArgs a;
        a.n = 0;
        int blocks = 0;
        wg3_reduce_add(a, &blocks, ws, pl.P.ws_stride, n_dw, db ? d->Cout : 0, (int)pl.gx, mid, dw, db, level);
        if (!a.n) continue;
        a.first[a.n] = blocks;
        rc = wg3_reduce_launch(a, blocks, stream);
        if (rc) return rc;
    }
    return 0;
}

#include "wgrad4.inc"

// fourth generation (variants 4 / 5 / 6, see wg4_plan): MSMC_E_SHAPE where it does not apply
static int wg4_launch(const msmc_conv_desc* d, const void* g, float* dw, float* db, msmc_stream stream, float* ws,
                      size_t ws_floats) {
    Wg4Plan pl;
    int rc = wg4_plan(d, g, &pl, d->variant - 4);
    if (rc) return rc;
    if (pl.ws_floats) {
        if (!ws || ws_floats < pl.ws_floats) return MSMC_E_WORKSPACE;
        pl.P.ws = ws;
    }
    const dim3 grid(pl.gx, pl.gy, pl.gz);
    const unsigned short* gp = (const unsigned short*)g;
#define WG4_GO(TP, DD)                                                                                       \
    do {                                                                                                     \
        rc = msmc_allow_lds((const void*)conv_wgrad4_kernel<TP, DD>, (int)pl.lds);                           \
        if (rc) return rc;                                                                                   \
        MSMC_LAUNCH((conv_wgrad4_kernel<TP, DD>), grid, dim3(256), pl.lds, (msmc_stream_t)stream, *d, gp, dw, db, pl.P); \
    } while (0)
    const bool ahead2 = d->variant == 4;
    if (pl.tpw == 1) { if (ahead2) WG4_GO(1, 2); else WG4_GO(1, 1); }
    else if (pl.tpw == 2) { if (ahead2) WG4_GO(2, 2); else WG4_GO(2, 1); }
    else if (pl.tpw == 3) { if (ahead2) WG4_GO(3, 2); else WG4_GO(3, 1); }
    else if (pl.tpw == 4) { if (ahead2) WG4_GO(4, 2); else WG4_GO(4, 1); }
    else WG4_GO(5, 1);
#undef WG4_GO
    msmc_conv_last = msmc_prof_name(msmc_kname("conv_wgrad4_kernel", nullptr, pl.tpw, (ahead2 && pl.tpw < 5) ? 2 : 1));
    rc = msmc_check_launch();
    if (rc || !pl.P.ws) return rc;
    const long n_dw = (long)d->ntaps * d->Cout * d->Cin;
    float* mid = ws + (size_t)pl.gx * pl.P.ws_stride;
    for (int level = 0; level < 2; ++level) {
        WgReduceArgs a;
        a.n = 0;
        int blocks = 0;
        wg3_reduce_add(a, &blocks, ws, pl.P.ws_stride, n_dw, db ? d->Cout : 0, (int)pl.gx, mid, dw, db, level);
        if (!a.n) continue;
        a.first[a.n] = blocks;
        rc = wg3_reduce_launch(a, blocks, stream);
        if (rc) return rc;
    }
    return 0;
}

#include "wgrad5.inc"
#include "wgrad6.inc"
#include "wgrad7.inc"

// general-lattice weight gradient with LDS-DMA staging (variant 7: interpreter-tested, not yet timed on the GPU)
static int wg5_launch(const msmc_conv_desc* d, const void* g, float* dw, float* db, msmc_stream stream, float* ws,
                      size_t ws_floats) {
    Wg5Plan pl;
    int rc = wg5_plan(d, g, &pl);
    if (rc) return rc;
    if (pl.ws_floats) {
        if (!ws || ws_floats < pl.ws_floats) return MSMC_E_WORKSPACE;
        pl.P.ws = ws;
    }
    const dim3 grid(pl.gx, pl.gy, pl.gz);
    const unsigned short* gp = (const unsigned short*)g;
#define WG5_GO(TP)                                                                                           \
    do {                                                                                                     \
        rc = msmc_allow_lds((const void*)conv_wgrad5_kernel<TP>, (int)pl.lds);                               \
        if (rc) return rc;                                                                                   \
        MSMC_LAUNCH((conv_wgrad5_kernel<TP>), grid, dim3(256), pl.lds, (msmc_stream_t)stream, *d, gp, dw, db, pl.G, pl.P); \
    } while (0)
    if (pl.tpw == 1) WG5_GO(1);
    else if (pl.tpw == 2) WG5_GO(2);
    else if (pl.tpw == 3) WG5_GO(3);
    else if (pl.tpw == 4) WG5_GO(4);
    else WG5_GO(5);
#undef WG5_GO
    msmc_conv_last = msmc_prof_name(msmc_kname("conv_wgrad5_kernel", nullptr, pl.tpw, -1));
    rc = msmc_check_launch();
    if (rc || !pl.P.ws) return rc;
    const long n_dw = (long)d->ntaps * d->Cout * d->Cin;
    float* mid = ws + (size_t)pl.gx * pl.P.ws_stride;
    for (int level = 0; level < 2; ++level) {
        WgReduceArgs a;
        a.n = 0;
        int blocks = 0;
        wg3_reduce_add(a, &blocks, ws, pl.P.ws_stride, n_dw, db ? d->Cout : 0, (int)pl.gx, mid, dw, db, level);
        if (!a.n) continue;
        a.first[a.n] = blocks;
        rc = wg3_reduce_launch(a, blocks, stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int msmc_conv_wgrad_ws(const msmc_conv_desc* d, const void* g, float* dw, float* db, void* workspace,
                                  size_t workspace_bytes, msmc_stream stream) {
    if (!d || !g || !dw || d->B <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->QH <= 0 || d->QW <= 0) return MSMC_E_SHAPE;
    if (d->ntaps <= 0 || d->ntaps > MSMC_CONV_MAX_TAPS) return MSMC_E_SHAPE;
    ++msmc_conv_launches;
    if (d->dtype == 0) return d->variant == 3 ? MSMC_E_SHAPE : wg_launch<float>(d, g, dw, db, stream);
    if (d->dtype == 1) {
        const int gen = d->variant > 0 ? d->variant : msmc_wgrad_generation;
        if (gen == 1) return wg_launch<unsigned short>(d, g, dw, db, stream);
        msmc_conv_desc e = *d;
        e.variant = gen;                                      // (the generation switch selects the third one too)
        if (gen == 7) return wg5_launch(&e, g, dw, db, stream, (float*)workspace, workspace_bytes / sizeof(float));
        if (gen == 8) return wg6_launch(&e, g, dw, db, stream);      // direct thin-layer kernel (E_SHAPE outside its scope)
        if (gen == 9) return wg7_launch(&e, g, dw, db, stream, (float*)workspace, workspace_bytes / sizeof(float));   // 128 x 128 channel tiles
        if (gen >= 4) {
            if (gen > 6) return MSMC_E_SHAPE;
            const int rc = wg4_launch(&e, g, dw, db, stream, (float*)workspace, workspace_bytes / sizeof(float));
            if (rc != MSMC_E_SHAPE || d->variant > 0) return rc;
            e.variant = 3;                                    // forced through the global switch: third generation elsewhere
        }
        return wg2_launch(&e, g, dw, db, stream, (float*)workspace, workspace_bytes / sizeof(float));
    }
    return MSMC_E_SHAPE;
}
extern "C" int msmc_conv_wgrad(const msmc_conv_desc* d, const void* g, float* dw, float* db, msmc_stream stream) {
    return msmc_conv_wgrad_ws(d, g, dw, db, nullptr, 0, stream);
}
extern "C" size_t msmc_conv_wgrad_workspace(const msmc_conv_desc* d, const void* g) {
    if (!d || d->dtype != 1) return 0;
    const int gen = d->variant > 0 ? d->variant : msmc_wgrad_generation;
    size_t need4 = 0;
    if (gen == 7) {
        Wg5Plan p5;
        return wg5_plan(d, g, &p5) == 0 ? p5.ws_floats * sizeof(float) : 0;
    }
    if (gen == 9) {
        Wg4Plan p7;
        return wg7_plan(d, g, &p7) == 0 ? p7.ws_floats * sizeof(float) : 0;
    }
    if (gen >= 4 && gen <= 6) {                     // (inside a shared grid the member runs as third generation: the larger)
        Wg4Plan p4;
        if (wg4_plan(d, g, &p4, gen - 4) == 0) need4 = p4.ws_floats * sizeof(float);
    } else if (gen != 3) {
        return 0;
    }
    Wg2Plan pl;
    if (wg2_plan(d, g, &pl, true)) return need4;
    const size_t need3 = pl.ws_floats * sizeof(float);
    return need3 > need4 ? need3 : need4;
}

// n independent weight gradients (msmc_conv_wgrad semantics each): bf16 second- / third-generation members share grids.
// Third-generation members (variant 3) take consecutive regions of the workspace; one grouped second-stage launch
// folds the partial results of all of them.
// group4 != 0: fourth-generation members (variants 4 / 5 / 6 inside wgrad4.inc's scope) share grids of their own kernel
// (conv_wgrad4_group_kernel, every member planned for its share of the chip); 0: they join the shared grid of the
// second / third generation as third-generation members.  The host layer times both against one launch per member.
extern "C" int msmc_conv_wgrad_group_ws4(const msmc_conv_desc* descs, const void* const* g, float* const* dw,
                                         float* const* db, int n, void* workspace, size_t workspace_bytes,
                                         msmc_stream stream, int group4) {
    if (!descs || !g || !dw || n <= 0 || n > MSMC_GROUP_LIMIT) return MSMC_E_SHAPE;
    Wg2Plan plans[MSMC_GROUP_LIMIT];
    bool pending[MSMC_GROUP_LIMIT];
    float* wsp = (float*)workspace;
    size_t ws_left = workspace_bytes / sizeof(float);
    if (group4 && msmc_conv_grouping && n > 1) {                // general-lattice LDS-DMA members (variant 7) first
        Wg5Plan p5[MSMC_GROUP_LIMIT];
        bool mine5[MSMC_GROUP_LIMIT], took5[MSMC_GROUP_LIMIT];
        int count5 = 0;
        for (int i = 0; i < n; ++i) {
            const msmc_conv_desc* d = &descs[i];
            const int gen = d->variant > 0 ? d->variant : msmc_wgrad_generation;
            took5[i] = false;
            mine5[i] = d->dtype == 1 && gen == 7 && g[i] && dw[i] && d->B > 0 && d->ntaps > 0 && d->ntaps <= MSMC_CONV_MAX_TAPS;
            if (mine5[i]) ++count5;
        }
        if (count5 > 1) {
            const int share = count5 < MSMC_GROUP_MAX ? count5 : MSMC_GROUP_MAX;
            for (int i = 0; i < n; ++i) {
                if (!mine5[i]) continue;
                if (wg5_plan(&descs[i], g[i], &p5[i], share)) { mine5[i] = false; continue; }
                if (p5[i].ws_floats) {
                    if (p5[i].ws_floats > ws_left) return MSMC_E_WORKSPACE;
                    p5[i].P.ws = wsp;
                    wsp += p5[i].ws_floats;
                    ws_left -= p5[i].ws_floats;
                }
            }
            for (int i = 0; i < n; ++i) {
                if (!mine5[i]) continue;
                Wg5GroupArgs a;
                a.n = 0;
                int members[MSMC_GROUP_MAX], nmembers = 0;
                int blocks = 0, tpw = 1;
                size_t lds = 0;
                for (int j = i; j < n && a.n < MSMC_GROUP_MAX; ++j) {
                    if (!mine5[j]) continue;
                    const int k = a.n++;
                    a.first[k] = blocks;
                    a.nx[k] = (int)p5[j].gx;
                    a.ny[k] = (int)p5[j].gy;
                    a.g[k] = (const unsigned short*)g[j];
                    a.dw[k] = dw[j];
                    a.db[k] = db ? db[j] : nullptr;
                    a.d[k] = descs[j];
                    a.G[k] = p5[j].G;
                    a.P[k] = p5[j].P;
                    blocks += (int)(p5[j].gx * p5[j].gy * p5[j].gz);
                    if (p5[j].lds > lds) lds = p5[j].lds;
                    if (p5[j].tpw > tpw) tpw = p5[j].tpw;
                    if (p5[j].P.ws) members[nmembers++] = j;
                    mine5[j] = false;
                    took5[j] = true;
                }
                a.first[a.n] = blocks;
                ++msmc_conv_launches;
                int rc;
                const dim3 grid((unsigned)blocks);
#define WG5G_GO(TP)                                                                                          \
    do {                                                                                                     \
        rc = msmc_allow_lds((const void*)conv_wgrad5_group_kernel<TP>, (int)lds);                            \
        if (rc) return rc;                                                                                   \
        MSMC_LAUNCH((conv_wgrad5_group_kernel<TP>), grid, dim3(256), lds, (msmc_stream_t)stream, a);         \
    } while (0)
                if (tpw == 1) WG5G_GO(1);
                else if (tpw == 2) WG5G_GO(2);
                else if (tpw == 3) WG5G_GO(3);
                else if (tpw == 4) WG5G_GO(4);
                else WG5G_GO(5);
#undef WG5G_GO
                msmc_conv_last = msmc_prof_name(msmc_kname("conv_wgrad5_group_kernel", nullptr, tpw, -1));
                rc = msmc_check_launch();
                if (rc) return rc;
                for (int level = 0; level < 2 && nmembers; ++level) {
                    WgReduceArgs r;
                    r.n = 0;
                    int rblocks = 0;
                    for (int q = 0; q < nmembers; ++q) {
                        const int j = members[q];
                        const msmc_conv_desc& dj = descs[j];
                        float* wsj = p5[j].P.ws;
                        wg3_reduce_add(r, &rblocks, wsj, p5[j].P.ws_stride, (long)dj.ntaps * dj.Cout * dj.Cin,
                                       (db && db[j]) ? dj.Cout : 0, (int)p5[j].gx,
                                       wsj + (size_t)p5[j].gx * p5[j].P.ws_stride, dw[j], db ? db[j] : nullptr, level);
                    }
                    if (!r.n) continue;
                    r.first[r.n] = rblocks;
                    rc = wg3_reduce_launch(r, rblocks, stream);
                    if (rc) return rc;
                }
            }
            int nrest = 0;
            msmc_conv_desc rest_d[MSMC_GROUP_LIMIT];
            const void* rest_g[MSMC_GROUP_LIMIT];
            float* rest_dw[MSMC_GROUP_LIMIT];
            float* rest_db[MSMC_GROUP_LIMIT];
            for (int i = 0; i < n; ++i) {
                if (took5[i]) continue;
                rest_d[nrest] = descs[i];
                rest_g[nrest] = g[i];
                rest_dw[nrest] = dw[i];
                rest_db[nrest] = db ? db[i] : nullptr;
                ++nrest;
            }
            if (!nrest) return 0;
            return msmc_conv_wgrad_group_ws4(rest_d, rest_g, rest_dw, rest_db, nrest, wsp, ws_left * sizeof(float), stream, 1);
        }
    }
    if (group4 && msmc_conv_grouping && n > 1) {
        Wg4Plan p4[MSMC_GROUP_LIMIT];
        bool mine[MSMC_GROUP_LIMIT], took[MSMC_GROUP_LIMIT];
        int count = 0;
        for (int i = 0; i < n; ++i) {
            const msmc_conv_desc* d = &descs[i];
            const int gen = d->variant > 0 ? d->variant : msmc_wgrad_generation;
            took[i] = false;
            mine[i] = d->dtype == 1 && gen >= 4 && gen <= 6 && g[i] && dw[i] && d->B > 0 && d->ntaps > 0 &&
                      d->ntaps <= MSMC_CONV_MAX_TAPS;
            if (mine[i]) ++count;
        }
        if (count > 1) {
            const int share = count < MSMC_GROUP_MAX ? count : MSMC_GROUP_MAX;
            for (int i = 0; i < n; ++i) {
                if (!mine[i]) continue;
                const msmc_conv_desc* d = &descs[i];
                const int gen = d->variant > 0 ? d->variant : msmc_wgrad_generation;
                if (wg4_plan(d, g[i], &p4[i], gen - 4, share)) { mine[i] = false; continue; }
                if (p4[i].ws_floats) {
                    if (p4[i].ws_floats > ws_left) return MSMC_E_WORKSPACE;
                    p4[i].P.ws = wsp;
                    wsp += p4[i].ws_floats;
                    ws_left -= p4[i].ws_floats;
                }
            }
            for (int i = 0; i < n; ++i) {
                if (!mine[i]) continue;
                Wg4GroupArgs a;
                a.n = 0;
                int members[MSMC_GROUP_MAX], nmembers = 0;
                int blocks = 0, tpw = 1;
                size_t lds = 0;
                for (int j = i; j < n && a.n < MSMC_GROUP_MAX; ++j) {
                    if (!mine[j]) continue;
                    const int k = a.n++;
                    a.first[k] = blocks;
                    a.nx[k] = (int)p4[j].gx;
                    a.ny[k] = (int)p4[j].gy;
                    a.g[k] = (const unsigned short*)g[j];
                    a.dw[k] = dw[j];
                    a.db[k] = db ? db[j] : nullptr;
                    a.d[k] = descs[j];
                    a.P[k] = p4[j].P;
                    blocks += (int)(p4[j].gx * p4[j].gy * p4[j].gz);
                    if (p4[j].lds > lds) lds = p4[j].lds;
                    if (p4[j].tpw > tpw) tpw = p4[j].tpw;      // the widest member sets the accumulator budget
                    if (p4[j].P.ws) members[nmembers++] = j;
                    mine[j] = false;
                    took[j] = true;
                }
                a.first[a.n] = blocks;
                ++msmc_conv_launches;
                int rc;
                const dim3 grid((unsigned)blocks);
#define WG4G_GO(TP)                                                                                          \
    do {                                                                                                     \
        rc = msmc_allow_lds((const void*)conv_wgrad4_group_kernel<TP, 1>, (int)lds);                         \
        if (rc) return rc;                                                                                   \
        MSMC_LAUNCH((conv_wgrad4_group_kernel<TP, 1>), grid, dim3(256), lds, (msmc_stream_t)stream, a);      \
    } while (0)
                if (tpw == 1) WG4G_GO(1);
                else if (tpw == 2) WG4G_GO(2);
                else if (tpw == 3) WG4G_GO(3);
                else if (tpw == 4) WG4G_GO(4);
                else WG4G_GO(5);
#undef WG4G_GO
                msmc_conv_last = msmc_prof_name(msmc_kname("conv_wgrad4_group_kernel", nullptr, tpw, 1));
                rc = msmc_check_launch();
                if (rc) return rc;
                for (int level = 0; level < 2 && nmembers; ++level) {
                    WgReduceArgs r;
                    r.n = 0;
                    int rblocks = 0;
                    for (int q = 0; q < nmembers; ++q) {
                        const int j = members[q];
                        const msmc_conv_desc& dj = descs[j];
                        float* wsj = p4[j].P.ws;
                        wg3_reduce_add(r, &rblocks, wsj, p4[j].P.ws_stride, (long)dj.ntaps * dj.Cout * dj.Cin,
                                       (db && db[j]) ? dj.Cout : 0, (int)p4[j].gx,
                                       wsj + (size_t)p4[j].gx * p4[j].P.ws_stride, dw[j], db ? db[j] : nullptr, level);
                    }
                    if (!r.n) continue;
                    r.first[r.n] = rblocks;
                    rc = wg3_reduce_launch(r, rblocks, stream);
                    if (rc) return rc;
                }
            }
            // the rest of the call: everything the fourth generation did not take
            int nrest = 0;
            msmc_conv_desc rest_d[MSMC_GROUP_LIMIT];
            const void* rest_g[MSMC_GROUP_LIMIT];
            float* rest_dw[MSMC_GROUP_LIMIT];
            float* rest_db[MSMC_GROUP_LIMIT];
            for (int i = 0; i < n; ++i) {
                if (took[i]) continue;
                rest_d[nrest] = descs[i];
                rest_g[nrest] = g[i];
                rest_dw[nrest] = dw[i];
                rest_db[nrest] = db ? db[i] : nullptr;
                ++nrest;
            }
            if (!nrest) return 0;
            return msmc_conv_wgrad_group_ws4(rest_d, rest_g, rest_dw, rest_db, nrest, wsp, ws_left * sizeof(float), stream, 0);
        }
    }
    for (int i = 0; i < n; ++i) {
        const msmc_conv_desc* d = &descs[i];
        pending[i] = false;
        const int gen = d->variant > 0 ? d->variant : msmc_wgrad_generation;
        // (fourth-generation members join a shared grid as third-generation members: the host layer times the shared
        //  grid against one launch per member, where each runs the kernel of its own choice)
        if (!msmc_conv_grouping || d->dtype != 1 || gen == 1 || gen == 7 || gen == 8 || gen == 9 || n == 1) {
            size_t need = msmc_conv_wgrad_workspace(d, g[i]) / sizeof(float);
            if (need > ws_left) return MSMC_E_WORKSPACE;
            int rc = msmc_conv_wgrad_ws(d, g[i], dw[i], db ? db[i] : nullptr, wsp, need * sizeof(float), stream);
            if (rc) return rc;
            wsp += need;
            ws_left -= need;
            continue;
        }
        if (!g[i] || !dw[i] || d->B <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->QH <= 0 || d->QW <= 0 || d->ntaps <= 0 ||
            d->ntaps > MSMC_CONV_MAX_TAPS)
            return MSMC_E_SHAPE;
        int rc = wg2_plan(d, g[i], &plans[i], gen >= 3);
        if (rc) return rc;
        if (plans[i].ws_floats) {
            if (plans[i].ws_floats > ws_left) return MSMC_E_WORKSPACE;
            plans[i].P.ws = wsp;
            wsp += plans[i].ws_floats;
            ws_left -= plans[i].ws_floats;
        }
        pending[i] = true;
    }
    for (int i = 0; i < n; ++i) {
        if (!pending[i]) continue;
        Wg2GroupArgs a;
        a.n = 0;
        int members[MSMC_GROUP_MAX], nmembers = 0;            // third-generation members of this launch
        int blocks = 0, tpw = 1;
        size_t lds = 0;
        for (int j = i; j < n && a.n < MSMC_GROUP_MAX; ++j) {
            if (!pending[j]) continue;
            const int k = a.n++;
            a.first[k] = blocks;
            a.nx[k] = (int)plans[j].gx;
            a.ny[k] = (int)plans[j].gy;
            a.g[k] = (const unsigned short*)g[j];
            a.dw[k] = dw[j];
            a.db[k] = db ? db[j] : nullptr;
            a.d[k] = descs[j];
            a.G[k] = plans[j].G;
            a.P[k] = plans[j].P;
            blocks += (int)(plans[j].gx * plans[j].gy * plans[j].gz);
            if (plans[j].lds > lds) lds = plans[j].lds;
            if (plans[j].tpw > tpw) tpw = plans[j].tpw;        // the widest member sets the accumulator budget
            if (plans[j].P.ws) members[nmembers++] = j;
            pending[j] = false;
        }
        a.first[a.n] = blocks;
        ++msmc_conv_launches;
        int rc;
        const dim3 grid((unsigned)blocks);
#define WG2G_GO(TP)                                                                                          \
    do {                                                                                                     \
        rc = msmc_allow_lds((const void*)conv_wgrad2_group_kernel<TP>, (int)lds);                            \
        if (rc) return rc;                                                                                   \
        MSMC_LAUNCH((conv_wgrad2_group_kernel<TP>), grid, dim3(256), lds, (msmc_stream_t)stream, a);          \
    } while (0)
        if (tpw == 1) WG2G_GO(1);
        else if (tpw == 2) WG2G_GO(2);
        else if (tpw == 3) WG2G_GO(3);
        else if (tpw == 4) WG2G_GO(4);
        else WG2G_GO(5);
#undef WG2G_GO
        msmc_conv_last = msmc_prof_name(msmc_kname("conv_wgrad2_group_kernel", nullptr, tpw, -1));
        rc = msmc_check_launch();
        if (rc) return rc;
        for (int level = 0; level < 2 && nmembers; ++level) {
            WgReduceArgs r;
            r.n = 0;
            int rblocks = 0;
            for (int q = 0; q < nmembers; ++q) {
                const int j = members[q];
                const msmc_conv_desc& dj = descs[j];
                float* wsj = plans[j].P.ws;
                wg3_reduce_add(r, &rblocks, wsj, plans[j].P.ws_stride, (long)dj.ntaps * dj.Cout * dj.Cin,
                               (db && db[j]) ? dj.Cout : 0, (int)plans[j].gx, wsj + (size_t)plans[j].gx * plans[j].P.ws_stride,
                               dw[j], db ? db[j] : nullptr, level);
            }
            if (!r.n) continue;
            r.first[r.n] = rblocks;
            rc = wg3_reduce_launch(r, rblocks, stream);
            if (rc) return rc;
        }
    }
    return 0;
}
extern "C" int msmc_conv_wgrad_group_ws(const msmc_conv_desc* descs, const void* const* g, float* const* dw,
                                        float* const* db, int n, void* workspace, size_t workspace_bytes,
                                        msmc_stream stream) {
    return msmc_conv_wgrad_group_ws4(descs, g, dw, db, n, workspace, workspace_bytes, stream, 0);
}
extern "C" int msmc_conv_wgrad_group(const msmc_conv_desc* descs, const void* const* g, float* const* dw, float* const* db,
                                     int n, msmc_stream stream) {
    return msmc_conv_wgrad_group_ws(descs, g, dw, db, n, nullptr, 0, stream);
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

// sum over the 256 work-items of a workgroup: wave reduction by lane exchange, then four partial sums through LDS
MSMC_DEV float block_sum_fast(float v, float* red4) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + wave_xor(v, m);
    const int w = threadIdx.x >> 6;
    __syncthreads();                                   // (red4 may still be read from a previous call)
    if ((threadIdx.x & 63) == 0) red4[w] = v;
    __syncthreads();
    return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

// the same for workgroups of one to four waves (blockDim.x / 64)
MSMC_DEV float block_sum_waves(float v, float* red4) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + wave_xor(v, m);
    const int w = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red4[w] = v;
    __syncthreads();
    float s = red4[0];
    for (int q = 1; q < nw; ++q) s = s + red4[q];
    return s;
}

// Row pass: one workgroup per normalised row a (n = Bc * T parameters, contiguous).  The row is read once in its own
// order (sum of squares), then written to layout 1 TAP-OUTER: for a fixed tap the Bc elements of the row are consecutive
// in layout 1 (s1[2] = 1 for every layer the banks build), so a wave stores 64 consecutive elements; the strided re-read
// of the row hits L1.  No per-element integer division (the previous form spent ~60 instructions per parameter on it).
__global__ __launch_bounds__(256) void wn_prepare_kernel(const msmc_wn_item* __restrict__ items, int nitems, int skip2) {
    __shared__ float red[4];
    const msmc_wn_item it = items[wn_find(items, nitems, blockIdx.x)];
    const int a = blockIdx.x - it.block0;
    const int n = it.Bc * it.T, T = it.T, Bc = it.Bc;
    const float* v = it.v + (size_t)a * n;
    float scale = 1.f;
    if (it.g) {                                        // weight norm; g == NULL: plain weight, layout conversion only
        float ss = 0.f;
        if ((n & 3) == 0) {
            const f32x4* v4 = (const f32x4*)v;         // (rows of 4k floats off a 16-byte aligned parameter)
            for (int e = threadIdx.x; e < (n >> 2); e += 256) {
                const f32x4 q = v4[e];
                ss = fmaf(q[0], q[0], ss);
                ss = fmaf(q[1], q[1], ss);
                ss = fmaf(q[2], q[2], ss);
                ss = fmaf(q[3], q[3], ss);
            }
        } else {
            for (int e = threadIdx.x; e < n; e += 256) ss = fmaf(v[e], v[e], ss);
        }
        ss = block_sum_fast(ss, red);
        const float norm = sqrtf(ss);
        scale = it.g[a] / norm;
        if (threadIdx.x == 0) it.inv_norm[a] = 1.f / norm;
    }
    const bool two = it.dst2 && !skip2;
    // (restrict-qualified locals: without them every load of v has to stay behind the previous store to the layout buffers
    //  -- the compiler cannot know they do not overlap -- and the loop runs one memory round trip per element)
    const float* __restrict__ vr = v;
    if (it.dtype == 0) {
        float* __restrict__ d1 = (float*)it.dst1;
        float* __restrict__ d2 = (float*)it.dst2;
        for (int t = 0; t < T; ++t) {
            const long o1 = t * it.s1[0] + a * it.s1[1], o2 = t * it.s2[0] + a * it.s2[1];
            for (int b = threadIdx.x; b < Bc; b += 256) {
                const float wv = vr[b * T + t] * scale;
                d1[o1 + b * it.s1[2]] = wv;
                if (two) d2[o2 + b * it.s2[2]] = wv;
            }
        }
    } else {
        unsigned short* __restrict__ d1 = (unsigned short*)it.dst1;
        unsigned short* __restrict__ d2 = (unsigned short*)it.dst2;
        for (int t = 0; t < T; ++t) {
            const long o1 = t * it.s1[0] + a * it.s1[1], o2 = t * it.s2[0] + a * it.s2[1];
            for (int b = threadIdx.x; b < Bc; b += 256) {
                const unsigned short wv = f32_to_bf16_bits(vr[b * T + t] * scale);
                d1[o1 + b * it.s1[2]] = wv;
                if (two) d2[o2 + b * it.s2[2]] = wv;
            }
        }
    }
}

// Layout 2 is the transpose of the parameter's own order (the normalised axis a runs fastest): written row by row from
// wn_prepare_kernel it is one 2-byte store per cache line.  Here a workgroup owns a tile of 64 rows (a) x 16 columns (b),
// all taps: the parameter is read in its own order (contiguous 16*T floats per row) into LDS and written out with a
// fastest -- 64 consecutive elements per store.  Runs after wn_prepare_kernel (inv_norm); item i owns tile-blocks
// [tblock0, tblock0 + ceil(A/64) * ceil(Bc/16)).  Index walks are incremental (no per-element division).
#define WN_TA 64
#define WN_TB 16
MSMC_DEV int wn_find_tile(const msmc_wn_item* items, int n, int blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (items[mid].tblock0 <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

#define WN_TC 128                                      // columns (b, t) of a tile staged at once: 33 KB of LDS, four workgroups per CU
__global__ __launch_bounds__(256) void wn_transpose_kernel(const msmc_wn_item* __restrict__ items, int nitems) {
    __shared__ float tile[WN_TA * (WN_TC + 1)];
    __shared__ float scl[WN_TA];
    const msmc_wn_item it = items[wn_find_tile(items, nitems, blockIdx.x)];
    const int tb = blockIdx.x - it.tblock0;
    const int nbt = (it.Bc + WN_TB - 1) / WN_TB;
    const int a0 = (tb / nbt) * WN_TA, b0 = (tb - (tb / nbt) * nbt) * WN_TB;
    const int T = it.T, pitch = WN_TC + 1;
    const int nb = it.Bc - b0 < WN_TB ? it.Bc - b0 : WN_TB, na = it.A - a0 < WN_TA ? it.A - a0 : WN_TA;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ncol = nb * T;
    if (threadIdx.x < WN_TA)
        scl[threadIdx.x] = (it.g && (int)threadIdx.x < na) ? it.g[a0 + threadIdx.x] * it.inv_norm[a0 + threadIdx.x] : 1.f;
    int b = 0, t = w;                                  // column c = b * T + t of this wave's next store
    while (t >= T) { t -= T; ++b; }
    for (int c0 = 0; c0 < ncol; c0 += WN_TC) {
        const int nc = ncol - c0 < WN_TC ? ncol - c0 : WN_TC;
        __syncthreads();                               // (previous chunk's reads done; scl visible)
        for (int r = w; r < na; r += 4) {              // a wave reads nc consecutive floats of one row
            const float* src = it.v + ((size_t)(a0 + r) * it.Bc + b0) * T + c0;
            for (int c = lane; c < nc; c += 64) tile[r * pitch + c] = src[c];
        }
        __syncthreads();
        // lane = row: 64 consecutive a per store; wave w takes columns w, w + 4, .. of the chunk (WN_TC % 4 == 0, so
        // the walk of (b, t) carries over from chunk to chunk)
        const float sc = lane < na ? scl[lane] : 0.f;
        for (int c = w; c < nc; c += 4) {
            if (lane < na)
                wn_store(it.dst2, it.dtype, t * it.s2[0] + (a0 + lane) * it.s2[1] + (b0 + b) * it.s2[2], tile[lane * pitch + c] * sc);
            t += 4;
            while (t >= T) { t -= T; ++b; }
        }
    }
}

// Norms of the weight-normalised rows alone (msmc_wn_prepare_multi_tiles): block k takes row norm_rows[k] (an index into the
// grid of ALL rows, as block0 counts them) of item row_item[that row] -- two dependent loads where the binary search over the
// items' block0 was eight, a search that cost each of the 16 000 row workgroups of the autoencoder ~4 us before its first load.
__global__ __launch_bounds__(256) void wn_norm_kernel(const msmc_wn_item* __restrict__ items, const int* __restrict__ row_item,
                                                      const int* __restrict__ norm_rows) {
    __shared__ float red[4];
    const int grow = norm_rows[blockIdx.x];
    const msmc_wn_item it = items[row_item[grow]];
    const int a = grow - it.block0;
    const int n = it.Bc * it.T;
    const float* v = it.v + (size_t)a * n;
    float ss = 0.f;
    if ((n & 3) == 0) {
        const f32x4* v4 = (const f32x4*)v;
        for (int e = threadIdx.x; e < (n >> 2); e += 256) {
            const f32x4 q = v4[e];
            ss = fmaf(q[0], q[0], ss);
            ss = fmaf(q[1], q[1], ss);
            ss = fmaf(q[2], q[2], ss);
            ss = fmaf(q[3], q[3], ss);
        }
    } else {
        for (int e = threadIdx.x; e < n; e += 256) ss = fmaf(v[e], v[e], ss);
    }
    ss = block_sum_fast(ss, red);
    if (threadIdx.x == 0) it.inv_norm[a] = 1.f / sqrtf(ss);
}

// Both kernel layouts from ONE read of the parameter (round 6).  The row pass + transposing pass above moved the autoencoder's
// 36.7 M weights in 91 + 138 us per step -- three reads of v, a workgroup per row, 16 of a wave's 64 lanes loading in the
// transposing pass of every one-tap layer -- for 8 bytes per weight of real traffic, alone on the chip at the head of the step
// (profiles/r06_step_timeline_*.txt).  Here a workgroup owns a tile of TA rows (a) x TB columns (b) x all T taps, TA / TB chosen
// from T so that a row's share of the tile is ~128 contiguous floats or more: the tile is read in the parameter's own order
// (coalesced, whole wave) into LDS and written twice -- layout 2 with a fastest (TA consecutive elements per store), layout 1
// with b fastest (TB consecutive per store; both banks' layouts have s1[2] == 1 and s2[1] == 1, checked by the launcher's
// caller).  The scale g / ||v|| of weight-normalised rows comes from a norms-only row pass in front (wn_norm_kernel).
MSMC_DEV_INLINE int wn_tile_a(int T) { return T <= 4 ? 64 : 32; }
MSMC_DEV_INLINE int wn_tile_b(int T) { return T == 1 ? 128 : T == 2 ? 64 : 32; }
__global__ __launch_bounds__(256) void wn_layout_kernel(const msmc_wn_item* __restrict__ items, int nitems,
                                                        const int* __restrict__ tile_item) {
    MSMC_DYN_LDS(smem);
    float* tile = (float*)smem;
    __shared__ float scl[64];
    const msmc_wn_item it = items[tile_item ? tile_item[blockIdx.x] : wn_find_tile(items, nitems, blockIdx.x)];
    const int T = it.T, TA = wn_tile_a(T), TB = wn_tile_b(T);
    const int tb = blockIdx.x - it.tblock0;
    const int nbt = (it.Bc + TB - 1) / TB;
    const int ti = tb / nbt;
    const int a0 = ti * TA, b0 = (tb - ti * nbt) * TB;
    const int nb = it.Bc - b0 < TB ? it.Bc - b0 : TB, na = it.A - a0 < TA ? it.A - a0 : TA;
    const int ncol = nb * T, pitch = TB * T + 1;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)threadIdx.x < TA)
        scl[threadIdx.x] = (it.g && (int)threadIdx.x < na) ? it.g[a0 + threadIdx.x] * it.inv_norm[a0 + threadIdx.x] : 1.f;
    {
        // a wave reads rows w, w + 4, .. of the tile, 64 consecutive floats per load; SIXTEEN loads are issued before the first
        // of them is written to LDS (one in flight per work-item made the tile a chain of memory round trips)
        const float* __restrict__ vb = it.v + ((size_t)a0 * it.Bc + b0) * T;
        const size_t rowlen = (size_t)it.Bc * T;
        const int n_k = (ncol + 63) >> 6;
        const int nr = na > w ? (na - w + 3) >> 2 : 0;
        const int P = nr * n_k;
        for (int q0 = 0; q0 < P; q0 += 16) {
            float buf[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int q = q0 + i;
                buf[i] = 0.f;
                if (q < P) {
                    const int j = q / n_k, c = lane + 64 * (q - j * n_k);
                    if (c < ncol) buf[i] = vb[(size_t)(w + 4 * j) * rowlen + c];
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int q = q0 + i;
                if (q < P) {
                    const int j = q / n_k, c = lane + 64 * (q - j * n_k);
                    if (c < ncol) tile[(w + 4 * j) * pitch + c] = buf[i];
                }
            }
        }
    }
    __syncthreads();
    // Stores: FOUR consecutive elements per lane (8 bytes of bf16, 16 of fp32) wherever the fastest axis of a layout is a
    // multiple of four -- a wave instruction of 2-byte stores is 64 separate write requests to the memory pipeline whatever
    // their addresses (the first version of this kernel, one element per lane, moved 1.3 TB/s: no faster than the two
    // passes it replaces); single elements otherwise (the one- and two-channel first layers of the discriminators).
    if (it.dst2) {
        if ((it.A & 3) == 0) {
            // layout 2, a fastest: a lane takes rows 4l .. 4l+3 of one column (b, t); TA / 4 lanes per column
            const int lpc = TA >> 2, cpw = 64 / lpc, csub = lane / lpc, al = (lane - csub * lpc) * 4;
            const int step = 4 * cpw, qs = step / T, rs = step - qs * T;
            int c = w * cpw + csub;
            int b = c / T, t = c - b * T;
            const bool live = al < na;                 // (na is a multiple of four here)
            const float s0 = live ? scl[al] : 0.f, s1_ = live ? scl[al + 1] : 0.f, s2_ = live ? scl[al + 2] : 0.f,
                        s3 = live ? scl[al + 3] : 0.f;
            const float* col = tile + al * pitch;
            for (; c < ncol; c += step) {
                if (live) {
                    const long o = t * it.s2[0] + (a0 + al) + (long)(b0 + b) * it.s2[2];
                    const float x0 = col[c] * s0, x1 = col[pitch + c] * s1_, x2 = col[2 * pitch + c] * s2_, x3 = col[3 * pitch + c] * s3;
                    if (it.dtype == 0) {
                        const f32x4 q = {x0, x1, x2, x3};
                        *(f32x4*)((float*)it.dst2 + o) = q;
                    } else {
                        const u32x2 q = {pack_bf16x2(x0, x1), pack_bf16x2(x2, x3)};
                        *(u32x2*)((unsigned short*)it.dst2 + o) = q;
                    }
                }
                b += qs;
                t += rs;
                if (t >= T) { t -= T; ++b; }
            }
        } else {
            const int cpw = 64 / TA, csub = lane / TA, al = lane - csub * TA;
            const int step = 4 * cpw, qs = step / T, rs = step - qs * T;
            int c = w * cpw + csub;
            int b = c / T, t = c - b * T;
            const float sc = al < na ? scl[al] : 0.f;
            for (; c < ncol; c += step) {
                if (al < na)
                    wn_store(it.dst2, it.dtype, t * it.s2[0] + (a0 + al) * it.s2[1] + (b0 + b) * it.s2[2], tile[al * pitch + c] * sc);
                b += qs;
                t += rs;
                if (t >= T) { t -= T; ++b; }
            }
        }
    }
    if ((it.Bc & 3) == 0) {
        // layout 1, b fastest: a lane takes columns 4l .. 4l+3 of one run (a, t); TB / 4 lanes per run
        const int lpr = TB >> 2, rpw = 64 / lpr, sub = lane / lpr, bl = (lane - sub * lpr) * 4;
        const int nrun = na * T;
        for (int run = w * rpw + sub; run < nrun; run += 4 * rpw) {
            if (bl >= nb) continue;                    // (nb is a multiple of four here)
            const int a = run / T, t = run - a * T;
            const float sc = scl[a];
            const long o = t * it.s1[0] + (a0 + a) * it.s1[1] + (b0 + bl);
            const float* row = tile + a * pitch + t + bl * T;
            const float x0 = row[0] * sc, x1 = row[T] * sc, x2 = row[2 * T] * sc, x3 = row[3 * T] * sc;
            if (it.dtype == 0) {
                const f32x4 q = {x0, x1, x2, x3};
                *(f32x4*)((float*)it.dst1 + o) = q;
            } else {
                const u32x2 q = {pack_bf16x2(x0, x1), pack_bf16x2(x2, x3)};
                *(u32x2*)((unsigned short*)it.dst1 + o) = q;
            }
        }
    } else {
        const int lpr = TB < 64 ? TB : 64, rpw = 64 / lpr, sub = lane / lpr, bl = lane - sub * lpr;
        const int nrun = na * T;
        for (int run = w * rpw + sub; run < nrun; run += 4 * rpw) {
            const int a = run / T, t = run - a * T;
            const float sc = scl[a];
            const long o = t * it.s1[0] + (a0 + a) * it.s1[1] + (long)b0 * it.s1[2];
            const float* row = tile + a * pitch + t;
            for (int bb = bl; bb < nb; bb += lpr) wn_store(it.dst1, it.dtype, o + bb * it.s1[2], row[bb * T] * sc);
        }
    }
}

// Row pass of the backward: dW arrives in layout 1 (tap-major), v / gv live in the parameter's own order.  Rows of up
// to WN_ROW_MAX parameters go through LDS: dW is read TAP-OUTER (64 consecutive floats per wave load, privatised copies
// folded and zeroed on the way), then everything else runs in the parameter's order (coalesced v reads, coalesced gv
// stores).  Longer rows keep the direct form.
// (Round 4, measured and reverted: all of a row's dW / v loads issued up front from fully unrolled 24-step register arrays
//  -- 77 -> 120 us per call: the predicated steps of short rows and the register footprint cost more than the loads in flight
//  gain; profiles/README.md.  What the loops needed is below: no store between two loads of one array.)
#define WN_ROW_MAX 6144
// NT work-items per row (blockDim.x: 256, or 128 when every row of the bank fits ``row_cap`` <= 4096 floats -- the pass is a chain
// of four memory round trips per row, so its speed is the number of rows in flight per CU: 6 at 256 work-items and a 24 KB row
// buffer, up to 16 at 128 work-items and a buffer sized to the bank's longest row)
__global__ __launch_bounds__(256) void wn_backward_kernel(const msmc_wn_item* __restrict__ items, int nitems,
                                                         int accumulate, int row_cap) {
    MSMC_DYN_LDS(smem);
    float* red = (float*)smem;                         // [4]
    float* row = red + 4;                              // [row_cap]
    const int NT = (int)blockDim.x;
    const msmc_wn_item it = items[wn_find(items, nitems, blockIdx.x)];
    const int a = blockIdx.x - it.block0;
    const int n = it.Bc * it.T, T = it.T, Bc = it.Bc;
    const float* v = it.v + (size_t)a * n;
    float* dw = (float*)it.dw;
    const int R = it.copies > 1 ? it.copies : 1;
    const bool staged = n <= row_cap;
    float dot = 0.f;
    if (staged) {
        // LOADS ONLY in this loop: the accumulators are zeroed in a pass of their own at the end.  With `dw[o] = 0` between two
        // loads of the same array the compiler must keep every load behind the previous store (it cannot prove o' != o), so the
        // loop ran one memory round trip per element: SQ_WAIT_ANY 89 % of the wave cycles, 18 % of the HBM roofline (round 4).
        const float* __restrict__ dwr = dw;
        // (the loads of four steps go out together: one memory round trip per four elements of a work-item instead of one per
        //  element -- the loop bounds are run-time values, so the compiler does not do this on its own)
        if (R == 1) {
            const int nb = (Bc + NT - 1) / NT, steps = T * nb;           // step s: tap s / nb, channel threadIdx.x + NT * (s % nb)
            for (int s0 = 0; s0 < steps; s0 += 4) {
                float q[4];
                int dst[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = s0 + u, t = s / nb, b = threadIdx.x + NT * (s - t * nb);
                    const bool ok = s < steps && b < Bc;
                    dst[u] = ok ? b * T + t : -1;
                    q[u] = ok ? dwr[t * it.s1[0] + a * it.s1[1] + b * it.s1[2]] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (dst[u] >= 0) row[dst[u]] = q[u];
            }
        } else if (R <= 8) {
            // privatised copies (the thin layers: eight accumulators per element): all copies of an element requested together,
            // folded in copy order
            for (int t = 0; t < T; ++t) {
                const long o1 = t * it.s1[0] + a * it.s1[1];
                for (int b = threadIdx.x; b < Bc; b += NT) {
                    const long o = o1 + b * it.s1[2];
                    float q[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) q[r] = r < R ? dwr[o + r * it.dw_copy_stride] : 0.f;
                    float sum = q[0];
#pragma unroll
                    for (int r = 1; r < 8; ++r)
                        if (r < R) sum = sum + q[r];
                    row[b * T + t] = sum;
                }
            }
        } else {
            for (int t = 0; t < T; ++t) {
                const long o1 = t * it.s1[0] + a * it.s1[1];
                for (int b = threadIdx.x; b < Bc; b += NT) {
                    const long o = o1 + b * it.s1[2];
                    float sum = dwr[o];
                    for (int r = 1; r < R; ++r) sum = sum + dwr[o + r * it.dw_copy_stride];      // privatised copies: fold
                    row[b * T + t] = sum;
                }
            }
        }
        __syncthreads();
        if (it.g) {
            const float* __restrict__ vd = v;
            int e = threadIdx.x;
            for (; e + 3 * NT < n; e += 4 * NT) {
                const float v0 = vd[e], v1 = vd[e + NT], v2 = vd[e + 2 * NT], v3 = vd[e + 3 * NT];
                dot = fmaf(row[e], v0, dot);
                dot = fmaf(row[e + NT], v1, dot);
                dot = fmaf(row[e + 2 * NT], v2, dot);
                dot = fmaf(row[e + 3 * NT], v3, dot);
            }
            for (; e < n; e += NT) dot = fmaf(row[e], vd[e], dot);
        }
    } else {
        int b = 0, t = threadIdx.x;
        while (t >= T) { t -= T; ++b; }
        const int db_ = NT / T, dt_ = NT - db_ * T;    // e += NT as (b, t) += (db_, dt_) with carry
        for (int e = threadIdx.x; e < n; e += NT) {
            const long o = t * it.s1[0] + a * it.s1[1] + b * it.s1[2];
            float sum = dw[o];
            for (int r = 1; r < R; ++r) {
                sum = sum + dw[o + r * it.dw_copy_stride];
                dw[o + r * it.dw_copy_stride] = 0.f;
            }
            if (R > 1) dw[o] = sum;
            dot = fmaf(sum, v[e], dot);
            b += db_;
            t += dt_;
            if (t >= T) { t -= T; ++b; }
        }
    }
    float k1 = 1.f, k2 = 0.f;                          // plain weight: gv = dW
    if (it.g) {
        dot = block_sum_waves(dot, red);
        const float inv = it.inv_norm[a], gval = it.g[a];
        if (threadIdx.x == 0) it.gg[a] = accumulate ? it.gg[a] + dot * inv : dot * inv;
        k1 = gval * inv;
        k2 = dot * inv * inv;
    }
    float* gv = it.gv + (size_t)a * n;
    if (staged) {
        const float* __restrict__ vr = v;                  // (v is never written here: its loads may run ahead of the gv stores)
        float* __restrict__ gvr = gv;
        if (accumulate) {
            for (int e = threadIdx.x; e < n; e += NT) gvr[e] = gvr[e] + k1 * (row[e] - vr[e] * k2);
        } else {
            int e = threadIdx.x;
            for (; e + 3 * NT < n; e += 4 * NT) {             // (v was just read by the dot pass: these are cache hits, issued together)
                const float v0 = vr[e], v1 = vr[e + NT], v2 = vr[e + 2 * NT], v3 = vr[e + 3 * NT];
                gvr[e] = k1 * (row[e] - v0 * k2);
                gvr[e + NT] = k1 * (row[e + NT] - v1 * k2);
                gvr[e + 2 * NT] = k1 * (row[e + 2 * NT] - v2 * k2);
                gvr[e + 3 * NT] = k1 * (row[e + 3 * NT] - v3 * k2);
            }
            for (; e < n; e += NT) gvr[e] = k1 * (row[e] - vr[e] * k2);
        }
        for (int t = 0; t < T; ++t) {                      // each accumulator element has exactly this one reader: leave zeros
            const long o1 = t * it.s1[0] + a * it.s1[1];
            for (int b = threadIdx.x; b < Bc; b += NT) {
                const long o = o1 + b * it.s1[2];
                for (int r = 0; r < R; ++r) dw[o + r * it.dw_copy_stride] = 0.f;
            }
        }
    } else {
        int b = 0, t = threadIdx.x;
        while (t >= T) { t -= T; ++b; }
        const int db_ = NT / T, dt_ = NT - db_ * T;
        for (int e = threadIdx.x; e < n; e += NT) {
            const long o = t * it.s1[0] + a * it.s1[1] + b * it.s1[2];
            const float gnew = k1 * (dw[o] - v[e] * k2);
            gv[e] = accumulate ? gv[e] + gnew : gnew;
            dw[o] = 0.f;
            b += db_;
            t += dt_;
            if (t >= T) { t -= T; ++b; }
        }
    }
    if (it.db && threadIdx.x == 0)
        for (int c = a; c < it.nbias; c += it.A) {
            float sum;
            if (R <= 8) {                           // (all copies requested together, folded in copy order: one round trip, not R)
                const float* __restrict__ dbr = it.db;
                float q[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) q[r] = r < R ? dbr[c + r * it.db_copy_stride] : 0.f;
                sum = q[0];
#pragma unroll
                for (int r = 1; r < 8; ++r)
                    if (r < R) sum = sum + q[r];
                for (int r = 0; r < R; ++r) it.db[c + r * it.db_copy_stride] = 0.f;
            } else {
                sum = it.db[c];
                it.db[c] = 0.f;
                for (int r = 1; r < R; ++r) {
                    sum = sum + it.db[c + r * it.db_copy_stride];
                    it.db[c + r * it.db_copy_stride] = 0.f;
                }
            }
            it.gb[c] = accumulate ? it.gb[c] + sum : sum;
        }
}

template <typename T>
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                       T* __restrict__ gx, long n, float slope) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float gv = Elt<T>::ld(g + e);
        Elt<T>::st(gx + e, Elt<T>::ld(y + e) > 0.f ? gv : gv * slope);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, float* __restrict__ out, long rows, int C,
                                                    int rows_per_block) {
    __shared__ float red[256];
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    const int tid = threadIdx.x;
    if (C >= 256 || (256 % C) != 0) {               // one or more whole columns per work-item
        for (int c = tid; c < C; c += 256) {
            float s = 0.f;
            for (long r = r0; r < r1; ++r) s = s + Elt<T>::ld(g + r * C + c);
            atomicAdd(out + c, s);
        }
        return;
    }
    // C divides 256: the flat index e = tid + 256*k always lands on column tid % C
    const long n = (r1 - r0) * C;
    const T* base = g + r0 * C;
    float s = 0.f;
    for (long e = tid; e < n; e += 256) s = s + Elt<T>::ld(base + e);
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st >= C; st >>= 1) {
        if (tid < st) red[tid] = red[tid] + red[tid + st];
        __syncthreads();
    }
    if (tid < C) atomicAdd(out + tid, red[tid]);
}

// Column sums without atomics (the bias gradients of the transposed convolutions: 10^4 - 10^5 rows of 32-256 channels; the
// atomic form above serialises several hundred same-address atomics per column: 38 us for 12 MB): a block sums a contiguous
// range of rows into part[block][C]; colsum_final_kernel adds the blocks in order (bit-reproducible) into out (+=).
template <typename T>
__global__ __launch_bounds__(256) void colsum_part_kernel(const T* __restrict__ g, float* __restrict__ part, long rows, int C,
                                                         int rows_per_block) {
    __shared__ float red[256 * 8];
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    const int tid = threadIdx.x;
    float* dst = part + (size_t)blockIdx.x * C;
    constexpr int VE = 16 / (int)sizeof(T);          // elements per 16-byte vector
    if (C % VE == 0 && (256 * VE) % C == 0 && ((size_t)g & 15) == 0) {
        // vector v = tid + 256 k of the block's flat element range always covers the VE columns (tid * VE) % C ..: one
        // 16-byte load per step, VE running sums per work-item, then the work-items of a column group are added in order
        const long nvec = (r1 > r0 ? (r1 - r0) : 0) * C / VE;
        const T* base = g + r0 * C;
        float acc[VE];
#pragma unroll
        for (int q = 0; q < VE; ++q) acc[q] = 0.f;
        for (long v = tid; v < nvec; v += 256) {
            const u32x4 w = *(const u32x4*)(base + v * VE);
#pragma unroll
            for (int q = 0; q < VE; ++q) {
                float x;
                if (sizeof(T) == 4) x = __uint_as_float(w[q & 3]);
                else x = bf16_bits_to_f32((unsigned short)(w[(q >> 1) & 3] >> (16 * (q & 1))));
                acc[q] = acc[q] + x;
            }
        }
#pragma unroll
        for (int q = 0; q < VE; ++q) red[tid * VE + q] = acc[q];
        __syncthreads();
        const int groups = C / VE;                    // work-items t with t % groups == c / VE hold column c at slot c % VE
        for (int c = tid; c < C; c += 256) {
            float s = 0.f;
            for (int t = c / VE; t < 256; t += groups) s = s + red[t * VE + (c % VE)];
            dst[c] = s;
        }
        return;
    }
    if (C >= 256 || (256 % C) != 0) {               // one or more whole columns per work-item
        for (int c = tid; c < C; c += 256) {
            float s = 0.f;
            for (long r = r0; r < r1; ++r) s = s + Elt<T>::ld(g + r * C + c);
            dst[c] = s;
        }
        return;
    }
    const long n = (r1 - r0) * C;                   // C divides 256: the flat index tid + 256 k stays on column tid % C
    const T* base = g + r0 * C;
    float s = 0.f;
    for (long e = tid; e < n; e += 256) s = s + Elt<T>::ld(base + e);
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st >= C; st >>= 1) {
        if (tid < st) red[tid] = red[tid] + red[tid + st];
        __syncthreads();
    }
    if (tid < C) dst[tid] = red[tid];
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nblocks, int C,
                                                           float* __restrict__ out, int accumulate) {
    __shared__ float red[256];
    // 256 / CG slices of blocks per column group of CG = min(C, 256) columns, each summed in order, then added in order
    const int CG = C < 256 ? C : 256, slices = 256 / CG;
    for (int c0 = blockIdx.x * CG; c0 < C; c0 += gridDim.x * CG) {
        const int c = c0 + (int)(threadIdx.x % CG), sl = threadIdx.x / CG;
        const int per = (nblocks + slices - 1) / slices, b0 = sl * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
        // four interleaved accumulators, combined in a fixed order: the loads of a slice are independent of each other (a single
        // running sum made them one L2 round trip each: 32 us for 512 partial rows)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (sl < slices && c < C) {
            int b = b0;
            for (; b + 4 <= b1; b += 4) {
                s0 = s0 + part[(size_t)b * C + c];
                s1 = s1 + part[(size_t)(b + 1) * C + c];
                s2 = s2 + part[(size_t)(b + 2) * C + c];
                s3 = s3 + part[(size_t)(b + 3) * C + c];
            }
            for (; b < b1; ++b) s0 = s0 + part[(size_t)b * C + c];
        }
        const float s = (s0 + s1) + (s2 + s3);
        __syncthreads();
        red[threadIdx.x] = s;
        __syncthreads();
        if (sl == 0 && c < C) {
            float t = red[threadIdx.x];
            for (int q = 1; q < slices; ++q) t = t + red[q * CG + threadIdx.x];
            out[c] = accumulate ? out[c] + t : t;
        }
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

// multi-tensor, vectorised versions of the two element-wise backward helpers (one launch for the same layer of all
// resolution sub-discriminators; V consecutive channels per work-item)
struct FoldMultiArgs {
    int n, p;
    float slope;
    int res_first;                      // 1: (fold + res) * mask -- res is the gradient of a second reader of the ACTIVATED map
    int first[MSMC_GROUP_MAX + 1];
    const void* gp[MSMC_GROUP_MAX];
    const void* mask[MSMC_GROUP_MAX];
    const void* res[MSMC_GROUP_MAX];    // NULL, or [B][H][W][C] added after the mask (a second consumer's gradient)
    void* gx[MSMC_GROUP_MAX];
    int H[MSMC_GROUP_MAX], W[MSMC_GROUP_MAX], C[MSMC_GROUP_MAX];
    long items[MSMC_GROUP_MAX];         // B * H * W * (C / V)
};
// V consecutive channels per work-item (V * sizeof(T) = 16, 8, 4 or sizeof(T) bytes: the widest vector every member's channel
// count allows -- the first layers of the resolution stacks have 4 channels, which kept the whole call on 2-byte accesses);
// 32-bit index arithmetic (items < 2^31 is checked by the launcher: the 64-bit divisions cost more than the memory accesses)
template <typename T, int V>
MSMC_DEV void fold_ld(const T* src, float* out) {
    alignas(16) T v[V];
    if (V * sizeof(T) == 16) *(u32x4*)v = *(const u32x4*)src;
    else if (V * sizeof(T) == 8) *(u32x2*)v = *(const u32x2*)src;
    else if (V * sizeof(T) == 4) *(unsigned int*)v = *(const unsigned int*)src;
    else v[0] = src[0];
#pragma unroll
    for (int q = 0; q < V; ++q) out[q] = Elt<T>::ld(&v[q]);
}
template <typename T, int V>
__global__ __launch_bounds__(256) void reflect_fold_multi_kernel(FoldMultiArgs a) {
    const int k = cv_group_member(a.first, a.n);
    const unsigned nb = (unsigned)(a.first[k + 1] - a.first[k]);
    const int p = a.p;
    const int H = a.H[k], W = a.W[k], C = a.C[k], Hp = H + 2 * p, Wp = W + 2 * p;
    const unsigned CV = (unsigned)(C / V), items = (unsigned)a.items[k];
    const T* gp = (const T*)a.gp[k];
    const T* mask = (const T*)a.mask[k];
    const T* res = (const T*)a.res[k];
    T* gx = (T*)a.gx[k];
    for (unsigned e = (unsigned)(blockIdx.x - a.first[k]) * 256u + threadIdx.x; e < items; e += nb * 256u) {
        const unsigned pix = e / CV;
        const int c = (int)(e - pix * CV) * V;
        const unsigned row = pix / (unsigned)W;
        const int x = (int)(pix - row * (unsigned)W);
        const unsigned b = row / (unsigned)H;
        const int y = (int)(row - b * (unsigned)H);
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + p;
        if (y >= 1 && y <= p) ys[ny++] = p - y;
        if (y <= H - 2 && y >= H - 1 - p) ys[ny++] = 2 * (H - 1) - y + p;
        xs[nx++] = x + p;
        if (x >= 1 && x <= p) xs[nx++] = p - x;
        if (x <= W - 2 && x >= W - 1 - p) xs[nx++] = 2 * (W - 1) - x + p;
        float sacc[V];
#pragma unroll
        for (int q = 0; q < V; ++q) sacc[q] = 0.f;
        for (int i = 0; i < ny; ++i)
            for (int j = 0; j < nx; ++j) {
                float v[V];
                fold_ld<T, V>(gp + (((size_t)b * Hp + ys[i]) * Wp + xs[j]) * C + c, v);
#pragma unroll
                for (int q = 0; q < V; ++q) sacc[q] = sacc[q] + v[q];
            }
        const size_t o = (size_t)pix * C + c;
        float mv[V], rv[V];
        if (mask) fold_ld<T, V>(mask + o, mv);
        if (res) fold_ld<T, V>(res + o, rv);
        alignas(16) T ov[V];
#pragma unroll
        for (int q = 0; q < V; ++q) {
            float sv = sacc[q];
            if (res && a.res_first) sv = sv + rv[q];
            if (mask) sv = sv * (mv[q] > 0.f ? 1.f : a.slope);
            if (res && !a.res_first) sv = sv + rv[q];
            Elt<T>::st(&ov[q], sv);
        }
        if (V * sizeof(T) == 16) *(u32x4*)(gx + o) = *(const u32x4*)ov;
        else if (V * sizeof(T) == 8) *(u32x2*)(gx + o) = *(const u32x2*)ov;
        else if (V * sizeof(T) == 4) *(unsigned int*)(gx + o) = *(const unsigned int*)ov;
        else gx[o] = ov[0];
    }
}

struct LreluMultiArgs {
    int n;
    float slope;
    int first[MSMC_GROUP_MAX + 1];
    const void* g[MSMC_GROUP_MAX];
    const void* y[MSMC_GROUP_MAX];
    void* gx[MSMC_GROUP_MAX];
    long items[MSMC_GROUP_MAX];         // elements / V
};
template <typename T, int V>
__global__ __launch_bounds__(256) void lrelu_bwd_multi_kernel(LreluMultiArgs a) {
    const int k = cv_group_member(a.first, a.n);
    const int nb = a.first[k + 1] - a.first[k];
    const T* g = (const T*)a.g[k];
    const T* y = (const T*)a.y[k];
    T* gx = (T*)a.gx[k];
    for (long e = (long)(blockIdx.x - a.first[k]) * 256 + threadIdx.x; e < a.items[k]; e += (long)nb * 256) {
        alignas(16) T gv[V], yv[V], ov[V];
        if (V * sizeof(T) == 16) {
            *(u32x4*)gv = *(const u32x4*)(g + e * V);
            *(u32x4*)yv = *(const u32x4*)(y + e * V);
        } else {
            gv[0] = g[e];
            yv[0] = y[e];
        }
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const float gf = Elt<T>::ld(&gv[q]);
            Elt<T>::st(&ov[q], Elt<T>::ld(&yv[q]) > 0.f ? gf : gf * a.slope);
        }
        if (V * sizeof(T) == 16) *(u32x4*)(gx + e * V) = *(const u32x4*)ov;
        else gx[e] = ov[0];
    }
}

__global__ void zero_kernel(float* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

template <typename T>
static int fold_multi_launch(FoldMultiArgs& a, int v, int blocks, msmc_stream stream) {
    constexpr int VMAX = Elt<T>::VEC;           // 16-byte vectors: 4 fp32 / 8 bf16
    if (v == VMAX) MSMC_LAUNCH((reflect_fold_multi_kernel<T, VMAX>), dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream, a);
    else if (v == VMAX / 2) MSMC_LAUNCH((reflect_fold_multi_kernel<T, VMAX / 2>), dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream, a);
    else if (v == 2 && VMAX == 8) MSMC_LAUNCH((reflect_fold_multi_kernel<T, 2>), dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream, a);
    else MSMC_LAUNCH((reflect_fold_multi_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream, a);
    return msmc_check_launch();
}

template <typename T>
static int lrelu_multi_launch(LreluMultiArgs& a, bool vec, int blocks, msmc_stream stream) {
    if (vec) MSMC_LAUNCH((lrelu_bwd_multi_kernel<T, Elt<T>::VEC>), dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream, a);
    else MSMC_LAUNCH((lrelu_bwd_multi_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream, a);
    return msmc_check_launch();
}

extern "C" {

int msmc_wn_prepare_multi(const msmc_wn_item* items, int nitems, int total_blocks, msmc_stream stream) {
    if (!items || nitems <= 0 || total_blocks <= 0) return MSMC_E_SHAPE;
    MSMC_LAUNCH(wn_prepare_kernel, dim3(total_blocks), dim3(256), 0, (msmc_stream_t)stream, items, nitems, 0);
    return msmc_check_launch();
}

int msmc_wn_prepare_multi_tiled(const msmc_wn_item* items, int nitems, int total_blocks, int total_tile_blocks,
                                msmc_stream stream) {
    if (!items || nitems <= 0 || total_blocks <= 0 || total_tile_blocks < 0) return MSMC_E_SHAPE;
    MSMC_LAUNCH(wn_prepare_kernel, dim3(total_blocks), dim3(256), 0, (msmc_stream_t)stream, items, nitems, 1);
    int rc = msmc_check_launch();
    if (rc || total_tile_blocks == 0) return rc;
    MSMC_LAUNCH(wn_transpose_kernel, dim3(total_tile_blocks), dim3(256), 0, (msmc_stream_t)stream, items, nitems);
    return msmc_check_launch();
}

// tile-blocks of one item under wn_layout_kernel's tile rule (the host lays tblock0 out with it)
int msmc_wn_tile_blocks(int A, int Bc, int T) {
    if (A <= 0 || Bc <= 0 || T <= 0) return 0;
    const int ta = T <= 4 ? 64 : 32, tb = T == 1 ? 128 : T == 2 ? 64 : 32;
    return ((A + ta - 1) / ta) * ((Bc + tb - 1) / tb);
}
int msmc_wn_prepare_multi_tiles(const msmc_wn_item* items, int nitems, int total_blocks, int total_tile_blocks, int max_taps,
                                const int* row_item, const int* norm_rows, int n_norm_rows, const int* tile_item,
                                msmc_stream stream) {
    if (!items || nitems <= 0 || total_blocks <= 0 || total_tile_blocks <= 0 || max_taps <= 0 || max_taps > MSMC_CONV_MAX_TAPS ||
        n_norm_rows < 0 || (n_norm_rows > 0 && (!row_item || !norm_rows)))
        return MSMC_E_SHAPE;
    if (n_norm_rows > 0) {
        MSMC_LAUNCH(wn_norm_kernel, dim3(n_norm_rows), dim3(256), 0, (msmc_stream_t)stream, items, row_item, norm_rows);
        int rc = msmc_check_launch();
        if (rc) return rc;
    }
    size_t lds = 0;
    for (int T = 1; T <= max_taps; ++T) {
        const int ta = T <= 4 ? 64 : 32, tb = T == 1 ? 128 : T == 2 ? 64 : 32;
        const size_t l = (size_t)ta * (tb * T + 1) * sizeof(float);
        if (l > lds) lds = l;
    }
    int rc = msmc_allow_lds((const void*)wn_layout_kernel, (int)lds);
    if (rc) return rc;
    MSMC_LAUNCH(wn_layout_kernel, dim3(total_tile_blocks), dim3(256), lds, (msmc_stream_t)stream, items, nitems, tile_item);
    return msmc_check_launch();
}

int msmc_wn_backward_multi_rows(const msmc_wn_item* items, int nitems, int total_blocks, int accumulate, int max_row,
                                msmc_stream stream) {
    if (!items || nitems <= 0 || total_blocks <= 0) return MSMC_E_SHAPE;
    // max_row: the longest normalised row (Bc * T parameters) among the items, 0 = unknown.  Rows up to 4096 floats: 128
    // work-items per row and a row buffer of that size (more rows in flight per CU); otherwise the 256 / 24 KB form, whose
    // rows beyond WN_ROW_MAX take the unstaged path
    const bool small = max_row > 0 && max_row <= 4096;
    const int cap = small ? ((max_row + 63) & ~63) : WN_ROW_MAX;
    const size_t lds = 16 + (size_t)cap * sizeof(float);
    int rc = msmc_allow_lds((const void*)wn_backward_kernel, (int)lds);
    if (rc) return rc;
    MSMC_LAUNCH(wn_backward_kernel, dim3(total_blocks), dim3(small ? 128 : 256), lds, (msmc_stream_t)stream, items, nitems,
                accumulate, cap);
    return msmc_check_launch();
}

int msmc_wn_backward_multi_acc(const msmc_wn_item* items, int nitems, int total_blocks, int accumulate, msmc_stream stream) {
    return msmc_wn_backward_multi_rows(items, nitems, total_blocks, accumulate, 0, stream);
}

int msmc_wn_backward_multi(const msmc_wn_item* items, int nitems, int total_blocks, msmc_stream stream) {
    return msmc_wn_backward_multi_acc(items, nitems, total_blocks, 0, stream);
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

int msmc_lrelu_bwd(const void* g, const void* y, void* gx, long n, float slope, int dtype, msmc_stream stream) {
    if (!g || !y || !gx || n <= 0) return MSMC_E_SHAPE;
    long blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    if (dtype == 0)
        MSMC_LAUNCH(lrelu_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream,
                    (const float*)g, (const float*)y, (float*)gx, n, slope);
    else if (dtype == 1)
        MSMC_LAUNCH(lrelu_bwd_kernel<unsigned short>, dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream,
                    (const unsigned short*)g, (const unsigned short*)y, (unsigned short*)gx, n, slope);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}

int msmc_reflect_fold_multi(const void* const* gp, const void* const* mask_src, void* const* gx, const int* B, const int* H,
                            const int* W, const int* C, int n, int p, float slope, int dtype, msmc_stream stream) {
    return msmc_reflect_fold_multi_res(gp, mask_src, nullptr, gx, B, H, W, C, n, p, slope, dtype, stream);
}

static int fold_multi_impl(const void* const* gp, const void* const* mask_src, const void* const* res, void* const* gx,
                           const int* B, const int* H, const int* W, const int* C, int n, int p, float slope, int dtype,
                           int res_first, msmc_stream stream);

int msmc_reflect_fold_multi_res(const void* const* gp, const void* const* mask_src, const void* const* res, void* const* gx,
                                const int* B, const int* H, const int* W, const int* C, int n, int p, float slope, int dtype,
                                msmc_stream stream) {
    return fold_multi_impl(gp, mask_src, res, gx, B, H, W, C, n, p, slope, dtype, 0, stream);
}

int msmc_reflect_fold_multi_tap(const void* const* gp, const void* const* mask_src, const void* const* tap, void* const* gx,
                                const int* B, const int* H, const int* W, const int* C, int n, int p, float slope, int dtype,
                                msmc_stream stream) {
    return fold_multi_impl(gp, mask_src, tap, gx, B, H, W, C, n, p, slope, dtype, 1, stream);
}

static int colsum_blocks(long rows) {
    long nb = (rows + 127) / 128;                    // >= 128 rows per block, at most one block per CU (the second stage is ONE
    if (nb > 256) nb = 256;                          // workgroup per 256 columns: it reads nb rows of partial sums)
    return (int)(nb < 1 ? 1 : nb);
}
size_t msmc_colsum_workspace(long rows, int C) { return rows > 0 && C > 0 ? (size_t)colsum_blocks(rows) * C * sizeof(float) : 0; }
int msmc_colsum_ws(const void* g, float* out, long rows, int C, int dtype, int accumulate, void* workspace,
                   size_t workspace_bytes, msmc_stream stream) {
    if (!g || !out || rows <= 0 || C <= 0) return MSMC_E_SHAPE;
    if (!workspace || workspace_bytes < msmc_colsum_workspace(rows, C)) return MSMC_E_WORKSPACE;
    const int nb = colsum_blocks(rows);
    const int rpb = (int)((rows + nb - 1) / nb);
    float* part = (float*)workspace;
    if (dtype == 0) MSMC_LAUNCH(colsum_part_kernel<float>, dim3(nb), dim3(256), 0, (msmc_stream_t)stream, (const float*)g, part, rows, C, rpb);
    else if (dtype == 1) MSMC_LAUNCH(colsum_part_kernel<unsigned short>, dim3(nb), dim3(256), 0, (msmc_stream_t)stream, (const unsigned short*)g, part, rows, C, rpb);
    else return MSMC_E_SHAPE;
    int rc = msmc_check_launch();
    if (rc) return rc;
    const int used = (int)((rows + rpb - 1) / rpb);  // (blocks that own rows)
    MSMC_LAUNCH(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, (msmc_stream_t)stream, (const float*)part, used, C, out,
                accumulate);
    return msmc_check_launch();
}

}  // extern "C"

static int fold_multi_impl(const void* const* gp, const void* const* mask_src, const void* const* res, void* const* gx,
                           const int* B, const int* H, const int* W, const int* C, int n, int p, float slope, int dtype,
                           int res_first, msmc_stream stream) {
    if (!gp || !gx || !B || !H || !W || !C || n <= 0 || n > MSMC_GROUP_MAX || p < 0 || dtype < 0 || dtype > 1)
        return MSMC_E_SHAPE;
    int v = dtype == 0 ? 4 : 8;                 // widest vector (in elements) every member's channel count is a multiple of
    for (int k = 0; k < n; ++k) {
        if (!gp[k] || !gx[k] || B[k] <= 0 || C[k] <= 0 || H[k] <= p || W[k] <= p) return MSMC_E_SHAPE;
        while (v > 1 && (C[k] % v) != 0) v >>= 1;
    }
    for (int k = 0; k < n; ++k)                 // (16 / 8 / 4-byte accesses need that alignment of every operand)
        while (v > 1 && (((size_t)gp[k] | (size_t)gx[k] | (size_t)(mask_src && mask_src[k] ? mask_src[k] : nullptr) |
                          (size_t)(res && res[k] ? res[k] : nullptr)) & (size_t)(v * (dtype == 0 ? 4 : 2) - 1)))
            v >>= 1;
    FoldMultiArgs a;
    a.n = n;
    a.p = p;
    a.slope = slope;
    a.res_first = res_first;
    int blocks = 0;
    for (int k = 0; k < n; ++k) {
        a.gp[k] = gp[k];
        a.mask[k] = mask_src ? mask_src[k] : nullptr;
        a.res[k] = res ? res[k] : nullptr;
        a.gx[k] = gx[k];
        a.H[k] = H[k];
        a.W[k] = W[k];
        a.C[k] = C[k];
        a.items[k] = (long)B[k] * H[k] * W[k] * (C[k] / v);
        if (a.items[k] >= (1L << 31)) return MSMC_E_SHAPE;
        long nb = (a.items[k] + 255) / 256;
        if (nb > 16L * MSMC_NUM_CU) nb = 16L * MSMC_NUM_CU;       // (one or two items per work-item: an item's loads cannot run ahead of the
        a.first[k] = blocks;                                      //  previous item's store, so parallelism has to come from the grid)
        blocks += (int)(nb < 1 ? 1 : nb);
    }
    a.first[n] = blocks;
    return dtype == 0 ? fold_multi_launch<float>(a, v, blocks, stream) : fold_multi_launch<unsigned short>(a, v, blocks, stream);
}

extern "C" {

int msmc_lrelu_bwd_multi(const void* const* g, const void* const* y, void* const* gx, const long* nelem, int n, float slope,
                         int dtype, msmc_stream stream) {
    if (!g || !y || !gx || !nelem || n <= 0 || n > MSMC_GROUP_MAX || dtype < 0 || dtype > 1) return MSMC_E_SHAPE;
    const int VEC = dtype == 0 ? 4 : 8;
    bool vec = true;
    for (int k = 0; k < n; ++k) {
        if (!g[k] || !y[k] || !gx[k] || nelem[k] <= 0) return MSMC_E_SHAPE;
        vec = vec && (nelem[k] % VEC) == 0;
    }
    LreluMultiArgs a;
    a.n = n;
    a.slope = slope;
    int blocks = 0;
    for (int k = 0; k < n; ++k) {
        a.g[k] = g[k];
        a.y[k] = y[k];
        a.gx[k] = gx[k];
        a.items[k] = nelem[k] / (vec ? VEC : 1);
        long nb = (a.items[k] + 255) / 256;
        if (nb > 4L * MSMC_NUM_CU) nb = 4L * MSMC_NUM_CU;
        a.first[k] = blocks;
        blocks += (int)(nb < 1 ? 1 : nb);
    }
    a.first[n] = blocks;
    return dtype == 0 ? lrelu_multi_launch<float>(a, vec, blocks, stream) : lrelu_multi_launch<unsigned short>(a, vec, blocks, stream);
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
