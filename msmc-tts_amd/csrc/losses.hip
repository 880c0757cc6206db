// losses.hip -- multi-tensor GAN loss terms (feature-matching L1, LSGAN MSE-to-constant) for gfx950.
//
// Replaces the python loops over score tensors / feature maps in VQGANTrainer.train_step
// (reference msmctts/trainers/msmctts_trainer.py:165-171 and :187-193): one launch reads every
// tensor pair once (HBM-bound, 16-byte loads), reduces per block and adds mean_i into one scalar;
// the backward is one launch writing every gradient.  Tensor tables travel as kernel arguments, so
// the launches are hipGraph-capturable.
#include <msmc_rt.hpp>
#include <msmc_hip.h>

template <typename T> struct LEl;
template <> struct LEl<float> {
    static MSMC_DEV_INLINE float ld(const float* p) { return *p; }
    static MSMC_DEV_INLINE void st(float* p, float v) { *p = v; }
};
template <> struct LEl<unsigned short> {
    static MSMC_DEV_INLINE float ld(const unsigned short* p) { return bf16_bits_to_f32(*p); }
    static MSMC_DEV_INLINE void st(unsigned short* p, float v) { *p = f32_to_bf16_bits(v); }
};

MSMC_DEV float loss_block_sum(float v, float* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = red[tid] + red[tid + s];
        __syncthreads();
    }
    return red[0];
}

// MODE 0: L1(a, b)   MODE 1: (a - target)^2
// part != NULL: the block's sum goes to part[tensor * gridDim.x + block] (loss_multi_final_kernel adds them in a fixed order:
// bit-reproducible, and no 14 000 same-address atomics per launch); part == NULL: atomicAdd of the block's share into out
template <typename T, int MODE>
__global__ __launch_bounds__(256) void loss_multi_fwd_kernel(msmc_tensor_table t, float target, float* __restrict__ out,
                                                            float* __restrict__ part) {
    __shared__ float red[256];
    const int i = blockIdx.y;
    const long n = t.n[i];
    const T* a = (const T*)t.a[i];
    const T* b = (const T*)t.b[i];
    float s = 0.f;
    // 16-byte vectors where the operands allow (the feature maps: 424 MB per step went through 2-byte loads at 2 TB/s)
    constexpr int VE = 16 / (int)sizeof(T);
    const bool vec = ((((size_t)a) | ((size_t)b)) & 15) == 0;
    const long nv = vec ? n / VE : 0;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (long)gridDim.x * 256) {
        const u32x4 av4 = *(const u32x4*)(a + v * VE);
        u32x4 bv4 = {0u, 0u, 0u, 0u};
        if (MODE == 0) bv4 = *(const u32x4*)(b + v * VE);
#pragma unroll
        for (int q = 0; q < VE; ++q) {
            float av, bv;
            if (sizeof(T) == 4) {
                av = __uint_as_float(av4[q & 3]);
                bv = __uint_as_float(bv4[q & 3]);
            } else {
                av = bf16_bits_to_f32((unsigned short)(av4[(q >> 1) & 3] >> (16 * (q & 1))));
                bv = bf16_bits_to_f32((unsigned short)(bv4[(q >> 1) & 3] >> (16 * (q & 1))));
            }
            if (MODE == 0) {
                s = s + fabsf(av - bv);
            } else {
                const float dlt = av - target;
                s = fmaf(dlt, dlt, s);
            }
        }
    }
    for (long e = nv * VE + (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float av = LEl<T>::ld(a + e);
        if (MODE == 0) {
            s = s + fabsf(av - LEl<T>::ld(b + e));
        } else {
            const float dlt = av - target;
            s = fmaf(dlt, dlt, s);
        }
    }
    s = loss_block_sum(s, red);
    if (part) {
        if (threadIdx.x == 0) part[(size_t)i * gridDim.x + blockIdx.x] = s;
        return;
    }
    if (threadIdx.x == 0 && (long)blockIdx.x * 256 < n) atomicAdd(out, s / (float)n);
}

// out[0] = sum_i (sum_x part[i][x]) / n_i, tensors in table order, blocks in index order (one workgroup)
__global__ __launch_bounds__(256) void loss_multi_final_kernel(msmc_tensor_table t, const float* __restrict__ part, int bx,
                                                               float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < t.count; i += 256) {
        float ti = 0.f;
        for (int x = 0; x < bx; ++x) ti = ti + part[(size_t)i * bx + x];
        s = s + ti / (float)t.n[i];
    }
    s = loss_block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void loss_multi_bwd_kernel(msmc_tensor_table t, float target,
                                                            const float* __restrict__ gout) {
    const int i = blockIdx.y;
    const long n = t.n[i];
    const T* a = (const T*)t.a[i];
    const T* b = (const T*)t.b[i];
    T* ga = (T*)t.ga[i];
    const float scale = gout[0] / (float)n;
    constexpr int VE = 16 / (int)sizeof(T);
    const bool vec = ((((size_t)a) | ((size_t)b) | ((size_t)ga)) & 15) == 0;
    const long nv = vec ? n / VE : 0;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (long)gridDim.x * 256) {
        const u32x4 av4 = *(const u32x4*)(a + v * VE);
        u32x4 bv4 = {0u, 0u, 0u, 0u};
        if (MODE == 0) bv4 = *(const u32x4*)(b + v * VE);
        float gq[VE];
#pragma unroll
        for (int q = 0; q < VE; ++q) {
            float av, bv;
            if (sizeof(T) == 4) {
                av = __uint_as_float(av4[q & 3]);
                bv = __uint_as_float(bv4[q & 3]);
            } else {
                av = bf16_bits_to_f32((unsigned short)(av4[(q >> 1) & 3] >> (16 * (q & 1))));
                bv = bf16_bits_to_f32((unsigned short)(bv4[(q >> 1) & 3] >> (16 * (q & 1))));
            }
            if (MODE == 0) {
                const float dlt = av - bv;
                gq[q] = dlt > 0.f ? scale : (dlt < 0.f ? -scale : 0.f);
            } else {
                gq[q] = 2.f * (av - target) * scale;
            }
        }
        u32x4 o;
        if (sizeof(T) == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = __float_as_uint(gq[q]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                o[q] = (unsigned int)f32_to_bf16_bits(gq[(2 * q) % VE]) | ((unsigned int)f32_to_bf16_bits(gq[(2 * q + 1) % VE]) << 16);
        }
        *(u32x4*)(ga + v * VE) = o;
    }
    for (long e = nv * VE + (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float av = LEl<T>::ld(a + e);
        float g;
        if (MODE == 0) {
            const float dlt = av - LEl<T>::ld(b + e);
            g = dlt > 0.f ? scale : (dlt < 0.f ? -scale : 0.f);
        } else {
            g = 2.f * (av - target) * scale;
        }
        LEl<T>::st(ga + e, g);
    }
}

__global__ void loss_zero_kernel(float* p) { p[0] = 0.f; }

// ---- masked means over [B][T][C] tensors with per-utterance lengths ------------------------------------------------------
// The length-masked scalar terms of the step -- QuantizerLoss (reference msmctts_trainer.py:52-62: diff.masked_fill(pad, 0)
// .sum() / length.sum() / C), the frame loss (:129-133: the same over mse(mel, mel_outputs)) and the 'mse' embedding loss of
// the prior predictor (vqgantts/msmc_vqgan.py:228-247) -- were chains of ~10 stock kernels each (arange, compare,
// masked_fill, sum, sum of lengths, two divisions ...).  Here: partial sums per (block, utterance) over the VALID rows only,
// one single-workgroup launch that adds them in a fixed order (bit-reproducible) and divides, and one backward launch.
// MODE 0: sum a;  MODE 1: sum (a - b)^2.
MSMC_DEV long loss_len(const void* len, int is64, int b) {
    return is64 ? (long)((const long long*)len)[b] : (long)((const int*)len)[b];
}
#define MM_BX 32
template <typename TA, typename TB, int MODE>
__global__ __launch_bounds__(256) void masked_mean_partial_kernel(const TA* __restrict__ a, const TB* __restrict__ b,
                                                                  const void* __restrict__ len, int is64, int T, int C,
                                                                  float* __restrict__ part) {
    __shared__ float red[256];
    const int bi = blockIdx.y;
    long L = loss_len(len, is64, bi);
    if (L > T) L = T;
    if (L < 0) L = 0;
    const long n = L * C, base = (long)bi * T * C;
    float s = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)MM_BX * 256) {
        const float av = LEl<TA>::ld(a + base + e);
        if (MODE == 0) {
            s = s + av;
        } else {
            const float dlt = av - LEl<TB>::ld(b + base + e);
            s = fmaf(dlt, dlt, s);
        }
    }
    s = loss_block_sum(s, red);
    if (threadIdx.x == 0) part[bi * MM_BX + blockIdx.x] = s;
}
// out[0] = (sum of the partials in index order) / sum_b len_b / C;  out[1] = 1 / (sum_b len_b * C) for the backward pass
__global__ __launch_bounds__(256) void masked_mean_final_kernel(const float* __restrict__ part, int nparts,
                                                                const void* __restrict__ len, int is64, int B, int T, int C,
                                                                float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int k = threadIdx.x; k < nparts; k += 256) s = s + part[k];
    s = loss_block_sum(s, red);
    if (threadIdx.x == 0) {
        float tot = 0.f;                                   // (the reference sums the lengths as they are: no clamp to T)
        for (int b = 0; b < B; ++b) tot = tot + (float)loss_len(len, is64, b);
        out[0] = s / tot / (float)C;
        out[1] = 1.f / tot / (float)C;
    }
}
// ga = gout * out[1] * (MODE 0: 1, MODE 1: 2 (a - b)) on valid rows, 0 on padding; gb = -ga (when asked for)
template <typename TA, typename TB, int MODE>
__global__ __launch_bounds__(256) void masked_mean_bwd_kernel(const TA* __restrict__ a, const TB* __restrict__ b,
                                                              const void* __restrict__ len, int is64, int T, int C,
                                                              const float* __restrict__ out, const float* __restrict__ gout,
                                                              TA* __restrict__ ga, TB* __restrict__ gb) {
    const int bi = blockIdx.y;
    long L = loss_len(len, is64, bi);
    if (L > T) L = T;
    if (L < 0) L = 0;
    const long n = (long)T * C, nv = L * C, base = (long)bi * T * C;
    const float k = gout[0] * out[1];
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        float g = 0.f;
        if (e < nv) g = MODE == 0 ? k : 2.f * (LEl<TA>::ld(a + base + e) - LEl<TB>::ld(b + base + e)) * k;
        if (ga) LEl<TA>::st(ga + base + e, g);
        if (gb) LEl<TB>::st(gb + base + e, -g);
    }
}

template <int MODE>
#define LOSS_WS_BX 64
static int loss_launch(const msmc_tensor_table* t, float target, float* out, const float* gout, bool bwd,
                       msmc_stream stream, float* part = nullptr) {
    if (!t || t->count <= 0 || t->count > MSMC_MAX_TENSORS) return MSMC_E_SHAPE;
    long nmax = 0;
    for (int i = 0; i < t->count; ++i) {
        if (t->n[i] <= 0) return MSMC_E_SHAPE;
        if (t->n[i] > nmax) nmax = t->n[i];
    }
    long bx = (nmax + 4095) / 4096;
    if (bx > 256) bx = 256;
    dim3 grid((unsigned)bx, (unsigned)t->count);
    if (!bwd && part) {
        const dim3 g2(LOSS_WS_BX, (unsigned)t->count);
        if (t->dtype == 0) MSMC_LAUNCH((loss_multi_fwd_kernel<float, MODE>), g2, dim3(256), 0, (msmc_stream_t)stream, *t, target, out, part);
        else if (t->dtype == 1) MSMC_LAUNCH((loss_multi_fwd_kernel<unsigned short, MODE>), g2, dim3(256), 0, (msmc_stream_t)stream, *t, target, out, part);
        else return MSMC_E_SHAPE;
        int rc = msmc_check_launch();
        if (rc) return rc;
        MSMC_LAUNCH(loss_multi_final_kernel, dim3(1), dim3(256), 0, (msmc_stream_t)stream, *t, (const float*)part, LOSS_WS_BX, out);
    } else if (!bwd) {
        MSMC_LAUNCH(loss_zero_kernel, dim3(1), dim3(64), 0, (msmc_stream_t)stream, out);
        if (t->dtype == 0) MSMC_LAUNCH((loss_multi_fwd_kernel<float, MODE>), grid, dim3(256), 0, (msmc_stream_t)stream, *t, target, out, (float*)nullptr);
        else if (t->dtype == 1) MSMC_LAUNCH((loss_multi_fwd_kernel<unsigned short, MODE>), grid, dim3(256), 0, (msmc_stream_t)stream, *t, target, out, (float*)nullptr);
        else return MSMC_E_SHAPE;
    } else {
        if (t->dtype == 0) MSMC_LAUNCH((loss_multi_bwd_kernel<float, MODE>), grid, dim3(256), 0, (msmc_stream_t)stream, *t, target, gout);
        else if (t->dtype == 1) MSMC_LAUNCH((loss_multi_bwd_kernel<unsigned short, MODE>), grid, dim3(256), 0, (msmc_stream_t)stream, *t, target, gout);
        else return MSMC_E_SHAPE;
    }
    return msmc_check_launch();
}

// ---- weighted sums of loss scalars ------------------------------------------------------------------------------------------
// vq_loss = sum_i lambda_i term_i, g_loss = vq_loss + lambda_frame frame + lambda_stft stft (+ adv + lambda_fm fm), d_loss =
// d_real + d_fake (reference msmctts_trainer.py:52-62,129-133,160-195) were a multiply and an add of 0-dim tensors per term: ~25
// one-thread launches per step.  One launch per sum here (and one for all the terms' gradients).
struct ScalarSumArgs {
    int n;
    const float* x[MSMC_MAX_TENSORS];
    float w[MSMC_MAX_TENSORS];
};
__global__ void scalar_wsum_fwd_kernel(ScalarSumArgs a, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < a.n; ++i) s = fmaf(a.w[i], a.x[i][0], s);      // (in term order)
        out[0] = s;
    }
}
__global__ void scalar_wsum_bwd_kernel(ScalarSumArgs a, const float* __restrict__ gout, float* __restrict__ gvec) {
    const int i = threadIdx.x;
    if (i < a.n) gvec[i] = gout[0] * a.w[i];
}

template <typename TA, typename TB>
static int masked_mean_go(const void* a, const void* b, const void* len, int is64, int B, int T, int C, int mode, float* part,
                          float* out, const float* gout, void* ga, void* gb, bool bwd, msmc_stream stream) {
    const dim3 grid(MM_BX, (unsigned)B);
    if (!bwd) {
        if (mode == 0) MSMC_LAUNCH((masked_mean_partial_kernel<TA, TB, 0>), grid, dim3(256), 0, (msmc_stream_t)stream, (const TA*)a, (const TB*)b, len, is64, T, C, part);
        else MSMC_LAUNCH((masked_mean_partial_kernel<TA, TB, 1>), grid, dim3(256), 0, (msmc_stream_t)stream, (const TA*)a, (const TB*)b, len, is64, T, C, part);
        int rc = msmc_check_launch();
        if (rc) return rc;
        MSMC_LAUNCH(masked_mean_final_kernel, dim3(1), dim3(256), 0, (msmc_stream_t)stream, (const float*)part, B * MM_BX, len, is64, B, T, C, out);
    } else {
        long bx = ((long)T * C + 2047) / 2048;
        if (bx > 64) bx = 64;
        if (bx < 1) bx = 1;
        const dim3 g2((unsigned)bx, (unsigned)B);
        if (mode == 0) MSMC_LAUNCH((masked_mean_bwd_kernel<TA, TB, 0>), g2, dim3(256), 0, (msmc_stream_t)stream, (const TA*)a, (const TB*)b, len, is64, T, C, (const float*)out, gout, (TA*)ga, (TB*)gb);
        else MSMC_LAUNCH((masked_mean_bwd_kernel<TA, TB, 1>), g2, dim3(256), 0, (msmc_stream_t)stream, (const TA*)a, (const TB*)b, len, is64, T, C, (const float*)out, gout, (TA*)ga, (TB*)gb);
    }
    return msmc_check_launch();
}
static int masked_mean_dispatch(const void* a, const void* b, const void* len, int is64, int B, int T, int C, int a_dtype,
                                int b_dtype, int mode, float* part, float* out, const float* gout, void* ga, void* gb, bool bwd,
                                msmc_stream stream) {
    if (!a || !len || !out || B <= 0 || T <= 0 || C <= 0 || (mode != 0 && mode != 1) || (mode == 1 && !b)) return MSMC_E_SHAPE;
    if (!bwd && !part) return MSMC_E_WORKSPACE;
    if (bwd && !gout) return MSMC_E_SHAPE;
    if (mode == 0) b_dtype = a_dtype;
    if (a_dtype == 0 && b_dtype == 0) return masked_mean_go<float, float>(a, b, len, is64, B, T, C, mode, part, out, gout, ga, gb, bwd, stream);
    if (a_dtype == 0 && b_dtype == 1) return masked_mean_go<float, unsigned short>(a, b, len, is64, B, T, C, mode, part, out, gout, ga, gb, bwd, stream);
    if (a_dtype == 1 && b_dtype == 0) return masked_mean_go<unsigned short, float>(a, b, len, is64, B, T, C, mode, part, out, gout, ga, gb, bwd, stream);
    if (a_dtype == 1 && b_dtype == 1) return masked_mean_go<unsigned short, unsigned short>(a, b, len, is64, B, T, C, mode, part, out, gout, ga, gb, bwd, stream);
    return MSMC_E_SHAPE;
}


// ---------------------------------------------------------------------------------------------------------------
// Triple (hinge) loss of predictor training against a frozen codebook -- Quantize.compute_triple_loss, reference
// msmctts/networks/vqgantts/modules.py:86-116 (+ the head loop :152-168), the 'masked version':
//   dist_k = (|p|^2 - 2 p.e_k) + |e_k|^2,  pos = sum_c (p_c - e_trg,c)^2,  t_k = pos - dist_k,
//   loss = reduce_k [t_k != 0] max(t_k + margin, 0) / d           (reduce = sum or mean over the K codewords)
// and its gradient, which does not depend on p beyond the active set:  d loss / d p = (2 / d) sum_{k active} (e_k - e_trg)
// (x 1 / K for the mean).  One work-item per (frame, head): p and the running sum of active codewords in registers, the head's
// codebook rows [K][d] and squared norms in LDS (every work-item reads the same row: broadcast), 2 x K x d multiply-adds per
// pair.  The stock chain materialises the [frames x K] distance matrix of every head several times (26 MB per head at B = 64).
// lossh [N][H] (the caller takes the mean over heads), gp [N][H d] = d lossh[n][h] / d p[n][h d ..].
template <int D4>
__global__ __launch_bounds__(256) void triple_loss_kernel(const float* __restrict__ p, const long long* __restrict__ trg,
                                                         const float* __restrict__ embed_t, const float* __restrict__ enorm,
                                                         float* __restrict__ lossh, float* __restrict__ gp, int N, int H, int K,
                                                         float margin, int mean) {
    MSMC_DYN_LDS(smem);
    constexpr int d = 4 * D4;
    float* erow = (float*)smem;                                 // [K][d]
    float* en = erow + (size_t)K * d;                           // [K]
    const int h = blockIdx.y, tid = threadIdx.x;
    const float* eh = embed_t + (size_t)h * K * d;
    for (int e = tid; e < K * d / 4; e += 256) *(f32x4*)(erow + 4 * e) = *(const f32x4*)(eh + 4 * e);
    for (int k = tid; k < K; k += 256) en[k] = enorm[(size_t)h * K + k];
    __syncthreads();
    const int n = blockIdx.x * 256 + tid;
    if (n >= N) return;
    const int D = H * d;
    f32x4 x[D4], s[D4];
    float pp = 0.f;
#pragma unroll
    for (int c = 0; c < D4; ++c) {
        x[c] = *(const f32x4*)(p + (size_t)n * D + h * d + 4 * c);
        s[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) pp = pp + x[c][j] * x[c][j];
    }
    long long t = trg[(size_t)n * H + h];
    t = t < 0 ? 0 : (t >= K ? K - 1 : t);
    const float* et = erow + (size_t)t * d;
    float pos = 0.f;
#pragma unroll
    for (int c = 0; c < D4; ++c) {
        const f32x4 e = *(const f32x4*)(et + 4 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float df = x[c][j] - e[j];
            pos = pos + df * df;
        }
    }
    const float invd = 1.f / (float)d;
    float loss = 0.f, nact = 0.f;
    for (int k = 0; k < K; ++k) {
        const float* ek = erow + (size_t)k * d;
        float dot = 0.f;
        f32x4 e[D4];
#pragma unroll
        for (int c = 0; c < D4; ++c) {
            e[c] = *(const f32x4*)(ek + 4 * c);
#pragma unroll
            for (int j = 0; j < 4; ++j) dot = dot + x[c][j] * e[c][j];
        }
        const float dist = (pp - 2.f * dot) + en[k];
        const float tr = pos - dist;
        const float v = tr + margin;
        const float a = (tr != 0.f && v > 0.f) ? 1.f : 0.f;
        loss = loss + a * (v * invd);
        nact = nact + a;
#pragma unroll
        for (int c = 0; c < D4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[c][j] = s[c][j] + a * e[c][j];
    }
    const float scale = mean ? 1.f / (float)K : 1.f;
    lossh[(size_t)n * H + h] = loss * scale;
    const float gs = 2.f * invd * scale;
#pragma unroll
    for (int c = 0; c < D4; ++c) {
        const f32x4 e = *(const f32x4*)(et + 4 * c);
        f32x4 g4;
#pragma unroll
        for (int j = 0; j < 4; ++j) g4[j] = gs * (s[c][j] - nact * e[j]);
        *(f32x4*)(gp + (size_t)n * D + h * d + 4 * c) = g4;
    }
}

extern "C" {
int msmc_triple_loss(const float* p, const int64_t* trg, const float* embed_t, const float* enorm, float* lossh, float* gp, int N,
                     int D, int H, int K, float margin, int mean, msmc_stream stream) {
    if (!p || !trg || !embed_t || !enorm || !lossh || !gp || N < 0 || H <= 0 || K <= 0 || D <= 0 || D % H) return MSMC_E_SHAPE;
    const int d = D / H;
    if (d % 4 || (((size_t)p | (size_t)embed_t | (size_t)gp) & 15)) return MSMC_E_SHAPE;
    const size_t lds = ((size_t)K * d + K) * sizeof(float);
    if (lds > 160 * 1024) return MSMC_E_SHAPE;
    if (N == 0) return 0;
    const dim3 grid((unsigned)((N + 255) / 256), (unsigned)H);
#define TRIPLE_GO(D4_)                                                                                              \
    do {                                                                                                           \
        int rc = msmc_allow_lds((const void*)triple_loss_kernel<D4_>, (int)lds);                                   \
        if (rc) return rc;                                                                                         \
        MSMC_LAUNCH((triple_loss_kernel<D4_>), grid, dim3(256), lds, (msmc_stream_t)stream, p, (const long long*)trg, embed_t, \
                    enorm, lossh, gp, N, H, K, margin, mean);                                                      \
    } while (0)
    switch (d) {
        case 16: TRIPLE_GO(4); break;
        case 32: TRIPLE_GO(8); break;
        case 64: TRIPLE_GO(16); break;
        case 128: TRIPLE_GO(32); break;
        default: return MSMC_E_SHAPE;
    }
#undef TRIPLE_GO
    return msmc_check_launch();
}
int msmc_masked_mean_parts(int B) { return B * MM_BX; }
int msmc_masked_mean_fwd(const void* a, const void* b, const void* lengths, int len_is_64, int B, int T, int C, int a_dtype,
                         int b_dtype, int mode, float* partial, float* out, msmc_stream stream) {
    return masked_mean_dispatch(a, b, lengths, len_is_64, B, T, C, a_dtype, b_dtype, mode, partial, out, nullptr, nullptr, nullptr,
                                false, stream);
}
int msmc_masked_mean_bwd(const void* a, const void* b, const void* lengths, int len_is_64, int B, int T, int C, int a_dtype,
                         int b_dtype, int mode, const float* out, const float* gout, void* ga, void* gb, msmc_stream stream) {
    return masked_mean_dispatch(a, b, lengths, len_is_64, B, T, C, a_dtype, b_dtype, mode, nullptr, (float*)out, gout, ga, gb, true,
                                stream);
}
int msmc_l1_multi_fwd(const msmc_tensor_table* t, float* out, msmc_stream stream) {
    return loss_launch<0>(t, 0.f, out, nullptr, false, stream);
}
int msmc_scalar_wsum_fwd(const float* const* terms, const float* weights, int n, float* out, msmc_stream stream) {
    if (!terms || !weights || !out || n <= 0 || n > MSMC_MAX_TENSORS) return MSMC_E_SHAPE;
    ScalarSumArgs a;
    a.n = n;
    for (int i = 0; i < n; ++i) {
        if (!terms[i]) return MSMC_E_SHAPE;
        a.x[i] = terms[i];
        a.w[i] = weights[i];
    }
    MSMC_LAUNCH(scalar_wsum_fwd_kernel, dim3(1), dim3(64), 0, (msmc_stream_t)stream, a, out);
    return msmc_check_launch();
}
int msmc_scalar_wsum_bwd(const float* gout, const float* weights, int n, float* gvec, msmc_stream stream) {
    if (!gout || !weights || !gvec || n <= 0 || n > MSMC_MAX_TENSORS) return MSMC_E_SHAPE;
    ScalarSumArgs a;
    a.n = n;
    for (int i = 0; i < n; ++i) {
        a.x[i] = nullptr;
        a.w[i] = weights[i];
    }
    MSMC_LAUNCH(scalar_wsum_bwd_kernel, dim3(1), dim3(64), 0, (msmc_stream_t)stream, a, gout, gvec);
    return msmc_check_launch();
}
int msmc_loss_multi_parts(void) { return MSMC_MAX_TENSORS * LOSS_WS_BX; }
int msmc_l1_multi_fwd_ws(const msmc_tensor_table* t, float* partial, float* out, msmc_stream stream) {
    if (!partial) return MSMC_E_WORKSPACE;
    return loss_launch<0>(t, 0.f, out, nullptr, false, stream, partial);
}
int msmc_mse_const_multi_fwd_ws(const msmc_tensor_table* t, float target, float* partial, float* out, msmc_stream stream) {
    if (!partial) return MSMC_E_WORKSPACE;
    return loss_launch<1>(t, target, out, nullptr, false, stream, partial);
}
int msmc_l1_multi_bwd(const msmc_tensor_table* t, const float* gout, msmc_stream stream) {
    return loss_launch<0>(t, 0.f, nullptr, gout, true, stream);
}
int msmc_mse_const_multi_fwd(const msmc_tensor_table* t, float target, float* out, msmc_stream stream) {
    return loss_launch<1>(t, target, out, nullptr, false, stream);
}
int msmc_mse_const_multi_bwd(const msmc_tensor_table* t, float target, const float* gout, msmc_stream stream) {
    return loss_launch<1>(t, target, nullptr, gout, true, stream);
}
}
