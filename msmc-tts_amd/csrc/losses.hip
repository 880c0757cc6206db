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
template <typename T, int MODE>
__global__ __launch_bounds__(256) void loss_multi_fwd_kernel(msmc_tensor_table t, float target, float* __restrict__ out) {
    __shared__ float red[256];
    const int i = blockIdx.y;
    const long n = t.n[i];
    const T* a = (const T*)t.a[i];
    const T* b = (const T*)t.b[i];
    float s = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float av = LEl<T>::ld(a + e);
        if (MODE == 0) {
            s = s + fabsf(av - LEl<T>::ld(b + e));
        } else {
            const float dlt = av - target;
            s = fmaf(dlt, dlt, s);
        }
    }
    s = loss_block_sum(s, red);
    if (threadIdx.x == 0 && (long)blockIdx.x * 256 < n) atomicAdd(out, s / (float)n);
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
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
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

template <int MODE>
static int loss_launch(const msmc_tensor_table* t, float target, float* out, const float* gout, bool bwd,
                       msmc_stream stream) {
    if (!t || t->count <= 0 || t->count > MSMC_MAX_TENSORS) return MSMC_E_SHAPE;
    long nmax = 0;
    for (int i = 0; i < t->count; ++i) {
        if (t->n[i] <= 0) return MSMC_E_SHAPE;
        if (t->n[i] > nmax) nmax = t->n[i];
    }
    long bx = (nmax + 2047) / 2048;
    if (bx > 256) bx = 256;
    dim3 grid((unsigned)bx, (unsigned)t->count);
    if (!bwd) {
        MSMC_LAUNCH(loss_zero_kernel, dim3(1), dim3(64), 0, (msmc_stream_t)stream, out);
        if (t->dtype == 0) MSMC_LAUNCH((loss_multi_fwd_kernel<float, MODE>), grid, dim3(256), 0, (msmc_stream_t)stream, *t, target, out);
        else if (t->dtype == 1) MSMC_LAUNCH((loss_multi_fwd_kernel<unsigned short, MODE>), grid, dim3(256), 0, (msmc_stream_t)stream, *t, target, out);
        else return MSMC_E_SHAPE;
    } else {
        if (t->dtype == 0) MSMC_LAUNCH((loss_multi_bwd_kernel<float, MODE>), grid, dim3(256), 0, (msmc_stream_t)stream, *t, target, gout);
        else if (t->dtype == 1) MSMC_LAUNCH((loss_multi_bwd_kernel<unsigned short, MODE>), grid, dim3(256), 0, (msmc_stream_t)stream, *t, target, gout);
        else return MSMC_E_SHAPE;
    }
    return msmc_check_launch();
}

extern "C" {
int msmc_l1_multi_fwd(const msmc_tensor_table* t, float* out, msmc_stream stream) {
    return loss_launch<0>(t, 0.f, out, nullptr, false, stream);
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
