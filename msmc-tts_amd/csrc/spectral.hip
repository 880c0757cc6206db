// spectral.hip -- STFT framing and magnitude / log / mel glue kernels for gfx950.
//
// The DFT itself and the mel projections run as one-tap channels-last "convolutions" (plain GEMMs) on the
// fp32 matrix cores through msmc_conv_gather; these kernels are the memory-bound pieces around them:
// reflect-padded framing and its overlap-add adjoint, magnitude, the MRD two-channel image with the
// transposition to the channels-last [B][F][T][2] layout the discriminator stack consumes, and log-clamp.
// Replaces torch.stft + glue of TorchSTFT.transform (reference msmctts/utils/audio.py:398-419),
// MelScale.forward (:348-376) and MelLoss.mel_spectrogram (reference criterions/stft_loss.py:76-108).
#include <msmc_rt.hpp>
#include <msmc_hip.h>

MSMC_DEV int sp_reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

MSMC_DEV void stft_frames_fwd_body(const float* __restrict__ x, float* __restrict__ fr, int L,
                                                             int T, int n_fft, int NP, int hop, int pad, long total, int bid, int nb) {
    for (long e = (long)bid * 256 + threadIdx.x; e < total; e += (long)nb * 256) {
        const int j = (int)(e % NP);
        const long r = e / NP;
        const int t = (int)(r % T);
        const long b = r / T;
        float v = 0.f;
        if (j < n_fft) v = x[b * L + sp_reflect(t * hop + j - pad, L)];
        fr[e] = v;
    }
}
__global__ __launch_bounds__(256) void stft_frames_fwd_kernel(const float* __restrict__ x, float* __restrict__ fr, int L,
                                                             int T, int n_fft, int NP, int hop, int pad, long total) {
    stft_frames_fwd_body(x, fr, L, T, n_fft, NP, hop, pad, total, (int)blockIdx.x, (int)gridDim.x);
}

// gx[b][l]: frames read padded position p = t*hop + j; sample l is read at p = l + pad and, through the
// reflection, at p = pad - l (l >= 1, left border) and p = 2(L-1) - l + pad (l <= L-2, right border).
MSMC_DEV void stft_frames_bwd_body(const float* __restrict__ gfr, float* __restrict__ gx,
                                                             int L, int T, int n_fft, int NP, int hop, int pad,
                                                             long total, int bid, int nb) {
    const int Lp = (T - 1) * hop + n_fft;           // padded positions actually covered by frames
    for (long e = (long)bid * 256 + threadIdx.x; e < total; e += (long)nb * 256) {
        const int l = (int)(e % L);
        const long b = e / L;
        int ps[3], np = 0;
        ps[np++] = l + pad;
        if (l >= 1 && l <= pad) ps[np++] = pad - l;
        if (l <= L - 2 && 2 * (L - 1) - l + pad < Lp) ps[np++] = 2 * (L - 1) - l + pad;
        float s = 0.f;
        for (int q = 0; q < np; ++q) {
            const int p = ps[q];
            if (p < 0 || p >= Lp) continue;
            int t_hi = p / hop;
            if (t_hi > T - 1) t_hi = T - 1;
            int t_lo = (p - n_fft + hop) / hop;      // ceil((p - n_fft + 1) / hop)
            if (p - n_fft + 1 <= 0) t_lo = 0;
            for (int t = t_lo; t <= t_hi; ++t) {
                const int j = p - t * hop;
                if (j >= 0 && j < n_fft) s = s + gfr[(b * T + t) * NP + j];
            }
        }
        gx[e] = s;
    }
}
__global__ __launch_bounds__(256) void stft_frames_bwd_kernel(const float* __restrict__ gfr, float* __restrict__ gx,
                                                             int L, int T, int n_fft, int NP, int hop, int pad,
                                                             long total) {
    stft_frames_bwd_body(gfr, gx, L, T, n_fft, NP, hop, pad, total, (int)blockIdx.x, (int)gridDim.x);
}

MSMC_DEV void spec_mag_fwd_body(const float* __restrict__ spec, float* __restrict__ mag, int F,
                                                          int CP, int FP, float lo, int clamp_mode, long total, int bid, int nb) {
    for (long e = (long)bid * 256 + threadIdx.x; e < total; e += (long)nb * 256) {
        const int f = (int)(e % FP);
        const long r = e / FP;
        float v = 0.f;
        if (f < F) {
            const float re = spec[r * CP + f], im = spec[r * CP + F + f];
            float s = re * re + im * im;
            s = clamp_mode ? (s < lo ? lo : s) : (s + lo);
            v = sqrtf(s);
        }
        mag[e] = v;
    }
}
__global__ __launch_bounds__(256) void spec_mag_fwd_kernel(const float* __restrict__ spec, float* __restrict__ mag, int F,
                                                          int CP, int FP, float lo, int clamp_mode, long total) {
    spec_mag_fwd_body(spec, mag, F, CP, FP, lo, clamp_mode, total, (int)blockIdx.x, (int)gridDim.x);
}

MSMC_DEV void spec_mag_bwd_body(const float* __restrict__ spec, const float* __restrict__ mag,
                                                          const float* __restrict__ gmag, float* __restrict__ gspec,
                                                          int F, int CP, int FP, float lo, int clamp_mode, long total, int bid, int nb) {
    for (long e = (long)bid * 256 + threadIdx.x; e < total; e += (long)nb * 256) {
        const int c = (int)(e % CP);
        const long r = e / CP;
        float g = 0.f;
        if (c < 2 * F) {
            const int f = c < F ? c : c - F;
            const float re = spec[r * CP + f], im = spec[r * CP + F + f];
            const float s = re * re + im * im;
            if (!clamp_mode || s > lo) g = gmag[r * FP + f] * (c < F ? re : im) / mag[r * FP + f];
        }
        gspec[e] = g;
    }
}
__global__ __launch_bounds__(256) void spec_mag_bwd_kernel(const float* __restrict__ spec, const float* __restrict__ mag,
                                                          const float* __restrict__ gmag, float* __restrict__ gspec,
                                                          int F, int CP, int FP, float lo, int clamp_mode, long total) {
    spec_mag_bwd_body(spec, mag, gmag, gspec, F, CP, FP, lo, clamp_mode, total, (int)blockIdx.x, (int)gridDim.x);
}

// img[b][f][t][0..1] <- mel[(b*T + t)][f]   (transposed write; reads are the strided side, tiny tensors)
// (image element type O: float, or bf16 bits when the discriminator stack computes in bf16 -- no separate cast launch)
MSMC_DEV void sp_store2(float* p, float a, float b) { f32x2 o = {a, b}; *(f32x2*)p = o; }
MSMC_DEV void sp_store2(unsigned short* p, float a, float b) {
    *(unsigned int*)p = (unsigned int)f32_to_bf16_bits(a) | ((unsigned int)f32_to_bf16_bits(b) << 16);
}
MSMC_DEV void sp_load2(const float* p, float& a, float& b) { const f32x2 v = *(const f32x2*)p; a = v[0]; b = v[1]; }
MSMC_DEV void sp_load2(const unsigned short* p, float& a, float& b) {
    const unsigned int v = *(const unsigned int*)p;
    a = __uint_as_float(v << 16);
    b = __uint_as_float(v & 0xffff0000u);
}
template <typename O>
MSMC_DEV void mrd_image_fwd_body(const float* __restrict__ mel, O* __restrict__ img, int T,
                                                           int F, int FP, long total, int bid, int nb) {
    for (long e = (long)bid * 256 + threadIdx.x; e < total; e += (long)nb * 256) {
        const int t = (int)(e % T);
        long r = e / T;
        const int f = (int)(r % F);
        const long b = r / F;
        const float m = mel[(b * T + t) * FP + f];
        float lg = (20.f * log10f(m) - 20.f + 100.f) / 100.f;
        lg = lg < 0.f ? 0.f : (lg > 1.f ? 1.f : lg);
        sp_store2(img + e * 2, m, lg);
    }
}
template <typename O>
__global__ __launch_bounds__(256) void mrd_image_fwd_kernel(const float* __restrict__ mel, O* __restrict__ img, int T,
                                                           int F, int FP, long total) {
    mrd_image_fwd_body<O>(mel, img, T, F, FP, total, (int)blockIdx.x, (int)gridDim.x);
}

template <typename O>
MSMC_DEV void mrd_image_bwd_body(const float* __restrict__ mel, const O* __restrict__ gimg,
                                                           float* __restrict__ gmel, int T, int F, int FP, long total, int bid, int nb) {
    for (long e = (long)bid * 256 + threadIdx.x; e < total; e += (long)nb * 256) {
        const int f = (int)(e % FP);
        const long r = e / FP;              // b*T + t
        const int t = (int)(r % T);
        const long b = r / T;
        float g = 0.f;
        if (f < F) {
            const float m = mel[e];
            float g0, g1;
            sp_load2(gimg + (((b * F + f) * (long)T) + t) * 2, g0, g1);
            const float lg = (20.f * log10f(m) - 20.f + 100.f) / 100.f;
            g = g0;
            if (lg > 0.f && lg < 1.f) g = g + g1 * (0.2f / (2.302585092994046f * m));
        }
        gmel[e] = g;
    }
}
template <typename O>
__global__ __launch_bounds__(256) void mrd_image_bwd_kernel(const float* __restrict__ mel, const O* __restrict__ gimg,
                                                           float* __restrict__ gmel, int T, int F, int FP, long total) {
    mrd_image_bwd_body<O>(mel, gimg, gmel, T, F, FP, total, (int)blockIdx.x, (int)gridDim.x);
}

MSMC_DEV void log_clamp_fwd_body(const float* __restrict__ x, float* __restrict__ y, long n,
                                                           float lo, int bid, int nb) {
    for (long e = (long)bid * 256 + threadIdx.x; e < n; e += (long)nb * 256) {
        const float v = x[e];
        y[e] = logf(v < lo ? lo : v);
    }
}
__global__ __launch_bounds__(256) void log_clamp_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                           float lo) {
    log_clamp_fwd_body(x, y, n, lo, (int)blockIdx.x, (int)gridDim.x);
}
MSMC_DEV void log_clamp_bwd_body(const float* __restrict__ x, const float* __restrict__ g,
                                                           float* __restrict__ gx, long n, float lo, int bid, int nb) {
    for (long e = (long)bid * 256 + threadIdx.x; e < n; e += (long)nb * 256) {
        const float v = x[e];
        gx[e] = v > lo ? g[e] / v : 0.f;
    }
}
__global__ __launch_bounds__(256) void log_clamp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                           float* __restrict__ gx, long n, float lo) {
    log_clamp_bwd_body(x, g, gx, n, lo, (int)blockIdx.x, (int)gridDim.x);
}

// Several of the element-wise stages above in ONE launch (msmc_spectral_multi): the five resolution discriminators' front-ends
// are five identical chains frames -> DFT -> magnitude -> filter bank -> image over different hop lengths, each stage a launch of
// 5-10 us for a microsecond of work, and a chain of 25 such launches sat on the step's critical chain forward and again
// backward (profiles/r06_step_timeline_start_of_round.txt).  A block looks its operation up in the argument and runs the stage's
// own body over its share of the operation's elements.
struct SpMultiArgs {
    int n;
    int first[MSMC_SPECTRAL_MULTI_MAX + 1];
    msmc_spectral_op op[MSMC_SPECTRAL_MULTI_MAX];
};
__global__ __launch_bounds__(256) void spectral_multi_kernel(SpMultiArgs a) {
    int k = 0;
    while (k + 1 < a.n && (int)blockIdx.x >= a.first[k + 1]) ++k;
    const msmc_spectral_op& o = a.op[k];
    const int bid = (int)blockIdx.x - a.first[k], nb = a.first[k + 1] - a.first[k];
    switch (o.kind) {
        case 0: stft_frames_fwd_body((const float*)o.a, (float*)o.out, o.L, o.T, o.n_fft, o.NP, o.hop, o.pad, (long)o.B * o.T * o.NP, bid, nb); break;
        case 1: stft_frames_bwd_body((const float*)o.a, (float*)o.out, o.L, o.T, o.n_fft, o.NP, o.hop, o.pad, (long)o.B * o.L, bid, nb); break;
        case 2: spec_mag_fwd_body((const float*)o.a, (float*)o.out, o.F, o.CP, o.FP, o.lo, o.clamp_mode, o.R * o.FP, bid, nb); break;
        case 3: spec_mag_bwd_body((const float*)o.a, (const float*)o.b, (const float*)o.c, (float*)o.out, o.F, o.CP, o.FP, o.lo, o.clamp_mode, o.R * o.CP, bid, nb); break;
        case 4:
            if (o.dtype == 0) mrd_image_fwd_body<float>((const float*)o.a, (float*)o.out, o.T, o.F, o.FP, (long)o.B * o.F * o.T, bid, nb);
            else mrd_image_fwd_body<unsigned short>((const float*)o.a, (unsigned short*)o.out, o.T, o.F, o.FP, (long)o.B * o.F * o.T, bid, nb);
            break;
        case 6: log_clamp_fwd_body((const float*)o.a, (float*)o.out, o.R, o.lo, bid, nb); break;
        case 7: log_clamp_bwd_body((const float*)o.a, (const float*)o.b, (float*)o.out, o.R, o.lo, bid, nb); break;
        default:
            if (o.dtype == 0) mrd_image_bwd_body<float>((const float*)o.a, (const float*)o.b, (float*)o.out, o.T, o.F, o.FP, (long)o.B * o.T * o.FP, bid, nb);
            else mrd_image_bwd_body<unsigned short>((const float*)o.a, (const unsigned short*)o.b, (float*)o.out, o.T, o.F, o.FP, (long)o.B * o.T * o.FP, bid, nb);
            break;
    }
}

static inline dim3 sp_grid(long total) {
    long b = (total + 1023) / 1024;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

// ---- waveform fan-out of the discriminator (reference hifigan/discriminator.py:102-116,135-145,180-190) ------------------
// One generated waveform feeds five resolution sub-discriminators (fp32 spectral front-ends) and five period sub-discriminators
// (a cast to the stack's dtype, a reflection pad to a multiple of the period for two of them, a fold).  As stock operators that
// is a cast + two pads forward and, backward, two pad gradients, a cast gradient and ~20 accumulations of [B][L] gradients by
// the autograd engine.  Forward here: every period's padded copy in ONE launch; backward: the sum of all consumers' gradients
// (fp32 front-end gradients + padded period gradients folded at the reflected tail) in ONE launch.
#define WF_MAX 8
struct WaveFanArgs {
    int n16, n32;                       // period copies / fp32 gradient tensors
    int Lp[WF_MAX];                     // padded length of copy k
    void* p16[WF_MAX];                  // copies [B][Lp[k]] (forward: written; backward: their gradients, NULL = none)
    const float* p32[WF_MAX];           // backward only: fp32 gradients [B][L] (NULL = none)
};
template <typename T>
__global__ __launch_bounds__(256) void wave_fan_fwd_kernel(const float* __restrict__ y, WaveFanArgs a, int L, int Lmax,
                                                          long total) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int t = (int)(e % Lmax);
        const long b = e / Lmax;
        const float v = y[b * L + sp_reflect(t, L)];
#pragma unroll
        for (int k = 0; k < WF_MAX; ++k)
            if (k < a.n16 && t < a.Lp[k]) {
                if (sizeof(T) == 4) ((float*)a.p16[k])[b * a.Lp[k] + t] = v;
                else ((unsigned short*)a.p16[k])[b * a.Lp[k] + t] = f32_to_bf16_bits(v);
            }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void wave_fan_bwd_kernel(WaveFanArgs a, float* __restrict__ gy, int L, long total) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int l = (int)(e % L);
        const long b = e / L;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < WF_MAX; ++k)
            if (k < a.n32 && a.p32[k]) s = s + a.p32[k][e];
        const int tr = 2 * (L - 1) - l;                  // padded position that reflects onto l (tail only: tr >= L)
#pragma unroll
        for (int k = 0; k < WF_MAX; ++k)
            if (k < a.n16 && a.p16[k]) {
                const long base = b * a.Lp[k];
                if (sizeof(T) == 4) {
                    const float* g = (const float*)a.p16[k];
                    s = s + g[base + l];
                    if (tr >= L && tr < a.Lp[k]) s = s + g[base + tr];
                } else {
                    const unsigned short* g = (const unsigned short*)a.p16[k];
                    s = s + bf16_bits_to_f32(g[base + l]);
                    if (tr >= L && tr < a.Lp[k]) s = s + bf16_bits_to_f32(g[base + tr]);
                }
            }
        gy[e] = s;
    }
}

// The vocoder windows of a captured step (reference msmctts_trainer.py:211-219 random_select, as device arithmetic): frame indices
// start_b + i and the waveform window wav[b][start_b * hop + j] of every utterance in ONE launch -- the operator chain (two
// aranges, a multiply, two broadcast adds, a gather) was six launches at the head of the step's critical chain.
__global__ __launch_bounds__(256) void window_gather_kernel(const long* __restrict__ starts, const float* __restrict__ wav,
                                                           long* __restrict__ frames, float* __restrict__ target, int nframes,
                                                           int hop, long L, long total) {
    const long per = (long)nframes * hop;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / per, j = e - b * per;
        const long s0 = starts[b];
        target[e] = wav[b * L + s0 * hop + j];
        if (j < nframes) frames[b * nframes + j] = s0 + j;
    }
}

extern "C" {

int msmc_stft_frames_fwd(const float* x, float* frames, int B, int L, int T, int n_fft, int NP, int hop, int pad,
                         msmc_stream stream) {
    if (!x || !frames || B <= 0 || L <= pad || T <= 0 || NP < n_fft || hop <= 0 || pad < 0) return MSMC_E_SHAPE;
    const long total = (long)B * T * NP;
    MSMC_LAUNCH(stft_frames_fwd_kernel, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, x, frames, L, T, n_fft, NP,
                hop, pad, total);
    return msmc_check_launch();
}
int msmc_stft_frames_bwd(const float* gframes, float* gx, int B, int L, int T, int n_fft, int NP, int hop, int pad,
                         msmc_stream stream) {
    if (!gframes || !gx || B <= 0 || L <= pad || T <= 0 || NP < n_fft || hop <= 0 || pad < 0) return MSMC_E_SHAPE;
    const long total = (long)B * L;
    MSMC_LAUNCH(stft_frames_bwd_kernel, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, gframes, gx, L, T, n_fft, NP,
                hop, pad, total);
    return msmc_check_launch();
}
int msmc_spec_mag_fwd(const float* spec, float* mag, long R, int F, int CP, int FP, float lo, int clamp_mode,
                      msmc_stream stream) {
    if (!spec || !mag || R <= 0 || F <= 0 || CP < 2 * F || FP < F) return MSMC_E_SHAPE;
    const long total = R * FP;
    MSMC_LAUNCH(spec_mag_fwd_kernel, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, spec, mag, F, CP, FP, lo,
                clamp_mode, total);
    return msmc_check_launch();
}
int msmc_spec_mag_bwd(const float* spec, const float* mag, const float* gmag, float* gspec, long R, int F, int CP,
                      int FP, float lo, int clamp_mode, msmc_stream stream) {
    if (!spec || !mag || !gmag || !gspec || R <= 0 || F <= 0 || CP < 2 * F || FP < F) return MSMC_E_SHAPE;
    const long total = R * CP;
    MSMC_LAUNCH(spec_mag_bwd_kernel, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, spec, mag, gmag, gspec, F, CP,
                FP, lo, clamp_mode, total);
    return msmc_check_launch();
}
int msmc_mrd_image_fwd_dt(const float* mel, void* img, int B, int T, int F, int FP, int dtype, msmc_stream stream) {
    if (!mel || !img || B <= 0 || T <= 0 || F <= 0 || FP < F || dtype < 0 || dtype > 1) return MSMC_E_SHAPE;
    const long total = (long)B * F * T;
    if (dtype == 0)
        MSMC_LAUNCH(mrd_image_fwd_kernel<float>, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, mel, (float*)img, T, F, FP,
                    total);
    else
        MSMC_LAUNCH(mrd_image_fwd_kernel<unsigned short>, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, mel,
                    (unsigned short*)img, T, F, FP, total);
    return msmc_check_launch();
}
int msmc_mrd_image_bwd_dt(const float* mel, const void* gimg, float* gmel, int B, int T, int F, int FP, int dtype,
                          msmc_stream stream) {
    if (!mel || !gimg || !gmel || B <= 0 || T <= 0 || F <= 0 || FP < F || dtype < 0 || dtype > 1) return MSMC_E_SHAPE;
    const long total = (long)B * T * FP;
    if (dtype == 0)
        MSMC_LAUNCH(mrd_image_bwd_kernel<float>, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, mel, (const float*)gimg,
                    gmel, T, F, FP, total);
    else
        MSMC_LAUNCH(mrd_image_bwd_kernel<unsigned short>, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, mel,
                    (const unsigned short*)gimg, gmel, T, F, FP, total);
    return msmc_check_launch();
}
int msmc_mrd_image_fwd(const float* mel, float* img, int B, int T, int F, int FP, msmc_stream stream) {
    return msmc_mrd_image_fwd_dt(mel, img, B, T, F, FP, 0, stream);
}
int msmc_mrd_image_bwd(const float* mel, const float* gimg, float* gmel, int B, int T, int F, int FP,
                       msmc_stream stream) {
    return msmc_mrd_image_bwd_dt(mel, gimg, gmel, B, T, F, FP, 0, stream);
}
int msmc_log_clamp_fwd(const float* x, float* y, long n, float lo, msmc_stream stream) {
    if (!x || !y || n <= 0) return MSMC_E_SHAPE;
    MSMC_LAUNCH(log_clamp_fwd_kernel, sp_grid(n), dim3(256), 0, (msmc_stream_t)stream, x, y, n, lo);
    return msmc_check_launch();
}
int msmc_log_clamp_bwd(const float* x, const float* g, float* gx, long n, float lo, msmc_stream stream) {
    if (!x || !g || !gx || n <= 0) return MSMC_E_SHAPE;
    MSMC_LAUNCH(log_clamp_bwd_kernel, sp_grid(n), dim3(256), 0, (msmc_stream_t)stream, x, g, gx, n, lo);
    return msmc_check_launch();
}


int msmc_wave_fan_fwd(const float* y, void* const* copies, const int* padded_len, int n, int B, int L, int dtype,
                      msmc_stream stream) {
    if (!y || !copies || !padded_len || n <= 0 || n > WF_MAX || B <= 0 || L <= 1 || dtype < 0 || dtype > 1) return MSMC_E_SHAPE;
    WaveFanArgs a;
    a.n16 = n;
    a.n32 = 0;
    int Lmax = 0;
    for (int k = 0; k < n; ++k) {
        if (!copies[k] || padded_len[k] < L || padded_len[k] > 2 * L - 1) return MSMC_E_SHAPE;   // (one reflection at most)
        a.p16[k] = copies[k];
        a.Lp[k] = padded_len[k];
        if (padded_len[k] > Lmax) Lmax = padded_len[k];
    }
    const long total = (long)B * Lmax;
    if (dtype == 0) MSMC_LAUNCH(wave_fan_fwd_kernel<float>, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, y, a, L, Lmax, total);
    else MSMC_LAUNCH(wave_fan_fwd_kernel<unsigned short>, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, y, a, L, Lmax, total);
    return msmc_check_launch();
}
int msmc_wave_fan_bwd(const float* const* g32, int n32, const void* const* gcopies, const int* padded_len, int n, float* gy,
                      int B, int L, int dtype, msmc_stream stream) {
    if (!gy || n < 0 || n > WF_MAX || n32 < 0 || n32 > WF_MAX || B <= 0 || L <= 1 || dtype < 0 || dtype > 1) return MSMC_E_SHAPE;
    if ((n && (!gcopies || !padded_len)) || (n32 && !g32)) return MSMC_E_SHAPE;
    WaveFanArgs a;
    a.n16 = n;
    a.n32 = n32;
    for (int k = 0; k < n; ++k) {
        if (padded_len[k] < L || padded_len[k] > 2 * L - 1) return MSMC_E_SHAPE;
        a.p16[k] = (void*)gcopies[k];
        a.Lp[k] = padded_len[k];
    }
    for (int k = 0; k < n32; ++k) a.p32[k] = g32[k];
    const long total = (long)B * L;
    if (dtype == 0) MSMC_LAUNCH(wave_fan_bwd_kernel<float>, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, a, gy, L, total);
    else MSMC_LAUNCH(wave_fan_bwd_kernel<unsigned short>, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, a, gy, L, total);
    return msmc_check_launch();
}
int msmc_window_gather(const long* starts, const float* wav, long* frames, float* target, int B, int nframes, int hop, long L,
                       msmc_stream stream) {
    if (!starts || !wav || !frames || !target || B <= 0 || nframes <= 0 || hop <= 0 || L < (long)nframes * hop) return MSMC_E_SHAPE;
    const long total = (long)B * nframes * hop;
    MSMC_LAUNCH(window_gather_kernel, sp_grid(total), dim3(256), 0, (msmc_stream_t)stream, starts, wav, frames, target, nframes, hop, L,
                total);
    return msmc_check_launch();
}
int msmc_spectral_multi(const msmc_spectral_op* ops, int n, msmc_stream stream) {
    if (!ops || n <= 0 || n > MSMC_SPECTRAL_MULTI_MAX) return MSMC_E_SHAPE;
    SpMultiArgs a;
    a.n = n;
    int blocks = 0;
    for (int k = 0; k < n; ++k) {
        const msmc_spectral_op& o = ops[k];
        long total;
        if (!o.a || !o.out) return MSMC_E_SHAPE;
        switch (o.kind) {
            case 0: case 1:
                if (o.B <= 0 || o.L <= o.pad || o.T <= 0 || o.NP < o.n_fft || o.hop <= 0 || o.pad < 0) return MSMC_E_SHAPE;
                total = o.kind == 0 ? (long)o.B * o.T * o.NP : (long)o.B * o.L;
                break;
            case 2: case 3:
                if (o.R <= 0 || o.F <= 0 || o.CP < 2 * o.F || o.FP < o.F || (o.kind == 3 && (!o.b || !o.c))) return MSMC_E_SHAPE;
                total = o.kind == 2 ? o.R * o.FP : o.R * o.CP;
                break;
            case 4: case 5:
                if (o.B <= 0 || o.T <= 0 || o.F <= 0 || o.FP < o.F || o.dtype < 0 || o.dtype > 1 || (o.kind == 5 && !o.b)) return MSMC_E_SHAPE;
                total = o.kind == 4 ? (long)o.B * o.F * o.T : (long)o.B * o.T * o.FP;
                break;
            case 6: case 7:
                if (o.R <= 0 || (o.kind == 7 && !o.b)) return MSMC_E_SHAPE;
                total = o.R;
                break;
            default: return MSMC_E_SHAPE;
        }
        a.first[k] = blocks;
        a.op[k] = o;
        blocks += (int)sp_grid(total).x;
    }
    a.first[n] = blocks;
    MSMC_LAUNCH(spectral_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream, a);
    return msmc_check_launch();
}
}  // extern "C"
