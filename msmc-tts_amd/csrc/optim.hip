// optim.hip -- multi-tensor gradient-norm clipping + AdamW in three launches per optimizer step.
//
// Replaces  torch.nn.utils.clip_grad_norm_(autoencoder.parameters(), 1.0)  followed by the per-child AdamW step of
//   Optimizer.step / VQGANTrainer.train_step   reference msmctts/trainers/optimizers/__init__.py:53-78,
//                                              msmctts/trainers/msmctts_trainer.py:205-206
// (PyTorch: ~10 foreach launches for the clip + 18 multi_tensor_apply launches for a 36.7 M-parameter child).
// Everything the update needs lives on the device (learning rate, step count, clip coefficient), so the three launches
// replay from a hipGraph unchanged.  All arithmetic fp32; reductions in a fixed order (bit-reproducible).
#include <msmc_rt.hpp>
#include <msmc_hip.h>

#define OPT_CHUNK 4096          // elements per workgroup

MSMC_DEV int opt_find(const msmc_opt_tensor* t, int n, int blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t[mid].first_chunk <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}
MSMC_DEV float opt_block_sum(float v, float* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = red[tid] + red[tid + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void opt_gradsq_kernel(const msmc_opt_tensor* __restrict__ table, int nt,
                                                         float* __restrict__ partial) {
    __shared__ float red[256];
    const msmc_opt_tensor t = table[opt_find(table, nt, blockIdx.x)];
    const long e0 = (long)(blockIdx.x - t.first_chunk) * OPT_CHUNK;
    float s = 0.f;
    if (e0 + OPT_CHUNK <= t.n && ((size_t)(t.g + e0) & 15) == 0) {
        // whole chunk, aligned: four 16-byte loads per work-item, all in flight at once.  The per-work-item order of the
        // additions differs from the scalar form below (elements 4k .. 4k+3 of a work-item are consecutive here); which form
        // a chunk takes depends on the tensor's size and alignment only, so a given model sums the same way every step
        f32x4 g4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g4[i] = *(const f32x4*)(t.g + e0 + threadIdx.x * 4 + 1024 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) s = fmaf(g4[i][q], g4[i][q], s);
    } else {
        for (int k = threadIdx.x; k < OPT_CHUNK; k += 256) {
            const long e = e0 + k;
            if (e < t.n) s = fmaf(t.g[e], t.g[e], s);
        }
    }
    s = opt_block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// one workgroup: norm = sqrt(sum of partials), coef = min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0);
// step += 1.  out[0] = norm, out[1] = coef.
__global__ __launch_bounds__(256) void opt_norm_kernel(const float* __restrict__ partial, int nblocks, float max_norm,
                                                       float* __restrict__ out, float* __restrict__ step) {
    __shared__ float red[256];
    float s = 0.f;
    if (max_norm > 0.f)
        for (int k = threadIdx.x; k < nblocks; k += 256) s = s + partial[k];
    s = opt_block_sum(s, red);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(s);
        float coef = 1.f;
        if (max_norm > 0.f) {
            coef = max_norm / (norm + 1e-6f);
            if (coef > 1.f) coef = 1.f;
        }
        out[0] = norm;
        out[1] = coef;
        step[0] = step[0] + 1.f;
    }
}

// torch.optim.AdamW (decoupled weight decay, bias correction by the step count):
//   g' = coef * g;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p = p (1 - lr wd) - lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void opt_adamw_kernel(const msmc_opt_tensor* __restrict__ table, int nt,
                                                        const float* __restrict__ norm_coef, const float* __restrict__ lr_ptr,
                                                        const float* __restrict__ step, float beta1, float beta2, float eps,
                                                        float wd, int write_grads) {
    const msmc_opt_tensor t = table[opt_find(table, nt, blockIdx.x)];
    const long e0 = (long)(blockIdx.x - t.first_chunk) * OPT_CHUNK;
    const float coef = norm_coef[1], lr = lr_ptr[0], tt = step[0];
    const float bc1 = 1.f - powf(beta1, tt), bc2 = 1.f - powf(beta2, tt);
    const float step_size = lr / bc1, rs2 = 1.f / sqrtf(bc2), decay = 1.f - lr * wd;
    // fast path: the whole 4096-element chunk lies inside the tensor and every operand is 16-byte aligned -- all sixteen vector
    // loads of a work-item are issued before the first store (a load behind a store to an array the compiler cannot tell apart
    // waits for nothing in hardware, but the compiler keeps program order and the loop ran four round trips deep)
    if (e0 + OPT_CHUNK <= t.n && (((size_t)(t.p + e0) | (size_t)(t.g + e0) | (size_t)(t.m + e0) | (size_t)(t.v + e0)) & 15) == 0) {
        f32x4 p4[4], g4[4], m4[4], v4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long e = e0 + threadIdx.x * 4 + 1024 * i;
            p4[i] = *(const f32x4*)(t.p + e);
            g4[i] = *(const f32x4*)(t.g + e);
            m4[i] = *(const f32x4*)(t.m + e);
            v4[i] = *(const f32x4*)(t.v + e);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long e = e0 + threadIdx.x * 4 + 1024 * i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float gq = g4[i][q] * coef;
                g4[i][q] = gq;
                m4[i][q] = beta1 * m4[i][q] + (1.f - beta1) * gq;
                v4[i][q] = beta2 * v4[i][q] + (1.f - beta2) * gq * gq;
                p4[i][q] = p4[i][q] * decay - step_size * (m4[i][q] / (sqrtf(v4[i][q]) * rs2 + eps));
            }
            *(f32x4*)(t.p + e) = p4[i];
            *(f32x4*)(t.m + e) = m4[i];
            *(f32x4*)(t.v + e) = v4[i];
            if (write_grads) *(f32x4*)(t.g + e) = g4[i];
        }
        return;
    }
    for (int k = threadIdx.x * 4; k < OPT_CHUNK; k += 1024) {
        const long e = e0 + k;
        if (e >= t.n) break;
        if (e + 4 <= t.n && (((size_t)(t.p + e) | (size_t)(t.g + e) | (size_t)(t.m + e) | (size_t)(t.v + e)) & 15) == 0) {
            f32x4 p = *(const f32x4*)(t.p + e), g = *(const f32x4*)(t.g + e), m = *(const f32x4*)(t.m + e), v = *(const f32x4*)(t.v + e);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float gq = g[q] * coef;
                g[q] = gq;
                m[q] = beta1 * m[q] + (1.f - beta1) * gq;
                v[q] = beta2 * v[q] + (1.f - beta2) * gq * gq;
                p[q] = p[q] * decay - step_size * (m[q] / (sqrtf(v[q]) * rs2 + eps));
            }
            *(f32x4*)(t.p + e) = p;
            *(f32x4*)(t.m + e) = m;
            *(f32x4*)(t.v + e) = v;
            if (write_grads) *(f32x4*)(t.g + e) = g;
        } else {
            for (long q = e; q < e + 4 && q < t.n; ++q) {
                const float gq = t.g[q] * coef;
                const float mq = beta1 * t.m[q] + (1.f - beta1) * gq;
                const float vq = beta2 * t.v[q] + (1.f - beta2) * gq * gq;
                t.p[q] = t.p[q] * decay - step_size * (mq / (sqrtf(vq) * rs2 + eps));
                t.m[q] = mq;
                t.v[q] = vq;
                if (write_grads) t.g[q] = gq;
            }
        }
    }
}

extern "C" {

int msmc_opt_chunk(void) { return OPT_CHUNK; }

int msmc_opt_clip_adamw(const msmc_opt_tensor* table, int ntensors, int nblocks, float max_norm, float* partial,
                        float* norm_coef, const float* lr, float* step, float beta1, float beta2, float eps,
                        float weight_decay, int write_grads, msmc_stream stream) {
    if (!table || ntensors <= 0 || nblocks <= 0 || !partial || !norm_coef || !lr || !step) return MSMC_E_SHAPE;
    if (max_norm > 0.f) {
        MSMC_LAUNCH(opt_gradsq_kernel, dim3((unsigned)nblocks), dim3(256), 0, (msmc_stream_t)stream, table, ntensors, partial);
        int rc = msmc_check_launch();
        if (rc) return rc;
    }
    MSMC_LAUNCH(opt_norm_kernel, dim3(1), dim3(256), 0, (msmc_stream_t)stream, (const float*)partial, nblocks, max_norm, norm_coef, step);
    int rc = msmc_check_launch();
    if (rc) return rc;
    MSMC_LAUNCH(opt_adamw_kernel, dim3((unsigned)nblocks), dim3(256), 0, (msmc_stream_t)stream, table, ntensors,
                (const float*)norm_coef, lr, (const float*)step, beta1, beta2, eps, weight_decay,
                (max_norm > 0.f && write_grads) ? 1 : 0);
    return msmc_check_launch();
}

}  // extern "C"
