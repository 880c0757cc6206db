// norm.hip -- fused element-wise / row-normalisation kernels of the FFT blocks and the quantiser glue.
//
// Replaces, in one launch each, the chains of stock kernels behind
//   MultiHeadAttention.forward / PositionwiseFeedForward.forward tails
//       layer_norm(dropout(h) + residual) [* non_pad_mask]        reference acoustic_models/transformer.py:262-266,318-323,352-356
//   the WaveNet gate  tanh(a) * sigmoid(b)                        reference vqgantts/modules.py:172-179
//   the Tanh between the 1x1 stacks of the quantiser              reference vqgantts/msmc_vqgan.py:115-136
// Activations are row-major [N][C] in fp32 (dtype 0) or bf16 (dtype 1); statistics and parameter gradients are fp32.
// Dropout masks are not stored: both passes derive them from a counter-based hash of (seed word on the device, call
// salt, element index), so a hipGraph replay draws fresh masks when the seed word is advanced inside the graph.
#include <msmc_rt.hpp>
#include <msmc_hip.h>

MSMC_DEV float nm_ld(const float* p, long i) { return p[i]; }
MSMC_DEV float nm_ld(const unsigned short* p, long i) { return bf16_bits_to_f32(p[i]); }
MSMC_DEV void nm_st(float* p, long i, float v) { p[i] = v; }
MSMC_DEV void nm_st(unsigned short* p, long i, float v) { p[i] = f32_to_bf16_bits(v); }

// keep-probability test: 32-bit mix of (key, element index) against the drop threshold
MSMC_DEV bool nm_keep(unsigned long long key, unsigned long long idx, unsigned int thresh) {
    unsigned long long z = key + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (unsigned int)(z >> 32) >= thresh;
}
MSMC_DEV unsigned long long nm_key(const long long* seed, long long salt) {
    const unsigned long long s = seed ? (unsigned long long)seed[0] : 0ull;
    return s * 0xD1342543DE82EF95ull + (unsigned long long)salt * 0x2545F4914F6CDD1Dull + 0x632BE59BD9B4E019ull;
}
MSMC_DEV float nm_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = v + wave_xor(v, m);
    return v;
}

#define NM_MAXE 16          // elements per lane: C <= 1024

// V consecutive elements of a row as one vector access (V = 4: 16 bytes of fp32, 8 bytes of bf16; V = 1: scalar)
template <int V> MSMC_DEV void nm_ldv(const float* p, long i, float (&o)[V]) {
    if constexpr (V == 4) {
        const f32x4 t = *(const f32x4*)(p + i);
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = t[q];
    } else {
        o[0] = p[i];
    }
}
template <int V> MSMC_DEV void nm_ldv(const unsigned short* p, long i, float (&o)[V]) {
    if constexpr (V == 4) {
        const u32x2 t = *(const u32x2*)(p + i);
        o[0] = __uint_as_float(t[0] << 16);
        o[1] = __uint_as_float(t[0] & 0xffff0000u);
        o[2] = __uint_as_float(t[1] << 16);
        o[3] = __uint_as_float(t[1] & 0xffff0000u);
    } else {
        o[0] = bf16_bits_to_f32(p[i]);
    }
}
template <int V> MSMC_DEV void nm_stv(float* p, long i, const float (&v)[V]) {
    if constexpr (V == 4) {
        f32x4 t;
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = v[q];
        *(f32x4*)(p + i) = t;
    } else {
        p[i] = v[0];
    }
}
template <int V> MSMC_DEV void nm_stv(unsigned short* p, long i, const float (&v)[V]) {
    if constexpr (V == 4) {
        u32x2 t;
        t[0] = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
        t[1] = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
        *(u32x2*)(p + i) = t;
    } else {
        p[i] = f32_to_bf16_bits(v[0]);
    }
}

// one wave per row: v = drop(x) + res; y = (v - mean) * rstd * gamma + beta, zeroed where keep_row == 0.
// A lane owns the V-element groups (lane + 64 jj) * V .. + V of its row (V = 4 when C % 4 == 0: vector accesses).
template <typename T, int V>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const unsigned char* __restrict__ keep_row, T* __restrict__ y,
                                                         T* __restrict__ v_out, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out, long N, int C, float eps,
                                                         float p_drop, const long long* seed, long long salt) {
    constexpr int NG = NM_MAXE / V;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const unsigned long long key = nm_key(seed, salt);
    const unsigned int thresh = p_drop > 0.f ? (unsigned int)(p_drop * 4294967296.0) : 0u;
    const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    float v[NG][V];
    float sum = 0.f;
#pragma unroll
    for (int jj = 0; jj < NG; ++jj) {
        const int c = (lane + 64 * jj) * V;
#pragma unroll
        for (int q = 0; q < V; ++q) v[jj][q] = 0.f;
        if (c < C) {
            const long i = row * C + c;
            float a[V], r[V];
            nm_ldv<V>(x, i, a);
            if (res) nm_ldv<V>(res, i, r);
#pragma unroll
            for (int q = 0; q < V; ++q) {
                if (thresh) a[q] = nm_keep(key, (unsigned long long)(i + q), thresh) ? a[q] * scale : 0.f;
                if (res) a[q] = a[q] + r[q];
                v[jj][q] = a[q];
                sum = sum + a[q];
            }
        }
    }
    const float mean = nm_wave_sum(sum) / C;
    float sq = 0.f;
#pragma unroll
    for (int jj = 0; jj < NG; ++jj)
        if ((lane + 64 * jj) * V < C)
#pragma unroll
            for (int q = 0; q < V; ++q) sq = fmaf(v[jj][q] - mean, v[jj][q] - mean, sq);
    const float rstd = 1.f / sqrtf(nm_wave_sum(sq) / C + eps);
    const bool live = !keep_row || keep_row[row] != 0;
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int jj = 0; jj < NG; ++jj) {
        const int c = (lane + 64 * jj) * V;
        if (c < C) {
            const long i = row * C + c;
            float gm[V], bt[V], o[V];
            nm_ldv<V>(gamma, c, gm);
            nm_ldv<V>(beta, c, bt);
#pragma unroll
            for (int q = 0; q < V; ++q) o[q] = live ? (v[jj][q] - mean) * rstd * gm[q] + bt[q] : 0.f;
            nm_stv<V>(v_out, i, v[jj]);
            nm_stv<V>(y, i, o);
        }
    }
}

// ---- the attention sub-layer's tail in ONE launch (round 6): h = a W^T + bias (the output projection ``fc``, reference
// acoustic_models/transformer.py:259-262) and layer_norm(dropout(h) + residual) * non_pad_mask (:262-266) -- before: a 1-tap GEMM
// launch writing h and the fused add + LayerNorm launch reading it back, each ~6 us for a fraction of a microsecond of work, 12
// times per forward pass.  A workgroup owns SIXTEEN whole rows: v_mfma_f32_16x16x32_bf16 with the operand roles swapped (A = weight
// rows, B = activation rows), so a lane ends up with runs of four consecutive channels of ONE row -- bias, dropout, residual,
// both LayerNorm reductions (lane-local, two cross-lane steps, one LDS exchange between the four waves) and the 8-byte stores all
// happen in registers.  Both operands come straight from global memory as 16-byte fragments (K = heads x d_v = 128: the whole
// contraction is four MFMA steps; the 64 KB weight matrix stays in L2).  h is rounded to bf16 before the dropout exactly as the
// GEMM's epilogue rounds it, so v (the saved pre-normalisation sum) is bit-identical to the two-launch chain's and the
// statistics differ only by the order of their sums.  Wave w owns the 16-channel tiles w, w + 4, ..: NT of them.
// Every global read of a lane is issued before the first use (the weight fragments of KU contraction steps at a time, the
// activation fragments, residual, bias, gamma, beta, the row mask): addresses past the edges are clamped instead of branched
// around, so the kernel pays ~two memory latencies, not one per load.
template <int NT, int KU>
__global__ __launch_bounds__(256) void fc_add_ln_fwd_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ W,
                                                            const float* __restrict__ bias, const unsigned short* __restrict__ res,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const unsigned char* __restrict__ keep_row, unsigned short* __restrict__ y,
                                                            unsigned short* __restrict__ v_out, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, long N, int C, int K, float eps,
                                                            float p_drop, const long long* seed, long long salt) {
    __shared__ float red[2][4][16];
    const int tid = threadIdx.x, w = wave_uniform(tid >> 6), lane = tid & 63, j = lane & 15, g = lane >> 4;
    const long row = (long)blockIdx.x * 16 + j;
    const long rl = row < N ? row : N - 1;
    const int MT = (C + 15) >> 4, nku = K / (32 * KU);
    const unsigned short* arow = a + rl * K + 8 * g;
    const unsigned short* wrow[NT];
    int c0[NT], cc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        int t = w + 4 * i;
        if (t >= MT) t = MT - 1;                          // (tiles past the last: a valid tile, results unused)
        int ch = 16 * t + j;
        if (ch >= C) ch = C - 1;
        wrow[i] = W + (size_t)ch * K + 8 * g;
        c0[i] = 16 * (w + 4 * i) + 4 * g;                 // this lane's four channels of tile i
        cc[i] = c0[i] < C ? c0[i] : C - 4;
    }
    // first batch of fragments, then everything the epilogue reads
    u32x4 bf[KU], af[NT][KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) bf[u] = *(const u32x4*)(arow + 32 * u);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int u = 0; u < KU; ++u) af[i][u] = *(const u32x4*)(wrow[i] + 32 * u);
    float r[NT][4], bs[NT][4], gm[NT][4], bt[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        nm_ldv<4>(res, rl * C + cc[i], r[i]);
        nm_ldv<4>(bias, cc[i], bs[i]);
        nm_ldv<4>(gamma, cc[i], gm[i]);
        nm_ldv<4>(beta, cc[i], bt[i]);
    }
    const bool live = !keep_row || keep_row[rl] != 0;
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < nku; ++kb) {
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int i = 0; i < NT; ++i)
                acc[i] = mfma_bf16_16x16x32(__builtin_bit_cast(bf16x8, af[i][u]), __builtin_bit_cast(bf16x8, bf[u]), acc[i]);
        if (kb + 1 < nku) {
            const int k0 = 32 * KU * (kb + 1);
#pragma unroll
            for (int u = 0; u < KU; ++u) bf[u] = *(const u32x4*)(arow + k0 + 32 * u);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int u = 0; u < KU; ++u) af[i][u] = *(const u32x4*)(wrow[i] + k0 + 32 * u);
        }
    }
    const unsigned long long key = nm_key(seed, salt);
    const unsigned int thresh = p_drop > 0.f ? (unsigned int)(p_drop * 4294967296.0) : 0u;
    const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    float v[NT][4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const bool in = c0[i] < C;
        const long e0 = rl * C + cc[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float h = acc[i][q] + bs[i][q];
            h = bf16_bits_to_f32(f32_to_bf16_bits(h));                           // (the GEMM's epilogue stores h in bf16)
            if (thresh) h = nm_keep(key, (unsigned long long)(e0 + q), thresh) ? h * scale : 0.f;
            h = h + r[i][q];
            v[i][q] = in ? h : 0.f;
            sum = sum + v[i][q];
        }
    }
    sum = sum + wave_xor(sum, 16);
    sum = sum + wave_xor(sum, 32);
    if (g == 0) red[0][w][j] = sum;
    __syncthreads();
    const float mean = (((red[0][0][j] + red[0][1][j]) + red[0][2][j]) + red[0][3][j]) / C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const bool in = c0[i] < C;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float dv = in ? v[i][q] - mean : 0.f;
            sq = fmaf(dv, dv, sq);
        }
    }
    sq = sq + wave_xor(sq, 16);
    sq = sq + wave_xor(sq, 32);
    if (g == 0) red[1][w][j] = sq;
    __syncthreads();
    const float rstd = 1.f / sqrtf((((red[1][0][j] + red[1][1][j]) + red[1][2][j]) + red[1][3][j]) / C + eps);
    if (row >= N) return;
    if (w == 0 && g == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (c0[i] < C) {
            const long e0 = row * C + c0[i];
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = live ? (v[i][q] - mean) * rstd * gm[i][q] + bt[i][q] : 0.f;
            nm_stv<4>(v_out, e0, v[i]);
            nm_stv<4>(y, e0, o);
        }
    }
}

// backward: gy = g * live; dxhat = gy * gamma; dv = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat));
// gres = dv; gx = dv * dropmask * scale.  Parameter-gradient partials per workgroup: part[block][0][c] = sum gy*xhat,
// part[block][1][c] = sum gy (reduced in a fixed order by add_ln_param_kernel).
// NG: 64 V-element lane groups per row (C <= 64 V NG: the registers of a 256-wide row are a quarter of a 1024-wide one's), RW: rows
// a wave keeps in flight -- all loads of RW rows issued before the first reduction.  Measured at 25 600 x 600 / x 1024 bf16
// (profiles/r05_norm_kernels.txt): RW = 2 is SLOWER than one row at a time (56 against 48 us, 106 against 86 us: the registers
// cost more occupancy than the second row buys), 4 / 8 / 12 / 16 rows per workgroup all land within 3 % -- the pass streams at
// 2.6 TB/s whatever the chain length.  Production: RW = 1.  Rows are accumulated into the parameter partials in row order.
template <typename T, int V, int NG, int RW>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const T* __restrict__ g, const T* __restrict__ v_in,
                                                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                         const float* __restrict__ gamma,
                                                         const unsigned char* __restrict__ keep_row, T* __restrict__ gx,
                                                         T* __restrict__ gres, float* __restrict__ part, long N, int C,
                                                         float p_drop, const long long* seed, long long salt, int rows_per_block) {
    MSMC_DYN_LDS(smem);
    float* acc = (float*)smem;                     // [4 waves][2][C]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long key = nm_key(seed, salt);
    const unsigned int thresh = p_drop > 0.f ? (unsigned int)(p_drop * 4294967296.0) : 0u;
    const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    float dg[NG][V], dbt[NG][V], gm[NG][V];
#pragma unroll
    for (int jj = 0; jj < NG; ++jj) {
        const int c = (lane + 64 * jj) * V;
#pragma unroll
        for (int q = 0; q < V; ++q) dg[jj][q] = dbt[jj][q] = gm[jj][q] = 0.f;
        if (c < C) nm_ldv<V>(gamma, c, gm[jj]);
    }
    const long r0 = (long)blockIdx.x * rows_per_block;
    long rend = r0 + rows_per_block;
    if (rend > N) rend = N;
    for (long rbase = r0 + w; rbase < rend; rbase += 4 * RW) {
        float dx[RW][NG][V], xh[RW][NG][V];          // loaded as (gy, v), turned into (dxhat, xhat) in place
        bool live[RW];
        float mean[RW], rstd[RW];
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            const long row = rbase + 4 * u;
            const bool has = row < rend;
            live[u] = has && (!keep_row || keep_row[has ? row : r0] != 0);
            mean[u] = has ? mean_in[row] : 0.f;
            rstd[u] = has ? rstd_in[row] : 0.f;
#pragma unroll
            for (int jj = 0; jj < NG; ++jj) {
                const int c = (lane + 64 * jj) * V;
#pragma unroll
                for (int q = 0; q < V; ++q) dx[u][jj][q] = xh[u][jj][q] = 0.f;
                if (has && c < C) {
                    nm_ldv<V>(g, row * C + c, dx[u][jj]);
                    nm_ldv<V>(v_in, row * C + c, xh[u][jj]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            const long row = rbase + 4 * u;
            if (row >= rend) break;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int jj = 0; jj < NG; ++jj) {
                const int c = (lane + 64 * jj) * V;
                if (c < C) {
#pragma unroll
                    for (int q = 0; q < V; ++q) {
                        const float gy = live[u] ? dx[u][jj][q] : 0.f;
                        const float x_ = (xh[u][jj][q] - mean[u]) * rstd[u];
                        const float d_ = gy * gm[jj][q];
                        xh[u][jj][q] = x_;
                        dx[u][jj][q] = d_;
                        s1 = s1 + d_;
                        s2 = fmaf(d_, x_, s2);
                        dg[jj][q] = fmaf(gy, x_, dg[jj][q]);
                        dbt[jj][q] = dbt[jj][q] + gy;
                    }
                }
            }
            const float m1 = nm_wave_sum(s1) / C, m2 = nm_wave_sum(s2) / C;
#pragma unroll
            for (int jj = 0; jj < NG; ++jj) {
                const int c = (lane + 64 * jj) * V;
                if (c < C) {
                    const long i = row * C + c;
                    float dv[V], d[V];
#pragma unroll
                    for (int q = 0; q < V; ++q) {
                        dv[q] = rstd[u] * (dx[u][jj][q] - m1 - xh[u][jj][q] * m2);
                        d[q] = dv[q];
                        if (thresh) d[q] = nm_keep(key, (unsigned long long)(i + q), thresh) ? dv[q] * scale : 0.f;
                    }
                    if (gres) nm_stv<V>(gres, i, dv);
                    nm_stv<V>(gx, i, d);
                }
            }
        }
    }
#pragma unroll
    for (int jj = 0; jj < NG; ++jj) {
        const int c = (lane + 64 * jj) * V;
        if (c < C)
#pragma unroll
            for (int q = 0; q < V; ++q) {
                acc[(w * 2 + 0) * C + c + q] = dg[jj][q];
                acc[(w * 2 + 1) * C + c + q] = dbt[jj][q];
            }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
        const int k = e / C, c = e - k * C;
        part[((size_t)blockIdx.x * 2 + k) * C + c] =
            ((acc[(0 * 2 + k) * C + c] + acc[(1 * 2 + k) * C + c]) + acc[(2 * 2 + k) * C + c]) + acc[(3 * 2 + k) * C + c];
    }
}

// dgamma[c] (+)= sum_b part[b][0][c], dbeta[c] (+)= sum_b part[b][1][c].  A workgroup owns 16 columns of one of the two
// vectors; its 16 slices each sum a contiguous range of blocks in order, then the slices are added in order (deterministic).
MSMC_DEV void add_ln_param_body(const float* __restrict__ part, int nblocks, int C, float* __restrict__ dgamma,
                                float* __restrict__ dbeta, int accumulate, const int block) {
    __shared__ float red[16][16];
    const int groups = (C + 15) / 16;
    const int k = block / groups, c = (block - k * groups) * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;
    const int per = (nblocks + 15) / 16, b0 = sl * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    float s = 0.f;
    if (c < C)
        for (int b = b0; b < b1; ++b) s = s + part[((size_t)b * 2 + k) * C + c];
    red[sl][threadIdx.x & 15] = s;
    __syncthreads();
    if (sl == 0 && c < C) {
        float t = red[0][threadIdx.x];
#pragma unroll
        for (int q = 1; q < 16; ++q) t = t + red[q][threadIdx.x];
        float* dst = k == 0 ? dgamma : dbeta;
        dst[c] = accumulate ? dst[c] + t : t;
    }
}
__global__ __launch_bounds__(256) void add_ln_param_kernel(const float* __restrict__ part, int nblocks, int C,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int accumulate) {
    add_ln_param_body(part, nblocks, C, dgamma, dbeta, accumulate, (int)blockIdx.x);
}
// the same reduction for up to MSMC_LN_PARAM_MAX LayerNorms in one launch (the backward passes of a step leave their partials
// in caller-kept workspaces; one launch at the end of the pass instead of one per LayerNorm): item i owns blocks
// [first[i], first[i + 1])
struct LnParamArgs {
    msmc_ln_param_item item[MSMC_LN_PARAM_MAX];
    int first[MSMC_LN_PARAM_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void add_ln_param_multi_kernel(LnParamArgs a) {
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.first[i + 1]) ++i;
    const msmc_ln_param_item& it = a.item[i];
    add_ln_param_body(it.part, it.nblocks, it.C, it.dgamma, it.dbeta, it.accumulate, (int)blockIdx.x - a.first[i]);
}

// ---- gate / tanh ---------------------------------------------------------------------------------------------
// x [N][2C] -> y [N][C] = tanh(x[:, :C]) * sigmoid(x[:, C:])   (optionally dropped out);  backward from x
template <typename T>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long N, int C, float p_drop,
                                                       const long long* seed, long long salt) {
    const long total = N * C;
    const unsigned long long key = nm_key(seed, salt);
    const unsigned int thresh = p_drop > 0.f ? (unsigned int)(p_drop * 4294967296.0) : 0u;
    const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / C;
        const int c = (int)(i - n * C);
        const float a = nm_ld(x, n * 2 * C + c), b = nm_ld(x, n * 2 * C + C + c);
        float v = tanhf(a) * (1.f / (1.f + expf(-b)));
        if (thresh) v = nm_keep(key, (unsigned long long)i, thresh) ? v * scale : 0.f;
        nm_st(y, i, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ gx, long N,
                                                       int C, float p_drop, const long long* seed, long long salt) {
    const long total = N * C;
    const unsigned long long key = nm_key(seed, salt);
    const unsigned int thresh = p_drop > 0.f ? (unsigned int)(p_drop * 4294967296.0) : 0u;
    const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / C;
        const int c = (int)(i - n * C);
        const float a = nm_ld(x, n * 2 * C + c), b = nm_ld(x, n * 2 * C + C + c);
        float gv = nm_ld(g, i);
        if (thresh) gv = nm_keep(key, (unsigned long long)i, thresh) ? gv * scale : 0.f;
        const float t = tanhf(a), s = 1.f / (1.f + expf(-b));
        nm_st(gx, n * 2 * C + c, gv * s * (1.f - t * t));
        nm_st(gx, n * 2 * C + C + c, gv * t * s * (1.f - s));
    }
}
// y = tanh(x);  gx = g * (1 - y*y)   (n elements)
template <typename T>
__global__ __launch_bounds__(256) void tanh_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) nm_st(y, i, tanhf(nm_ld(x, i)));
}
template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const T* __restrict__ y, const T* __restrict__ g, T* __restrict__ gx, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float t = nm_ld(y, i);
        nm_st(gx, i, nm_ld(g, i) * (1.f - t * t));
    }
}

// the vocoder's output activation (reference hifigan/generator.py:52-54: tanh of the fp32 waveform): the compute-dtype output of the
// last convolution in, fp32 out -- cast + tanh were two launches forward and two backward (the gradient leaves in the compute dtype)
template <typename T>
__global__ __launch_bounds__(256) void tanh_f32_fwd_kernel(const T* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = tanhf(nm_ld(x, i));
}
template <typename T>
__global__ __launch_bounds__(256) void tanh_f32_bwd_kernel(const float* __restrict__ y, const float* __restrict__ g, T* __restrict__ gx, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float t = y[i];
        nm_st(gx, i, g[i] * (1.f - t * t));
    }
}


// ---- small glue of the quantiser and the generator's backward pass (round 6: stock element-wise launches of the step) ----------------
// sum_n: out = ((a + b) + c) + d in fp32, rounded once (c, d optional) -- the input gradients of the parallel ResBlocks of a generator
// stage (reference hifigan/generator.py:47-52: one tensor feeds num_kernels blocks; the autograd engine adds their gradients pairwise).
// dropout_add: y = dropout(x) + res with the counter-hash masks of this file (res optional); backward gx = g * mask * scale -- the
// quantiser's F.dropout + residual add (reference vqgantts/msmc_vqgan.py:141-176).  row_mask: keep[b][t] = t < len[b] in the compute
// dtype (reference utils/utils.py:9-16 get_mask_from_lengths, inverted and cast: four stock launches).
template <typename T>
__global__ __launch_bounds__(256) void sum_n_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
                                                    const T* __restrict__ d, T* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float va[4], vb[4], vc[4], vd[4], o[4];
        nm_ldv<4>(a, 4 * i, va);
        nm_ldv<4>(b, 4 * i, vb);
        if (c) nm_ldv<4>(c, 4 * i, vc);
        if (d) nm_ldv<4>(d, 4 * i, vd);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o[q] = va[q] + vb[q];
            if (c) o[q] = o[q] + vc[q];
            if (d) o[q] = o[q] + vd[q];
        }
        nm_stv<4>(out, 4 * i, o);
    }
}
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void dropout_add_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, long n4,
                                                          float p_drop, const long long* seed, long long salt) {
    const unsigned long long key = nm_key(seed, salt);
    const unsigned int thresh = p_drop > 0.f ? (unsigned int)(p_drop * 4294967296.0) : 0u;
    const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float v[4], r[4];
        nm_ldv<4>(x, 4 * i, v);
        if (!BWD && res) nm_ldv<4>(res, 4 * i, r);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (thresh) v[q] = nm_keep(key, (unsigned long long)(4 * i + q), thresh) ? v[q] * scale : 0.f;
            if (!BWD && res) v[q] = v[q] + r[q];
        }
        nm_stv<4>(y, 4 * i, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void row_mask_kernel(const void* __restrict__ lengths, int is64, T* __restrict__ keep, int B, int Tn) {
    const long n = (long)B * Tn;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int b = (int)(i / Tn), t = (int)(i - (long)b * Tn);
        const long len = is64 ? (long)((const long long*)lengths)[b] : (long)((const int*)lengths)[b];
        nm_st(keep, i, t < len ? 1.f : 0.f);
    }
}

// ---- FFT-stack prologue ---------------------------------------------------------------------------------------
// The head of FFTBlocks.forward (reference acoustic_models/transformer.py:375-395) with the positions of
// vqgantts/msmc_vqgan.py:56-58 (1 .. len per utterance, 0 on padding) folded in -- one launch instead of the chain
// arange / repeat / compare / masked_fill / embedding gather / add / cast / ne / fill / masked_fill:
//   out[b][t][:]   = seq[b][t][:] + table[t < len[b] ? t + 1 : 0][:]      (fp32 sum, rounded once to the output dtype)
//   keep_row[b T + t] = t < len[b]                                        (the non-pad mask: rows of 0 / 1)
//   key_bias[b][t']  = t' < len[b] ? 0 : -inf      for t' < Tp            (additive key-padding bias of csrc/attn.hip)
// One wave per row, V-element vector accesses as the LayerNorm kernels.
template <typename TI, typename TO, int V>
__global__ __launch_bounds__(256) void fft_prologue_kernel(const TI* __restrict__ seq, const void* __restrict__ lengths,
                                                           int len64, const float* __restrict__ table, TO* __restrict__ out,
                                                           unsigned char* __restrict__ keep_row, float* __restrict__ key_bias,
                                                           int B, int T, int C, int Tp) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + w;
    if (row >= (long)B * T) return;
    const int b = (int)(row / T), t = (int)(row - (long)b * T);
    const long len = len64 ? (long)((const long long*)lengths)[b] : (long)((const int*)lengths)[b];
    const bool live = t < len;
    const float* trow = table + (size_t)(live ? t + 1 : 0) * C;
    for (int c = lane * V; c < C; c += 64 * V) {
        float a[V], e[V], o[V];
        nm_ldv<V>(seq, row * C + c, a);
        nm_ldv<V>(trow, c, e);
#pragma unroll
        for (int q = 0; q < V; ++q) o[q] = a[q] + e[q];
        nm_stv<V>(out, row * C + c, o);
    }
    const float ninf = -__builtin_huge_valf();
    if (lane == 0) {
        keep_row[row] = live ? 1 : 0;
        if (key_bias) key_bias[(size_t)b * Tp + t] = live ? 0.f : ninf;
    }
    if (key_bias && t == T - 1 && lane < Tp - T) key_bias[(size_t)b * Tp + T + lane] = ninf;      // (Tp - T < 32)
}

static int nm_grid(long n) {
    long b = (n + 255) / 256;
    const long cap = 8L * MSMC_NUM_CU;
    return (int)(b < 1 ? 1 : b > cap ? cap : b);
}

extern "C" {

int msmc_add_ln_fwd(const void* x, const void* res, const float* gamma, const float* beta, const unsigned char* keep_row,
                    void* y, void* v, float* mean, float* rstd, long N, int C, float eps, float p_drop,
                    const long long* seed, long long salt, int dtype, msmc_stream stream) {
    if (!x || !gamma || !beta || !y || !v || !mean || !rstd || N < 0 || C <= 0 || C > 64 * NM_MAXE || p_drop < 0.f || p_drop >= 1.f)
        return MSMC_E_SHAPE;
    if (N == 0) return 0;
    const dim3 grid((unsigned)((N + 3) / 4));
#define NM_FWD(T_, V_)                                                                                                  \
    MSMC_LAUNCH((add_ln_fwd_kernel<T_, V_>), grid, dim3(256), 0, (msmc_stream_t)stream, (const T_*)x, (const T_*)res, gamma, \
                beta, keep_row, (T_*)y, (T_*)v, mean, rstd, N, C, eps, p_drop, seed, salt)
    const bool vec = (C % 4) == 0;          // rows start 8 / 16-byte aligned: vector accesses
    if (dtype == 0) { if (vec) NM_FWD(float, 4); else NM_FWD(float, 1); }
    else if (dtype == 1) { if (vec) NM_FWD(unsigned short, 4); else NM_FWD(unsigned short, 1); }
    else return MSMC_E_SHAPE;
#undef NM_FWD
    return msmc_check_launch();
}

// the fused output projection + add + LayerNorm of the attention sub-layer (bf16): a [N][K], W [C][K] (the projection's forward
// kernel-layout slice), bias [C] fp32, res / y / v [N][C]; K % 32 == 0, C % 4 == 0, C <= 640; bias / gamma / beta 16-byte aligned.
int msmc_fc_add_ln_fwd(const void* a, const void* W, const float* bias, const void* res, const float* gamma, const float* beta,
                       const unsigned char* keep_row, void* y, void* v, float* mean, float* rstd, long N, int C, int K, float eps,
                       float p_drop, const long long* seed, long long salt, msmc_stream stream) {
    if (!a || !W || !bias || !res || !gamma || !beta || !y || !v || !mean || !rstd || N < 0 || C <= 0 || (C & 3) || C > 640 || K <= 0 ||
        (K & 31) || p_drop < 0.f || p_drop >= 1.f)
        return MSMC_E_SHAPE;
    if ((((size_t)a) | ((size_t)W) | ((size_t)bias) | ((size_t)gamma) | ((size_t)beta)) & 15) return MSMC_E_SHAPE;
    if ((((size_t)res) | ((size_t)y) | ((size_t)v)) & 7) return MSMC_E_SHAPE;
    if (N == 0) return 0;
    const dim3 grid((unsigned)((N + 15) / 16));
#define NM_FC(NT_, KU_)                                                                                                     \
    MSMC_LAUNCH((fc_add_ln_fwd_kernel<NT_, KU_>), grid, dim3(256), 0, (msmc_stream_t)stream, (const unsigned short*)a,      \
                (const unsigned short*)W, bias, (const unsigned short*)res, gamma, beta, keep_row, (unsigned short*)y,      \
                (unsigned short*)v, mean, rstd, N, C, K, eps, p_drop, seed, salt)
    if (C <= 256) {
        if ((K & 127) == 0) NM_FC(4, 4);
        else if ((K & 63) == 0) NM_FC(4, 2);
        else NM_FC(4, 1);
    } else {
        if ((K & 63) == 0) NM_FC(10, 2);
        else NM_FC(10, 1);
    }
#undef NM_FC
    return msmc_check_launch();
}

#define NM_BWD_ROWS 16          // rows per workgroup of the backward pass (4 per wave)
size_t msmc_add_ln_bwd_workspace(long N, int C) {
    const long rows = NM_BWD_ROWS;
    return (size_t)((N + rows - 1) / rows) * 2 * C * sizeof(float);
}

int msmc_add_ln_bwd(const void* g, const void* v, const float* mean, const float* rstd, const float* gamma,
                    const unsigned char* keep_row, void* gx, void* gres, float* dgamma, float* dbeta, void* workspace,
                    size_t workspace_bytes, long N, int C, float p_drop, const long long* seed, long long salt, int accumulate,
                    int dtype, msmc_stream stream) {
    if (!g || !v || !mean || !rstd || !gamma || !gx || (!dgamma != !dbeta) || N < 0 || C <= 0 || C > 64 * NM_MAXE) return MSMC_E_SHAPE;
    const int rows = NM_BWD_ROWS;
    const int nblocks = (int)((N + rows - 1) / rows);
    if (workspace_bytes < msmc_add_ln_bwd_workspace(N, C) || (nblocks && !workspace)) return MSMC_E_WORKSPACE;
    const size_t lds = (size_t)4 * 2 * C * sizeof(float);
    if (nblocks) {
#define NM_BWD(T_, V_, NG_, RW_)                                                                                        \
    MSMC_LAUNCH((add_ln_bwd_kernel<T_, V_, NG_, RW_>), dim3((unsigned)nblocks), dim3(256), lds, (msmc_stream_t)stream, (const T_*)g, \
                (const T_*)v, mean, rstd, gamma, keep_row, (T_*)gx, (T_*)gres, (float*)workspace, N, C, p_drop, seed, salt, rows)
#define NM_BWD_VEC(T_)                                                                                                  \
    do {                                                                                                                \
        if (C <= 256) NM_BWD(T_, 4, 1, 1);                                                                              \
        else if (C <= 512) NM_BWD(T_, 4, 2, 1);                                                                         \
        else if (C <= 768) NM_BWD(T_, 4, 3, 1);                                                                         \
        else NM_BWD(T_, 4, 4, 1);                                                                                       \
    } while (0)
        const bool vec = (C % 4) == 0;
        if (dtype == 0) { if (vec) NM_BWD_VEC(float); else NM_BWD(float, 1, 16, 1); }
        else if (dtype == 1) { if (vec) NM_BWD_VEC(unsigned short); else NM_BWD(unsigned short, 1, 16, 1); }
        else return MSMC_E_SHAPE;
#undef NM_BWD_VEC
#undef NM_BWD
        int rc = msmc_check_launch();
        if (rc) return rc;
    }
    if (!dgamma) return 0;           // the partials stay in the workspace: msmc_add_ln_param_multi reduces them later
    MSMC_LAUNCH(add_ln_param_kernel, dim3((unsigned)(2 * ((C + 15) / 16))), dim3(256), 0, (msmc_stream_t)stream,
                (const float*)workspace, nblocks, C, dgamma, dbeta, accumulate);
    return msmc_check_launch();
}

int msmc_add_ln_param_multi(const msmc_ln_param_item* items, int nitems, msmc_stream stream) {
    if (nitems < 0 || (nitems && !items)) return MSMC_E_SHAPE;
    for (int i0 = 0; i0 < nitems; i0 += MSMC_LN_PARAM_MAX) {
        LnParamArgs a;
        a.n = nitems - i0 < MSMC_LN_PARAM_MAX ? nitems - i0 : MSMC_LN_PARAM_MAX;
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const msmc_ln_param_item& it = items[i0 + i];
            if (!it.part || !it.dgamma || !it.dbeta || it.nblocks < 0 || it.C <= 0 || it.C > 64 * NM_MAXE) return MSMC_E_SHAPE;
            a.item[i] = it;
            a.first[i] = blocks;
            blocks += 2 * ((it.C + 15) / 16);
        }
        a.first[a.n] = blocks;
        MSMC_LAUNCH(add_ln_param_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (msmc_stream_t)stream, a);
        int rc = msmc_check_launch();
        if (rc) return rc;
    }
    return 0;
}

int msmc_fft_prologue(const void* seq, const void* lengths, int len_is_64, const float* table, int table_rows, void* out,
                      unsigned char* keep_row, float* key_bias, int B, int T, int C, int Tp, int in_dtype, int out_dtype,
                      msmc_stream stream) {
    if (!seq || !lengths || !table || !out || !keep_row || B <= 0 || T <= 0 || C <= 0 || T + 1 > table_rows) return MSMC_E_SHAPE;
    if (key_bias && (Tp < T || Tp - T >= 64)) return MSMC_E_SHAPE;
    if (in_dtype < 0 || in_dtype > 1 || out_dtype < 0 || out_dtype > 1) return MSMC_E_SHAPE;
    const dim3 grid((unsigned)(((long)B * T + 3) / 4));
#define NM_PRO(TI_, TO_, V_)                                                                                           \
    MSMC_LAUNCH((fft_prologue_kernel<TI_, TO_, V_>), grid, dim3(256), 0, (msmc_stream_t)stream, (const TI_*)seq, lengths, \
                len_is_64, table, (TO_*)out, keep_row, key_bias, B, T, C, Tp)
    const bool vec = (C % 4) == 0;
    if (in_dtype == 0 && out_dtype == 0) { if (vec) NM_PRO(float, float, 4); else NM_PRO(float, float, 1); }
    else if (in_dtype == 0) { if (vec) NM_PRO(float, unsigned short, 4); else NM_PRO(float, unsigned short, 1); }
    else if (out_dtype == 0) { if (vec) NM_PRO(unsigned short, float, 4); else NM_PRO(unsigned short, float, 1); }
    else { if (vec) NM_PRO(unsigned short, unsigned short, 4); else NM_PRO(unsigned short, unsigned short, 1); }
#undef NM_PRO
    return msmc_check_launch();
}

int msmc_gate_fwd(const void* x, void* y, long N, int C, float p_drop, const long long* seed, long long salt, int dtype,
                  msmc_stream stream) {
    if (!x || !y || N < 0 || C <= 0 || p_drop < 0.f || p_drop >= 1.f) return MSMC_E_SHAPE;
    if (N == 0) return 0;
    const dim3 grid((unsigned)nm_grid(N * C));
    if (dtype == 0) MSMC_LAUNCH(gate_fwd_kernel<float>, grid, dim3(256), 0, (msmc_stream_t)stream, (const float*)x, (float*)y, N, C, p_drop, seed, salt);
    else if (dtype == 1) MSMC_LAUNCH(gate_fwd_kernel<unsigned short>, grid, dim3(256), 0, (msmc_stream_t)stream, (const unsigned short*)x, (unsigned short*)y, N, C, p_drop, seed, salt);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}
int msmc_gate_bwd(const void* x, const void* g, void* gx, long N, int C, float p_drop, const long long* seed, long long salt,
                  int dtype, msmc_stream stream) {
    if (!x || !g || !gx || N < 0 || C <= 0) return MSMC_E_SHAPE;
    if (N == 0) return 0;
    const dim3 grid((unsigned)nm_grid(N * C));
    if (dtype == 0) MSMC_LAUNCH(gate_bwd_kernel<float>, grid, dim3(256), 0, (msmc_stream_t)stream, (const float*)x, (const float*)g, (float*)gx, N, C, p_drop, seed, salt);
    else if (dtype == 1) MSMC_LAUNCH(gate_bwd_kernel<unsigned short>, grid, dim3(256), 0, (msmc_stream_t)stream, (const unsigned short*)x, (const unsigned short*)g, (unsigned short*)gx, N, C, p_drop, seed, salt);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}
int msmc_tanh_fwd(const void* x, void* y, long n, int dtype, msmc_stream stream) {
    if (!x || !y || n < 0) return MSMC_E_SHAPE;
    if (n == 0) return 0;
    const dim3 grid((unsigned)nm_grid(n));
    if (dtype == 0) MSMC_LAUNCH(tanh_fwd_kernel<float>, grid, dim3(256), 0, (msmc_stream_t)stream, (const float*)x, (float*)y, n);
    else if (dtype == 1) MSMC_LAUNCH(tanh_fwd_kernel<unsigned short>, grid, dim3(256), 0, (msmc_stream_t)stream, (const unsigned short*)x, (unsigned short*)y, n);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}
int msmc_tanh_f32_fwd(const void* x, float* y, long n, int dtype, msmc_stream stream) {
    if (!x || !y || n < 0) return MSMC_E_SHAPE;
    if (n == 0) return 0;
    const dim3 grid((unsigned)nm_grid(n));
    if (dtype == 0) MSMC_LAUNCH(tanh_f32_fwd_kernel<float>, grid, dim3(256), 0, (msmc_stream_t)stream, (const float*)x, y, n);
    else if (dtype == 1) MSMC_LAUNCH(tanh_f32_fwd_kernel<unsigned short>, grid, dim3(256), 0, (msmc_stream_t)stream, (const unsigned short*)x, y, n);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}
int msmc_tanh_f32_bwd(const float* y, const float* g, void* gx, long n, int dtype, msmc_stream stream) {
    if (!y || !g || !gx || n < 0) return MSMC_E_SHAPE;
    if (n == 0) return 0;
    const dim3 grid((unsigned)nm_grid(n));
    if (dtype == 0) MSMC_LAUNCH(tanh_f32_bwd_kernel<float>, grid, dim3(256), 0, (msmc_stream_t)stream, y, g, (float*)gx, n);
    else if (dtype == 1) MSMC_LAUNCH(tanh_f32_bwd_kernel<unsigned short>, grid, dim3(256), 0, (msmc_stream_t)stream, y, g, (unsigned short*)gx, n);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}
int msmc_sum_n(const void* a, const void* b, const void* c, const void* d, void* out, long n, int dtype, msmc_stream stream) {
    if (!a || !b || !out || n < 0 || (n & 3) || (d && !c)) return MSMC_E_SHAPE;
    if ((((size_t)a) | ((size_t)b) | ((size_t)c) | ((size_t)d) | ((size_t)out)) & (dtype == 0 ? 15 : 7)) return MSMC_E_SHAPE;
    if (n == 0) return 0;
    const dim3 grid((unsigned)nm_grid(n / 4));
    if (dtype == 0) MSMC_LAUNCH(sum_n_kernel<float>, grid, dim3(256), 0, (msmc_stream_t)stream, (const float*)a, (const float*)b, (const float*)c, (const float*)d, (float*)out, n / 4);
    else if (dtype == 1) MSMC_LAUNCH(sum_n_kernel<unsigned short>, grid, dim3(256), 0, (msmc_stream_t)stream, (const unsigned short*)a, (const unsigned short*)b, (const unsigned short*)c, (const unsigned short*)d, (unsigned short*)out, n / 4);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}
int msmc_dropout_add_fwd(const void* x, const void* res, void* y, long n, float p_drop, const long long* seed, long long salt, int dtype,
                         msmc_stream stream) {
    if (!x || !y || n < 0 || (n & 3) || p_drop < 0.f || p_drop >= 1.f) return MSMC_E_SHAPE;
    if ((((size_t)x) | ((size_t)res) | ((size_t)y)) & (dtype == 0 ? 15 : 7)) return MSMC_E_SHAPE;
    if (n == 0) return 0;
    const dim3 grid((unsigned)nm_grid(n / 4));
    if (dtype == 0) MSMC_LAUNCH((dropout_add_kernel<float, false>), grid, dim3(256), 0, (msmc_stream_t)stream, (const float*)x, (const float*)res, (float*)y, n / 4, p_drop, seed, salt);
    else if (dtype == 1) MSMC_LAUNCH((dropout_add_kernel<unsigned short, false>), grid, dim3(256), 0, (msmc_stream_t)stream, (const unsigned short*)x, (const unsigned short*)res, (unsigned short*)y, n / 4, p_drop, seed, salt);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}
int msmc_dropout_bwd(const void* g, void* gx, long n, float p_drop, const long long* seed, long long salt, int dtype, msmc_stream stream) {
    if (!g || !gx || n < 0 || (n & 3) || p_drop < 0.f || p_drop >= 1.f) return MSMC_E_SHAPE;
    if ((((size_t)g) | ((size_t)gx)) & (dtype == 0 ? 15 : 7)) return MSMC_E_SHAPE;
    if (n == 0) return 0;
    const dim3 grid((unsigned)nm_grid(n / 4));
    if (dtype == 0) MSMC_LAUNCH((dropout_add_kernel<float, true>), grid, dim3(256), 0, (msmc_stream_t)stream, (const float*)g, (const float*)nullptr, (float*)gx, n / 4, p_drop, seed, salt);
    else if (dtype == 1) MSMC_LAUNCH((dropout_add_kernel<unsigned short, true>), grid, dim3(256), 0, (msmc_stream_t)stream, (const unsigned short*)g, (const unsigned short*)nullptr, (unsigned short*)gx, n / 4, p_drop, seed, salt);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}
int msmc_row_mask(const void* lengths, int lengths_are_int64, void* keep, int B, int T, int dtype, msmc_stream stream) {
    if (!lengths || !keep || B <= 0 || T <= 0) return MSMC_E_SHAPE;
    const dim3 grid((unsigned)nm_grid((long)B * T));
    if (dtype == 0) MSMC_LAUNCH(row_mask_kernel<float>, grid, dim3(256), 0, (msmc_stream_t)stream, lengths, lengths_are_int64, (float*)keep, B, T);
    else if (dtype == 1) MSMC_LAUNCH(row_mask_kernel<unsigned short>, grid, dim3(256), 0, (msmc_stream_t)stream, lengths, lengths_are_int64, (unsigned short*)keep, B, T);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}
int msmc_tanh_bwd(const void* y, const void* g, void* gx, long n, int dtype, msmc_stream stream) {
    if (!y || !g || !gx || n < 0) return MSMC_E_SHAPE;
    if (n == 0) return 0;
    const dim3 grid((unsigned)nm_grid(n));
    if (dtype == 0) MSMC_LAUNCH(tanh_bwd_kernel<float>, grid, dim3(256), 0, (msmc_stream_t)stream, (const float*)y, (const float*)g, (float*)gx, n);
    else if (dtype == 1) MSMC_LAUNCH(tanh_bwd_kernel<unsigned short>, grid, dim3(256), 0, (msmc_stream_t)stream, (const unsigned short*)y, (const unsigned short*)g, (unsigned short*)gx, n);
    else return MSMC_E_SHAPE;
    return msmc_check_launch();
}

}  // extern "C"
