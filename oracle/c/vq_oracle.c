/* vq_oracle.c -- plain-C restatement of the multi-head nearest-codeword search.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): never linked into the product.
 *
 * Follows Quantize.forward / MultiHeadQuantize.forward of the reference
 * (msmctts/networks/vqgantts/modules.py:24-34,59-60,137-151):
 *   dist[n,k] = (|x_n|^2 - 2 x_n.e_k) + |e_k|^2   three separately rounded fp32 terms  (:26-30)
 *   ind[n]    = first index of the minimum                                               (:31)
 *   quant     = x + (e_ind - x) ; diff = (e_ind - x)^2, heads averaged                   (:33,59-60,147)
 * fp32 summation orders are fixed here to the orders csrc/vq.hip uses (dot product: k-ordered fmaf
 * chain from 0, |x|^2: four interleaved partial sums combined pairwise, |e|^2: sequential), so the
 * GPU kernel can be required to match this file BIT FOR BIT on arbitrary data, near-ties included;
 * tests/test_oracle_vs_golden.py separately pins this file to the reference's own indices.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* embed [H][d][K], x [N][D]; outputs may be NULL. Returns 0. */
int vq_oracle_search(const float* x, const float* embed, int64_t* ind, float* quant, float* diff,
                     float* best_dist, long N, int D, int H, int K) {
    const int d = D / H;
    float* enorm = (float*)malloc(sizeof(float) * (size_t)H * K);
    for (int h = 0; h < H; ++h)
        for (int k = 0; k < K; ++k) {
            float acc = 0.f;
            for (int j = 0; j < d; ++j) {
                float v = embed[((size_t)h * d + j) * K + k];
                float sq = v * v;
                acc = acc + sq;
            }
            enorm[h * K + k] = acc;
        }
    for (long n = 0; n < N; ++n) {
        for (int h = 0; h < H; ++h) {
            const float* xh = x + (size_t)n * D + (size_t)h * d;
            /* channel visiting order of the kernel: natural for the LDS-tile kernel (d % 16 != 0), and
             * (t, jj, g) -> 16t + 4g + jj for the register-resident kernel (d % 16 == 0); the four
             * partial sums of |x|^2 belong to the lane groups g */
            const int perm = (d % 16) == 0;
            float p[4] = {0.f, 0.f, 0.f, 0.f};
            if (perm) {
                for (int g = 0; g < 4; ++g)
                    for (int t = 0; t < d / 16; ++t)
                        for (int jj = 0; jj < 4; ++jj) {
                            float v = xh[16 * t + 4 * g + jj];
                            float sq = v * v;
                            p[g] = p[g] + sq;
                        }
            } else {
                for (int j = 0; j < d; ++j) {
                    float sq = xh[j] * xh[j];
                    p[j & 3] = p[j & 3] + sq;
                }
            }
            /* lane group g combines as (p_g + p_{g^1}) + (p_{g^2} + p_{g^3}); addition commutes */
            float xx = (p[0] + p[1]) + (p[2] + p[3]);
            float best = INFINITY;
            int bi = 0;
            for (int k = 0; k < K; ++k) {
                float dot = 0.f;
                if (perm) {
                    for (int t = 0; t < d / 16; ++t)
                        for (int jj = 0; jj < 4; ++jj)
                            for (int g = 0; g < 4; ++g) {
                                int j = 16 * t + 4 * g + jj;
                                dot = fmaf(embed[((size_t)h * d + j) * K + k], xh[j], dot);
                            }
                } else {
                    for (int j = 0; j < d; ++j) dot = fmaf(embed[((size_t)h * d + j) * K + k], xh[j], dot);
                }
                float t2 = 2.f * dot;
                float dist = (xx - t2) + enorm[h * K + k];
                if (dist < best) { best = dist; bi = k; }
            }
            if (ind) ind[(size_t)n * H + h] = bi;
            if (best_dist) best_dist[(size_t)n * H + h] = best;
            for (int j = 0; j < d; ++j) {
                float e = embed[((size_t)h * d + j) * K + bi] - xh[j];
                if (quant) quant[(size_t)n * D + (size_t)h * d + j] = xh[j] + e;
                if (diff) {
                    float sq = e * e;
                    float* o = diff + (size_t)n * d + j;
                    *o = (h == 0) ? sq : (*o + sq);
                }
            }
        }
        if (diff && H > 1)
            for (int j = 0; j < d; ++j) diff[(size_t)n * d + j] = diff[(size_t)n * d + j] / (float)H;
    }
    free(enorm);
    return 0;
}
