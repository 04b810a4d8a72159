/* ORACLE -- test infrastructure, NOT the product.
 *
 * Canonical-arithmetic restatement of the reference's brute-force kNN
 * (PointDA/model_utils.py:9-16; twin at PointSegDA/Models.py:8-15):
 *
 *     inner = -2 * x^T x ;  xx = sum_c x^2 ;  pd = -xx - inner - xx^T ;  idx = topk(pd, k)
 *
 * The reference evaluates this with a BLAS matmul whose fp32 summation order is unspecified, so
 * "bit-exact indices" needs a fixed arithmetic.  This file fixes it (hard
 * part 1) and the HIP kernel (mlsp_amd/csrc/knn.hip) reproduces it bit for bit:
 *
 *   dot(i,j) = fmaf chain over c = 0..C-1 starting from +0.0f   (== gfx950 f32 MFMA numerics)
 *   xx(j)    = the same chain on (x_j, x_j)
 *   t(i,j)   = fl(2*dot(i,j) - xx(j))        one rounding  (== (-xx) - (-2*dot), model_utils.py:12)
 *   pd(i,j)  = fl(t(i,j) - xx(i))            second rounding
 *   order    = pd descending, ties -> lower j first; the k best, nearest first.
 *
 * Pinned against the reference by tests/test_oracle_golden.py: equal index rows wherever the
 * reference's own rank gaps exceed its rounding noise, equal sets elsewhere (fixtures
 * tests/golden/knn_*.npz).
 *
 * Layout: x is POINT-major [B][N][C] (the reference transposes to this at model_utils.py:35).
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static inline float chain_dot(const float* a, const float* b, int C) {
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) acc = fmaf(a[c], b[c], acc);
    return acc;
}

/* returns 0 on success, -1 on bad arguments */
int oracle_knn_f32(const float* x, int B, int N, int C, int k, int32_t* idx, float* pd_out /* nullable [B][N][k] */) {
    if (!x || !idx || B < 0 || N <= 0 || C <= 0 || k <= 0 || k > N) return -1;
#pragma omp parallel
    {
        float* xx = (float*)malloc(sizeof(float) * (size_t)N);
        float* pd = (float*)malloc(sizeof(float) * (size_t)N);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            const float* xb = x + (size_t)b * N * C;
            for (int j = 0; j < N; ++j) xx[j] = chain_dot(xb + (size_t)j * C, xb + (size_t)j * C, C);
            for (int i = 0; i < N; ++i) {
                for (int j = 0; j < N; ++j) {
                    float dot = chain_dot(xb + (size_t)i * C, xb + (size_t)j * C, C);
                    float t = fmaf(2.0f, dot, -xx[j]);
                    pd[j] = t - xx[i];
                }
                int32_t* out = idx + ((size_t)b * N + i) * k;
                float* po = pd_out ? pd_out + ((size_t)b * N + i) * k : 0;
                /* selection of the k best under (pd desc, j asc) */
                for (int s = 0; s < k; ++s) {
                    int best = -1;
                    float bv = 0.0f;
                    for (int j = 0; j < N; ++j) {
                        if (isnan(pd[j])) continue;
                        if (best < 0 || pd[j] > bv) { best = j; bv = pd[j]; }
                    }
                    if (best < 0) best = 0; /* all-NaN row: undefined in the reference as well */
                    out[s] = best;
                    if (po) po[s] = bv;
                    pd[best] = NAN; /* consumed */
                }
            }
        }
        free(xx);
        free(pd);
    }
    return 0;
}
