// Skinny contractions of the per-cloud layers (rows = batch <= 32): T-Net FC tail, classifier, the x5 "gbias" halves
// of the head layers.  With <= 32 rows the general 128x128 tile kernel runs 1/4-filled tiles, needs split-K plus a slab
// reduce, and BatchNorm1d over 32 rows costs three more launches.  Here a workgroup owns ALL rows of a 32-column slice, so
//   forward   Y = X W^T (+bias), batch statistics, scale/shift, activation, dropout, running statistics: ONE kernel
//   backward  BN/activation backward over the batch: one kernel;  dX = dY W and dW = dY^T X: one kernel each
// and no split-K slabs are written at all.  Same arithmetic as the general path (fp32 MFMA 32x32x2, fp32 statistics over
// <= 32 rows); only the summation order of the K loop differs.
#include "common.h"

// k assignment shared by the A and B side of every kernel here: within a chunk of 8 consecutive k, half-wave h takes
// k = kb + 4h + e for MFMA step e = 0..3 (any bijection works as long as both operands use the same one).

// float4 at row-major src[row][k..k+3]; rows >= nrows and k >= K read as zero
__device__ __forceinline__ f32x4 sk_load4(const float* __restrict__ src, int ld, int row, int nrows, int k, int K, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < nrows) {
        const float* g = src + (size_t)row * ld + k;
        if (vec && k + 3 < K) v = *(const f32x4*)g;
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (k + e < K) v[e] = g[e];
        }
    }
    return v;
}

// ---- forward: C[M<=32][N] = A[M][K] * W[N][K]^T (+bias) (+gbias row 0.. only via rows_per_group) and the optional fused BN tail.
// grid = ceil(N/32) workgroups of SK_WAVES waves; wave w covers its slice of K; partial tiles are added in wave order.
struct SkinnyFwdArgs {
    const float* X; const float* W; const float* bias;
    int ldx, ldw, M, N, K;
    float* Y; int ldy;                 // linear output (pre-BN); may be null when only Z is wanted and there is no BN
    // fused BN + activation + dropout tail (gamma == null: plain linear, Y gets the result)
    const float* gamma; const float* beta; float* run_mean; float* run_var;
    float momentum, eps; int training, act; float slope; uint32_t thresh; float inv_keep; uint64_t seed;
    float* Z; float* bn_save;          // Z [M][N]; bn_save = scale | shift | mean | invstd, N floats each
};

#ifndef SK_WAVES
#define SK_WAVES 8           // waves per workgroup (2 per SIMD -> 256 VGPRs each): the K range is cut 8 ways
#endif
#ifndef SK_BATCH
#define SK_BATCH 8           // chunks (of 8 k) whose loads are all in flight before the first MFMA of the batch
#endif

// sum of the SK_WAVES partial tiles in wave order; result in wave 0 (other waves return false)
__device__ __forceinline__ bool sk_reduce_tiles(f32x16& acc, float (*red)[16][64], int wave, int lane) {
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return false;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r];
#pragma unroll
        for (int w = 0; w < SK_WAVES - 1; ++w) v += red[w][r][lane];
        acc[r] = v;
    }
    return true;
}

__global__ __launch_bounds__(64 * SK_WAVES) void skinny_fwd_kernel(SkinnyFwdArgs p) {
    __shared__ float red[SK_WAVES - 1][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const bool vec = (p.ldx % 4 == 0) && (p.ldw % 4 == 0) && ((((uintptr_t)p.X | (uintptr_t)p.W) & 15) == 0);
    const int chunks = (p.K + 7) / 8, cper = (chunks + SK_WAVES - 1) / SK_WAVES;
    const int c0 = wave * cper, c1 = min(chunks, c0 + cper);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int cb = c0; cb < c1; cb += SK_BATCH) {
        f32x4 a[SK_BATCH], b[SK_BATCH];
#pragma unroll
        for (int u = 0; u < SK_BATCH; ++u) {
            const int k = (cb + u) * 8 + 4 * h;
            const bool on = cb + u < c1;
            a[u] = sk_load4(p.X, p.ldx, l31, on ? p.M : 0, k, p.K, vec);
            b[u] = sk_load4(p.W, p.ldw, n0 + l31, on ? p.N : 0, k, p.K, vec);
        }
#pragma unroll
        for (int u = 0; u < SK_BATCH; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], b[u][e], acc, 0, 0, 0);
    }
    if (!sk_reduce_tiles(acc, red, wave, lane)) return;
    // wave 0: the whole 32x32 tile.  lane = (col l31, row half h); acc[r] is row (r&3) + 8*(r>>2) + 4h
    const int col = n0 + l31;
    const bool cok = col < p.N;
    const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
    float y[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = acc[r] + bv;
    if (!p.gamma) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (cok && row < p.M) p.Y[(size_t)row * p.ldy + col] = y[r];
        }
        return;
    }
    float sc, sh;
    if (p.training) {
        // batch statistics over the M rows of this column (two passes in registers)
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += ((r & 3) + 8 * (r >> 2) + 4 * h < p.M) ? y[r] : 0.f;
        s += __shfl_xor(s, 32, 64);
        const float mean = s / (float)p.M;
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = y[r] - mean;
            q += ((r & 3) + 8 * (r >> 2) + 4 * h < p.M) ? d * d : 0.f;
        }
        q += __shfl_xor(q, 32, 64);
        const float var = q / (float)p.M;
        const float invstd = 1.0f / sqrtf(var + p.eps);
        sc = cok ? p.gamma[col] * invstd : 0.f;
        sh = cok ? p.beta[col] - mean * sc : 0.f;
        if (cok && h == 0) {
            p.bn_save[col] = sc; p.bn_save[p.N + col] = sh; p.bn_save[2 * p.N + col] = mean; p.bn_save[3 * p.N + col] = invstd;
            if (p.run_mean) {
                const float unb = p.M > 1 ? var * (float)p.M / (float)(p.M - 1) : var;
                p.run_mean[col] = (1.f - p.momentum) * p.run_mean[col] + p.momentum * mean;
                p.run_var[col] = (1.f - p.momentum) * p.run_var[col] + p.momentum * unb;
            }
        }
    } else {
        const float rm = cok ? p.run_mean[col] : 0.f, rv = cok ? p.run_var[col] : 1.f;
        const float invstd = 1.0f / sqrtf(rv + p.eps);
        sc = cok ? p.gamma[col] * invstd : 0.f;
        sh = cok ? p.beta[col] - rm * sc : 0.f;
        if (cok && h == 0) { p.bn_save[col] = sc; p.bn_save[p.N + col] = sh; p.bn_save[2 * p.N + col] = rm; p.bn_save[3 * p.N + col] = invstd; }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (cok && row < p.M) {
            const size_t i = (size_t)row * p.N + col;
            p.Y[(size_t)row * p.ldy + col] = y[r];
            float a = lrelu_or_relu(fmaf(y[r], sc, sh), p.act, p.slope);
            if (p.thresh) a = dropout_keep(p.seed, i, p.thresh) ? a * p.inv_keep : 0.f;
            p.Z[i] = a;
        }
    }
}

// ---- BN + activation backward over a batch of <= 32 rows: dY = scale * (d - mean(d) - yhat * mean(d*yhat)), dgamma, dbeta.
// A workgroup owns 64 columns; its 4 waves take 8 rows each (registers), column sums meet in LDS in wave order.
__global__ __launch_bounds__(256) void skinny_bn_bwd_kernel(const float* __restrict__ dZ, const float* __restrict__ Y, float* __restrict__ dY,
                                                            int M, int C, const float* __restrict__ bn_save, int training, int act,
                                                            float slope, uint32_t thresh, float inv_keep, uint64_t seed,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ zero_vec) {
    __shared__ float ss[4][64], sq[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool cok = c < C;
    const float sc = cok ? bn_save[c] : 0.f, sh = cok ? bn_save[C + c] : 0.f, mu = cok ? bn_save[2 * C + c] : 0.f,
                is = cok ? bn_save[3 * C + c] : 0.f;
    float d[8], yh[8], yv[8], dz[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = wave * 8 + u;
        const bool on = cok && r < M;
        const size_t i = (size_t)r * C + c;
        yv[u] = on ? Y[i] : 0.f;
        dz[u] = on ? dZ[i] : 0.f;
    }
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = wave * 8 + u;
        const size_t i = (size_t)r * C + c;
        float g = dz[u];
        if (thresh) g = dropout_keep(seed, i, thresh) ? g * inv_keep : 0.f;
        if (act) {
            const float a = fmaf(yv[u], sc, sh);
            if (!(a > 0.f)) g *= (act == 1 ? 0.f : slope);
        }
        if (!(cok && r < M)) g = 0.f;
        d[u] = g; yh[u] = (yv[u] - mu) * is;
        s += g; q = fmaf(g, yh[u], q);
    }
    ss[wave][lane] = s; sq[wave][lane] = q;
    __syncthreads();
    s = ((ss[0][lane] + ss[1][lane]) + ss[2][lane]) + ss[3][lane];
    q = ((sq[0][lane] + sq[1][lane]) + sq[2][lane]) + sq[3][lane];
    if (wave == 0 && cok) { dgamma[c] = q; dbeta[c] = s; if (zero_vec) zero_vec[c] = 0.f; }      // (zero_vec: the bias gradient in front of a batch-statistics BatchNorm)
    const float k1 = training ? s / (float)M : 0.f, k2 = training ? q / (float)M : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = wave * 8 + u;
        if (cok && r < M) dY[(size_t)r * C + c] = sc * (d[u] - k1 - yh[u] * k2);
    }
}

// ---- dgrad: C[M<=32][N] = A[M][K] * B[K][N]   (A row-major, B k-major).  Same 4-wave K split as the forward.
__device__ __forceinline__ void sk_nn_body(int blk, float (*red)[16][64], const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                           float* __restrict__ Cm, int ldc, int M, int N, int K, bool accumulate = false) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int n0 = blk * 32, col = n0 + l31;
    const bool vec = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
    const int chunks = (K + 7) / 8, cper = (chunks + SK_WAVES - 1) / SK_WAVES;
    const int c0 = wave * cper, c1 = min(chunks, c0 + cper);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int cb = c0; cb < c1; cb += SK_BATCH) {
        f32x4 a[SK_BATCH];
        float b[SK_BATCH][4];
#pragma unroll
        for (int u = 0; u < SK_BATCH; ++u) {
            const int k = (cb + u) * 8 + 4 * h;
            const bool on = cb + u < c1;
            a[u] = sk_load4(A, lda, l31, on ? M : 0, k, K, vec);
#pragma unroll
            for (int e = 0; e < 4; ++e) b[u][e] = (on && col < N && k + e < K) ? B[(size_t)(k + e) * ldb + col] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SK_BATCH; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], b[u][e], acc, 0, 0, 0);
    }
    if (!sk_reduce_tiles(acc, red, wave, lane) || col >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M) Cm[(size_t)row * ldc + col] = accumulate ? Cm[(size_t)row * ldc + col] + acc[r] : acc[r];      // (beta = 1: the input gradient is summed into a shared buffer)
    }
}

__global__ __launch_bounds__(64 * SK_WAVES) void skinny_nn_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                                  float* __restrict__ Cm, int ldc, int M, int N, int K) {
    __shared__ float red[SK_WAVES - 1][16][64];
    sk_nn_body(blockIdx.x, red, A, lda, B, ldb, Cm, ldc, M, N, K);
}

// ---- wgrad: C[Mo][No] = A^T B with A [K<=32][Mo], B [K][No] (both k-major).  One wave per 32x32 output tile.
__device__ __forceinline__ void sk_tn_body(int tile, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                           float* __restrict__ Cm, int ldc, int Mo, int No, int K) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int ntn = (No + 31) / 32;
    if (tile >= ((Mo + 31) / 32) * ntn) return;
    const int m0 = (tile / ntn) * 32, n0 = (tile % ntn) * 32;
    const int arow = m0 + l31, bcol = n0 + l31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a[16], b[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = 2 * s + h;
        a[s] = (k < K && arow < Mo) ? A[(size_t)k * lda + arow] : 0.f;
        b[s] = (k < K && bcol < No) ? B[(size_t)k * ldb + bcol] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    if (bcol >= No) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < Mo) Cm[(size_t)row * ldc + bcol] = acc[r];
    }
}
__global__ __launch_bounds__(256) void skinny_tn_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                        float* __restrict__ Cm, int ldc, int Mo, int No, int K) {
    sk_tn_body(blockIdx.x * 4 + (threadIdx.x >> 6), A, lda, B, ldb, Cm, ldc, Mo, No, K);
}
// ---- both gradients of a per-cloud Linear layer (rows = batch <= 32) in ONE launch: workgroups [0, nb_nn) run the dgrad dX = G W
// (skinny_nn_kernel's body), the rest the weight gradient dW = G^T X (skinny_tn_kernel's body, eight tiles per workgroup).  Same arithmetic,
// same summation order as the two launches it replaces.
__global__ __launch_bounds__(64 * SK_WAVES) void skinny_bwd_pair_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ W, int ldw,
                                                                        const float* __restrict__ X, int ldx, float* __restrict__ dX, int lddx,
                                                                        float* __restrict__ dW, int M, int Cin, int Cout, int nb_nn, int dx_accumulate) {
    __shared__ float red[SK_WAVES - 1][16][64];
    if ((int)blockIdx.x < nb_nn) sk_nn_body(blockIdx.x, red, G, ldg, W, ldw, dX, lddx, M, Cin, Cout, dx_accumulate != 0);
    else sk_tn_body(((int)blockIdx.x - nb_nn) * SK_WAVES + (threadIdx.x >> 6), G, ldg, X, ldx, dW, Cin, Cout, Cin, M);
}
int launch_skinny_bwd_pair(hipStream_t st, const float* G, int ldg, const float* W, int ldw, const float* X, int ldx, float* dX, int lddx,
                           float* dW, int M, int Cin, int Cout, int dx_accumulate) {
    if (M > 32 || !G || !W || !X || !dX || !dW) return MLSP_ERR_UNSUPPORTED;
    const int nb_nn = (Cin + 31) / 32, tiles = ((Cout + 31) / 32) * ((Cin + 31) / 32);
    hipLaunchKernelGGL(skinny_bwd_pair_kernel, dim3(nb_nn + (tiles + SK_WAVES - 1) / SK_WAVES), dim3(64 * SK_WAVES), 0, st, G, ldg, W, ldw, X, ldx,
                       dX, lddx, dW, M, Cin, Cout, nb_nn, dx_accumulate);
    return mlsp_launch_status();
}

// ---- host side ------------------------------------------------------------------------------------------------------
static inline uint32_t sk_drop_thresh(float p) { return dropout_thresh8(p); }

// plain skinny GEMMs behind launch_gemm (no split-K, no slab).  Returns MLSP_ERR_UNSUPPORTED when the shape is not skinny.
int launch_skinny_gemm(hipStream_t st, bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                       float* C, int ldc, const float* bias) {
    if (!ta && tb && M <= 32) {
        SkinnyFwdArgs p = {};
        p.X = A; p.W = B; p.bias = bias; p.ldx = lda; p.ldw = ldb; p.M = M; p.N = N; p.K = K; p.Y = C; p.ldy = ldc;
        hipLaunchKernelGGL(skinny_fwd_kernel, dim3((N + 31) / 32), dim3(64 * SK_WAVES), 0, st, p);
        return mlsp_launch_status();
    }
    if (!ta && !tb && M <= 32 && !bias) {
        hipLaunchKernelGGL(skinny_nn_kernel, dim3((N + 31) / 32), dim3(64 * SK_WAVES), 0, st, A, lda, B, ldb, C, ldc, M, N, K);
        return mlsp_launch_status();
    }
    if (ta && !tb && K <= 32 && !bias) {
        const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
        hipLaunchKernelGGL(skinny_tn_kernel, dim3((tiles + 3) / 4), dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K);
        return mlsp_launch_status();
    }
    return MLSP_ERR_UNSUPPORTED;
}

// fused Linear + BatchNorm1d + activation + dropout over <= 32 rows
int launch_skinny_linear_bn_act(hipStream_t st, const float* X, int ldx, int M, int Cin, const float* W, int ldw, int Cout,
                                const float* bias, const float* gamma, const float* beta, float* run_mean, float* run_var,
                                float momentum, float eps, int training, int act, float slope, float p_drop, uint64_t seed, float* Y,
                                float* Z, float* bn_save) {
    if (M > 32 || !gamma || !beta || !Y || !Z || !bn_save) return MLSP_ERR_ARG;
    if (!training && (!run_mean || !run_var)) return MLSP_ERR_ARG;
    SkinnyFwdArgs p = {};
    p.X = X; p.W = W; p.bias = bias; p.ldx = ldx; p.ldw = ldw; p.M = M; p.N = Cout; p.K = Cin; p.Y = Y; p.ldy = Cout;
    p.gamma = gamma; p.beta = beta; p.run_mean = run_mean; p.run_var = run_var; p.momentum = momentum; p.eps = eps;
    p.training = training; p.act = act; p.slope = slope;
    const float pd = training ? p_drop : 0.f;
    p.thresh = sk_drop_thresh(pd); p.inv_keep = dropout_inv_keep8(pd); p.seed = seed;
    p.Z = Z; p.bn_save = bn_save;
    hipLaunchKernelGGL(skinny_fwd_kernel, dim3((Cout + 31) / 32), dim3(64 * SK_WAVES), 0, st, p);
    return mlsp_launch_status();
}

int launch_skinny_bn_bwd_z(hipStream_t st, const float* dZ, const float* Y, float* dY, int M, int C, const float* bn_save, int training,
                           int act, float slope, float p_drop, uint64_t seed, float* dgamma, float* dbeta, float* zero_vec) {
    if (M > 32) return MLSP_ERR_ARG;
    const float pd = training ? p_drop : 0.f;
    hipLaunchKernelGGL(skinny_bn_bwd_kernel, dim3((C + 63) / 64), dim3(256), 0, st, dZ, Y, dY, M, C, bn_save, training, act, slope,
                       sk_drop_thresh(pd), dropout_inv_keep8(pd), seed, dgamma, dbeta, zero_vec);
    return mlsp_launch_status();
}
int launch_skinny_bn_bwd(hipStream_t st, const float* dZ, const float* Y, float* dY, int M, int C, const float* bn_save, int training,
                         int act, float slope, float p_drop, uint64_t seed, float* dgamma, float* dbeta) {
    return launch_skinny_bn_bwd_z(st, dZ, Y, dY, M, C, bn_save, training, act, slope, p_drop, seed, dgamma, dbeta, nullptr);
}


// ---- composition of two linear maps (PointSegDA's conv pairs without an activation in between: include/mlsp_hip.h mlsp_compose_linear_*) ----
// forward: workgroup = output row o; thread c < Ci: W[o][c] = sum_m Wb[o][m] Wa[m][c]; thread Ci: b[o] = sum_m Wb[o][m] ba[m] + bb[o]
__global__ __launch_bounds__(256) void compose_fwd_kernel(const float* __restrict__ Wa, const float* __restrict__ ba, const float* __restrict__ Wb,
                                                          const float* __restrict__ bb, int Cm, int Ci, float* __restrict__ W, float* __restrict__ b) {
    const int o = blockIdx.x;
    for (int c = threadIdx.x; c <= Ci; c += blockDim.x) {
        float acc = 0.f;
        if (c < Ci) {
#pragma unroll 16
            for (int m = 0; m < Cm; ++m) acc = fmaf(Wb[(size_t)o * Cm + m], Wa[(size_t)m * Ci + c], acc);      // (sixteen loads in flight: the chain of Cm load latencies was the launch's 14 us)
            W[(size_t)o * Ci + c] = acc;
        } else {
#pragma unroll 16
            for (int m = 0; m < Cm; ++m) acc = fmaf(Wb[(size_t)o * Cm + m], ba[m], acc);
            b[o] = acc + bb[o];
        }
    }
}
// backward: workgroups [0, Cm): row m of dWa | dba (sums over o ascending); workgroups [Cm, Cm + Co): row o of dWb (per m a wave's tree sum
// over c, then the bias term)
__global__ __launch_bounds__(256) void compose_bwd_kernel(const float* __restrict__ dW, const float* __restrict__ db, const float* __restrict__ Wa,
                                                          const float* __restrict__ ba, const float* __restrict__ Wb, int Cm, int Ci, int Co,
                                                          float* __restrict__ dWa, float* __restrict__ dba, float* __restrict__ dWb) {
    const int blk = blockIdx.x;
    if (blk < Cm) {
        const int m = blk;
        for (int c = threadIdx.x; c <= Ci; c += blockDim.x) {
            float acc = 0.f;
            if (c < Ci) {
#pragma unroll 16
                for (int o = 0; o < Co; ++o) acc = fmaf(Wb[(size_t)o * Cm + m], dW[(size_t)o * Ci + c], acc);
                dWa[(size_t)m * Ci + c] = acc;
            } else {
#pragma unroll 16
                for (int o = 0; o < Co; ++o) acc = fmaf(Wb[(size_t)o * Cm + m], db[o], acc);
                dba[m] = acc;
            }
        }
    } else {
        // one wave per m, lanes over c (coalesced rows of Wa; a thread per m walked Ci strided loads one after the other: 23 us per launch
        // at Ci = 128), fixed-order tree sum
        const int o = blk - Cm, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
        for (int m = threadIdx.x >> 6; m < Cm; m += nw) {
            float acc = 0.f;
            for (int c = lane; c < Ci; c += 64) acc = fmaf(dW[(size_t)o * Ci + c], Wa[(size_t)m * Ci + c], acc);
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) acc += __shfl_xor(acc, sh, 64);
            if (lane == 0) dWb[(size_t)o * Cm + m] = fmaf(db[o], ba[m], acc);
        }
    }
}
int launch_compose_fwd(hipStream_t st, const float* Wa, const float* ba, const float* Wb, const float* bb, int Cm, int Ci, int Co, float* W, float* b) {
    hipLaunchKernelGGL(compose_fwd_kernel, dim3(Co), dim3(Ci + 1 <= 64 ? 64 : Ci + 1 <= 128 ? 128 : 256), 0, st, Wa, ba, Wb, bb, Cm, Ci, W, b);
    return mlsp_launch_status();
}
int launch_compose_bwd(hipStream_t st, const float* dW, const float* db, const float* Wa, const float* ba, const float* Wb, int Cm, int Ci, int Co,
                       float* dWa, float* dba, float* dWb) {
    hipLaunchKernelGGL(compose_bwd_kernel, dim3(Cm + Co), dim3(256), 0, st, dW, db, Wa, ba, Wb, Cm, Ci, Co, dWa, dba, dWb);
    return mlsp_launch_status();
}
