// BatchNorm (training & eval) + activation + dropout, and the max-reductions of the hot path.
// Replaces the BatchNorm2d/BatchNorm1d + ReLU/LeakyReLU + Dropout modules of conv_2d / fc_layer
// (PointDA/model_utils.py:45-87), bn5 (PointDA/Models.py:132), the head BN stacks
// (Models.py:192-197,226-231,272-279), `.max(dim=-1)` over k (Models.py:117-129, model_utils.py:114),
// `torch.max(x, dim=2)` (model_utils.py:117) and adaptive_max_pool1d (Models.py:136).
//
// All activations are POINT-major row matrices [rows][C]; BN statistics are per column.
// These kernels are HBM-bound streaming passes: lanes run along the contiguous channel axis.
// Column sums are accumulated in fp64 (per-thread) so that var = E[y^2]-E[y]^2 does not lose
// precision at 655,360 rows.
#include "common.h"
#include <math.h>

#define STAT_ROWS 512     // rows per partial-sum block

// ---------------------------------------------------------------------------------------------
// column sums of y and y^2 -> partials [nparts][2][C] (fp64)
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ Y, int M, int C, int ld,
                                                       double* __restrict__ part) {
    __shared__ double sh[2][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * STAT_ROWS, r1 = min(M, r0 + STAT_ROWS);
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int r = r0 + w; r < r1; r += 4) {
            float v = Y[(size_t)r * ld + c];
            s += v; q += (double)v * v;
        }
    sh[0][w][lane] = s; sh[1][w][lane] = q;
    __syncthreads();
    if (w == 0 && c < C) {
        double ts = sh[0][0][lane] + sh[0][1][lane] + sh[0][2][lane] + sh[0][3][lane];
        double tq = sh[1][0][lane] + sh[1][1][lane] + sh[1][2][lane] + sh[1][3][lane];
        part[((size_t)blockIdx.y * 2 + 0) * C + c] = ts;
        part[((size_t)blockIdx.y * 2 + 1) * C + c] = tq;
    }
}

// Column c of the partial planes [nparts][2][C] summed by a 256-thread workgroup: every thread strides over the partial rows with four
// loads of each plane in flight (one wave per channel left 8+ dependent round trips on the critical path of every BN layer), then a
// fixed-order butterfly inside the wave and a fixed-order sum of the four waves -> reproducible.  Result valid in thread 0.
#define FIN_THREADS 256
__device__ __forceinline__ void fin_part_sums(const double* __restrict__ part, int nparts, int C, int c, double& s, double& q) {
    __shared__ double fred[2][FIN_THREADS / 64];
    const int tid = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    int i = tid;
    for (; i + 3 * FIN_THREADS < nparts; i += 4 * FIN_THREADS) {
        const double a0 = part[((size_t)i * 2) * C + c], b0 = part[((size_t)i * 2 + 1) * C + c];
        const double a1 = part[((size_t)(i + FIN_THREADS) * 2) * C + c], b1 = part[((size_t)(i + FIN_THREADS) * 2 + 1) * C + c];
        const double a2 = part[((size_t)(i + 2 * FIN_THREADS) * 2) * C + c], b2 = part[((size_t)(i + 2 * FIN_THREADS) * 2 + 1) * C + c];
        const double a3 = part[((size_t)(i + 3 * FIN_THREADS) * 2) * C + c], b3 = part[((size_t)(i + 3 * FIN_THREADS) * 2 + 1) * C + c];
        s0 += a0; s1 += a1; s2 += a2; s3 += a3; q0 += b0; q1 += b1; q2 += b2; q3 += b3;
    }
    for (; i < nparts; i += FIN_THREADS) { s0 += part[((size_t)i * 2) * C + c]; q0 += part[((size_t)i * 2 + 1) * C + c]; }
    s = (s0 + s1) + (s2 + s3); q = (q0 + q1) + (q2 + q3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if ((tid & 63) == 0) { fred[0][tid >> 6] = s; fred[1][tid >> 6] = q; }
    __syncthreads();
    s = (fred[0][0] + fred[0][1]) + (fred[0][2] + fred[0][3]);
    q = (fred[1][0] + fred[1][1]) + (fred[1][2] + fred[1][3]);
}

// partials -> mean / biased var -> scale, shift, saved stats, running-stat update (torch semantics:
// running_var uses the unbiased estimate, momentum 0.1; model_utils.py:56-58 defaults)
__global__ void bn_finalize_kernel(const double* __restrict__ part, int nparts, double count, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ run_mean, float* __restrict__ run_var, float momentum, float eps,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean,
                                   float* __restrict__ save_invstd, float* __restrict__ gsum, int ppg,
                                   float* __restrict__ bound_out = nullptr, float sqrt_count = 0.f) {
    // bound_out (nullable, [C]): |gamma| sqrt(count) + |beta| -- a bound of every value act(BN(y)) of the channel (|yhat| <= sqrt(count) for
    // batch statistics over exactly these `count` values; |act(z)| <= |z|): what the two-piece f16 products of the layers that READ this
    // layer's output need (gemm.hip GemmArgs a_amax), at no pass over the data
    const int c = blockIdx.x;
    double s, q;
    fin_part_sums(part, nparts, C, c, s, q);
    if (bound_out && threadIdx.x == 0) bound_out[c] = fabsf(gamma[c]) * sqrt_count + fabsf(beta[c]);
    // gsum (nullable, [nparts / ppg][C]): the column sums of every cloud (ppg row panels each) -- kept for the backward of a layer with a
    // per-cloud bias whose output gradient is never formed (launch_bn_dy_gbias)
    if (gsum)
        for (int g = threadIdx.x; g < nparts / ppg; g += blockDim.x) {
            double t = 0.0;
            for (int p = 0; p < ppg; ++p) t += part[((size_t)(g * ppg + p) * 2) * C + c];
            gsum[(size_t)g * C + c] = (float)t;
        }
    if (threadIdx.x != 0) return;
    double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    if (run_mean) {
        double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
}

// eval mode: scale/shift from the running statistics
__global__ void bn_eval_prepare_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ run_mean, const float* __restrict__ run_var, float eps,
                                       float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean,
                                       float* __restrict__ save_invstd) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float invstd = 1.0f / sqrtf(run_var[c] + eps);
    float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - run_mean[c] * sc;
    save_mean[c] = run_mean[c];
    save_invstd[c] = invstd;
}

// Z = dropout(act(Y*scale + shift)).  thresh = round(256 p), inv_keep = 256 / (256 - thresh)  (common.h)
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ Y, float* __restrict__ Z, size_t total,
                                                         int C, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int act, float slope,
                                                         uint32_t thresh, float inv_keep, uint64_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        float a = lrelu_or_relu(fmaf(Y[i], scale[c], shift[c]), act, slope);
        if (thresh) a = dropout_keep(seed, i, thresh) ? a * inv_keep : 0.f;
        Z[i] = a;
    }
}

// gradient wrt the BN output:  dz' = dZ * dropmask * act'(a),  a = Y*scale+shift
__device__ __forceinline__ float dz_prime(float dz, float y, float sc, float sh, int act, float slope, uint32_t thresh,
                                          float inv_keep, uint64_t seed, size_t i) {
    if (thresh) dz = dropout_keep(seed, i, thresh) ? dz * inv_keep : 0.f;
    if (act) {
        float a = fmaf(y, sc, sh);
        if (!(a > 0.f)) dz *= (act == 1 ? 0.f : slope);
    }
    return dz;
}

// the same for element e of an aligned quad whose hash hq = dropout_hash4(seed, i >> 2) the caller computed once
__device__ __forceinline__ float dz_prime_q(float dz, float y, float sc, float sh, int act, float slope, uint32_t thresh,
                                            float inv_keep, uint32_t hq, int e) {
    if (thresh) dz = ((hq >> (8 * e)) & 255u) >= thresh ? dz * inv_keep : 0.f;
    if (act) {
        float a = fmaf(y, sc, sh);
        if (!(a > 0.f)) dz *= (act == 1 ? 0.f : slope);
    }
    return dz;
}

// partial column sums of dz' and dz'*yhat  (fp64 partials [nparts][2][C])
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const float* __restrict__ dZ, const float* __restrict__ Y,
                                                                int M, int C, const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int act, float slope,
                                                                uint32_t thresh, float inv_keep, uint64_t seed,
                                                                double* __restrict__ part) {
    __shared__ double sh[2][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * STAT_ROWS, r1 = min(M, r0 + STAT_ROWS);
    double s = 0.0, q = 0.0;
    if (c < C) {
        float sc = scale[c], shf = shift[c], mu = mean[c], is = invstd[c];
        for (int r = r0 + w; r < r1; r += 4) {
            size_t i = (size_t)r * C + c;
            float y = Y[i];
            float d = dz_prime(dZ[i], y, sc, shf, act, slope, thresh, inv_keep, seed, i);
            s += d; q += (double)d * ((y - mu) * is);
        }
    }
    sh[0][w][lane] = s; sh[1][w][lane] = q;
    __syncthreads();
    if (w == 0 && c < C) {
        part[((size_t)blockIdx.y * 2 + 0) * C + c] = sh[0][0][lane] + sh[0][1][lane] + sh[0][2][lane] + sh[0][3][lane];
        part[((size_t)blockIdx.y * 2 + 1) * C + c] = sh[1][0][lane] + sh[1][1][lane] + sh[1][2][lane] + sh[1][3][lane];
    }
}

// dgamma = sum dz'*yhat, dbeta = sum dz'; coefficient vectors for the apply pass
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ part, int nparts, double count, int C,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ mean_dz, float* __restrict__ mean_dzy, float* __restrict__ zero_vec = nullptr) {
    const int c = blockIdx.x;
    double s, q;
    fin_part_sums(part, nparts, C, c, s, q);
    if (threadIdx.x != 0) return;
    if (zero_vec) zero_vec[c] = 0.f;          // the gradient of a bias in front of a batch-statistics BatchNorm: analytically zero (no memset launch)
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
    mean_dz[c] = (float)(s / count);
    mean_dzy[c] = (float)(q / count);
}

// dY = scale * (dz' - mean_dz - yhat*mean_dzy)   (training)   |   dY = scale*dz'   (eval: mean_dz == null)
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const float* __restrict__ dZ, const float* __restrict__ Y,
                                                               float* __restrict__ dY, size_t total, int C,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ mean_dz,
                                                               const float* __restrict__ mean_dzy, int act, float slope,
                                                               uint32_t thresh, float inv_keep, uint64_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        float y = Y[i];
        float d = dz_prime(dZ[i], y, scale[c], shift[c], act, slope, thresh, inv_keep, seed, i);
        if (mean_dz) d = d - mean_dz[c] - (y - mean[c]) * invstd[c] * mean_dzy[c];
        dY[i] = scale[c] * d;
    }
}


// ---------------------------------------------------------------------------------------------
// Vectorised forms (C % 4 == 0, C <= 1024, 16-byte aligned, ld == C): a thread owns 4 consecutive columns
// and walks rows; C/4 threads cover a row with one 16-byte load each, 256/(C/4) rows per pass.
#define VROWS 64      // rows per partial block of the vectorised reductions (512 blocks at 32,768 rows)

__device__ __forceinline__ void vec_block_reduce_write(double (&s)[4], double (&q)[4], int C, int tpr, int tid, double* shd,
                                                       double* __restrict__ part, int pblock) {
    // shd: [256][8] doubles.  thread (rg = tid / tpr, cg = tid % tpr) -> sum over rg
#pragma unroll
    for (int e = 0; e < 4; ++e) { shd[tid * 8 + e] = s[e]; shd[tid * 8 + 4 + e] = q[e]; }
    __syncthreads();
    if (tid < tpr) {
        const int nrg = 256 / tpr;
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < nrg; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += shd[(g * tpr + tid) * 8 + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            part[((size_t)pblock * 2 + 0) * C + tid * 4 + e] = a[e];
            part[((size_t)pblock * 2 + 1) * C + tid * 4 + e] = a[4 + e];
        }
    }
}

// 4 consecutive elements of an activation matrix held as fp32 or bf16 (configs[4] activation storage) <-> four floats
typedef __bf16 bn_bf16x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 ld4<__bf16>(const __bf16* p) {
    const bn_bf16x4 v = *(const bn_bf16x4*)p;
    return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void st4<__bf16>(__bf16* p, f32x4 v) {
    bn_bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
    *(bn_bf16x4*)p = o;
}

template <typename TY>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_vec_kernel(const TY* __restrict__ dZ, const TY* __restrict__ Y,
                                                                    int M, int C, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd, int act, float slope,
                                                                    uint32_t thresh, float inv_keep, uint64_t seed,
                                                                    double* __restrict__ part, int premasked) {
    // premasked: dZ already carries the activation derivative and the dropout mask (see bn_act_bwd_apply_vec_kernel)
    __shared__ double shd[256 * 8];
    const int tid = threadIdx.x, tpr = C >> 2, nrg = 256 / tpr;
    const int cg = tid % tpr, rg = tid / tpr, c = cg * 4;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (rg < nrg) {
        const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
        const f32x4 mu = *(const f32x4*)(mean + c), is = *(const f32x4*)(invstd + c);
        const int r0 = blockIdx.x * VROWS, r1 = min(M, r0 + VROWS);
        for (int rb = r0 + rg; rb < r1; rb += 4 * nrg) {                    // four rows in flight per thread, summed in row order
            f32x4 y[4], dz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (rb + u * nrg < r1) { y[u] = ld4<TY>(Y + (size_t)(rb + u * nrg) * C + c); dz[u] = ld4<TY>(dZ + (size_t)(rb + u * nrg) * C + c); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (rb + u * nrg >= r1) break;
                const size_t i = (size_t)(rb + u * nrg) * C + c;
                const uint32_t hq = (thresh && !premasked) ? dropout_hash4(seed, i >> 2) : 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = premasked ? dz[u][e] : dz_prime_q(dz[u][e], y[u][e], sc[e], sh[e], act, slope, thresh, inv_keep, hq, e);
                    s[e] += d; q[e] += (double)d * ((y[u][e] - mu[e]) * is[e]);
                }
            }
        }
    }
    vec_block_reduce_write(s, q, C, tpr, tid, shd, part, blockIdx.x);
}

__global__ __launch_bounds__(256) void colstats_vec_kernel(const float* __restrict__ Y, int M, int C, double* __restrict__ part) {
    __shared__ double shd[256 * 8];
    const int tid = threadIdx.x, tpr = C >> 2, nrg = 256 / tpr;
    const int cg = tid % tpr, rg = tid / tpr, c = cg * 4;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (rg < nrg) {
        const int r0 = blockIdx.x * VROWS, r1 = min(M, r0 + VROWS);
        for (int r = r0 + rg; r < r1; r += nrg) {
            const f32x4 y = *(const f32x4*)(Y + (size_t)r * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += y[e]; q[e] += (double)y[e] * y[e]; }
        }
    }
    vec_block_reduce_write(s, q, C, tpr, tid, shd, part, blockIdx.x);
}

// column-stationary elementwise passes: a thread keeps its 4 columns' constants in registers and walks rows
template <typename TY>
__global__ __launch_bounds__(256) void bn_act_fwd_vec_kernel(const TY* __restrict__ Y, TY* __restrict__ Z, int M, int C,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             int act, float slope, uint32_t thresh, float inv_keep,
                                                             uint64_t seed) {
    const int tid = threadIdx.x, tpr = C >> 2, nrg = 256 / tpr;
    const int cg = tid % tpr, rg = tid / tpr, c = cg * 4;
    if (rg >= nrg) return;
    const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
    const int r0 = blockIdx.x * VROWS, r1 = min(M, r0 + VROWS);
    for (int rb = r0 + rg; rb < r1; rb += 4 * nrg) {                        // four rows in flight per thread (measured on the
        f32x4 y[4];                                                         // per-channel twins in multi.hip: 6-17 % per pass)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (rb + u * nrg < r1) y[u] = ld4<TY>(Y + (size_t)(rb + u * nrg) * C + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (rb + u * nrg >= r1) break;
            const size_t i = (size_t)(rb + u * nrg) * C + c;
            f32x4 o;
            const uint32_t hq = thresh ? dropout_hash4(seed, i >> 2) : 0u;  // i is a multiple of 4: one hash for the quad
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = lrelu_or_relu(fmaf(y[u][e], sc[e], sh[e]), act, slope);
                if (thresh) a = ((hq >> (8 * e)) & 255u) >= thresh ? a * inv_keep : 0.f;
                o[e] = a;
            }
            st4<TY>(Z + i, o);
        }
    }
}

template <typename TY>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_vec_kernel(const TY* __restrict__ dZ, const TY* __restrict__ Y,
                                                                   TY* __restrict__ dY, int M, int C,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ mean_dz,
                                                                   const float* __restrict__ mean_dzy, int act, float slope,
                                                                   uint32_t thresh, float inv_keep, uint64_t seed,
                                                                   float* __restrict__ gpart, int premasked) {
    // gpart (fp32 storage, 256 % (C / 4) == 0, a block's VROWS rows inside one group): this block's column sums of dY, laid out and
    // summed exactly as colsum_groups_vec_kernel does with one slab per block -- the per-cloud bias gradient of the heads' first
    // layer without reading dY again (colsum_groups_fin_kernel finishes it)
    __shared__ float shd[256 * 4];
    const int tid = threadIdx.x, tpr = C >> 2, nrg = 256 / tpr;
    const int cg = tid % tpr, rg = tid / tpr, c = cg * 4;
    if (rg >= nrg) return;                                              // (never with gpart: nrg * tpr == 256)
    const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
    f32x4 gs = {0.f, 0.f, 0.f, 0.f};
    f32x4 mu = {0, 0, 0, 0}, k1 = {0, 0, 0, 0}, k2 = {0, 0, 0, 0};      // d - k1 - (y - mu)*k2
    if (mean_dz) {
        mu = *(const f32x4*)(mean + c);
        k1 = *(const f32x4*)(mean_dz + c);
        const f32x4 is = *(const f32x4*)(invstd + c), mz = *(const f32x4*)(mean_dzy + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) k2[e] = is[e] * mz[e];
    }
    const int r0 = blockIdx.x * VROWS, r1 = min(M, r0 + VROWS);
    for (int rb = r0 + rg; rb < r1; rb += 4 * nrg) {                        // four rows in flight per thread
        f32x4 y[4], dz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (rb + u * nrg < r1) { y[u] = ld4<TY>(Y + (size_t)(rb + u * nrg) * C + c); dz[u] = ld4<TY>(dZ + (size_t)(rb + u * nrg) * C + c); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (rb + u * nrg >= r1) break;
            const size_t i = (size_t)(rb + u * nrg) * C + c;
            const uint32_t hq = (thresh && !premasked) ? dropout_hash4(seed, i >> 2) : 0u;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // premasked: dZ already carries the activation derivative and the dropout mask (applied where the consumer's dgrad produced it)
                float d = premasked ? dz[u][e] : dz_prime_q(dz[u][e], y[u][e], sc[e], sh[e], act, slope, thresh, inv_keep, hq, e);
                d = d - k1[e] - (y[u][e] - mu[e]) * k2[e];
                o[e] = sc[e] * d;
            }
            st4<TY>(dY + i, o);
            gs = gs + o;
        }
    }
    if (gpart) {
#pragma unroll
        for (int e = 0; e < 4; ++e) shd[tid * 4 + e] = gs[e];
        __syncthreads();
        if (tid < tpr) {
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < nrg; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] += shd[(q * tpr + tid) * 4 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) gpart[(size_t)blockIdx.x * C + tid * 4 + e] = a[e];
        }
    }
}

static inline bool vec_ok(int C, const void* a, const void* b = nullptr, const void* c = nullptr) {
    uintptr_t m = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c;
    return C % 4 == 0 && C >= 4 && C <= 1024 && (256 % (C / 4) == 0 || C / 4 > 128) && (m & 15) == 0;
}
int bn_vec_parts(int M) { return (M + VROWS - 1) / VROWS; }

// per-group column sums: out[g][c] = sum over rows of group g   (rows_per_group consecutive rows)
__global__ __launch_bounds__(256) void colsum_groups_kernel(const float* __restrict__ X, int C, int rows_per_group,
                                                            float* __restrict__ out) {
    __shared__ double sh[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, g = blockIdx.y;
    double s = 0.0;
    if (c < C)
        for (int r = w; r < rows_per_group; r += 4) s += X[((size_t)g * rows_per_group + r) * C + c];
    sh[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < C) out[(size_t)g * C + c] = (float)(sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane]);
}

// ---------------------------------------------------------------------------------------------
// max over the N points of each cloud:  out[b][c] = max_n Z[b*N+n][c], arg = first n attaining it
__global__ __launch_bounds__(256) void colmax_fwd_kernel(const float* __restrict__ Z, int N, int C,
                                                         float* __restrict__ out, int* __restrict__ arg) {
    __shared__ float sv[4][64];
    __shared__ int si[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, b = blockIdx.y;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (c < C)
        for (int n = w; n < N; n += 4) {
            float v = Z[((size_t)b * N + n) * C + c];
            if (v > best) { best = v; bi = n; }
        }
    sv[w][lane] = best; si[w][lane] = bi;
    __syncthreads();
    if (w == 0 && c < C) {
        for (int u = 1; u < 4; ++u) {
            float v = sv[u][lane]; int i = si[u][lane];
            if (v > best || (v == best && i < bi)) { best = v; bi = i; }
        }
        out[(size_t)b * C + c] = best;
        arg[(size_t)b * C + c] = bi == 0x7fffffff ? 0 : bi;
    }
}

// dZ[b*N + arg[b][c]][c] = dOut[b][c]   (dZ pre-zeroed by the caller)
__global__ void colmax_bwd_kernel(const float* __restrict__ dOut, const int* __restrict__ arg, int B, int N, int C,
                                  float* __restrict__ dZ) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    int b = i / C, c = i % C;
    dZ[((size_t)b * N + arg[i]) * C + c] = dOut[i];
}

// max over the k edges of each point: out[i][c] = max_s Z[i*k+s][c], argk = first s attaining it
__global__ __launch_bounds__(256) void segmax_fwd_kernel(const float* __restrict__ Z, int P, int k, int C,
                                                         float* __restrict__ out, uint8_t* __restrict__ argk) {
    size_t total = (size_t)P * C;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        size_t i = t / C; int c = (int)(t % C);
        const float* z = Z + (i * k) * C + c;
        float best = z[0]; int bs = 0;
        for (int s = 1; s < k; ++s) {
            float v = z[(size_t)s * C];
            if (v > best) { best = v; bs = s; }
        }
        out[t] = best; argk[t] = (uint8_t)bs;
    }
}

// dZ[(i*k+s)][c] = (s == argk[i][c]) ? dOut[i][c] : 0     (writes every element)
__global__ __launch_bounds__(256) void segmax_bwd_kernel(const float* __restrict__ dOut, const uint8_t* __restrict__ argk,
                                                         int P, int k, int C, float* __restrict__ dZ) {
    size_t total = (size_t)P * k * C;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(t % C);
        size_t e = t / C;
        size_t i = e / k; int s = (int)(e % k);
        dZ[t] = (argk[i * C + c] == s) ? dOut[i * C + c] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
static inline int ew_blocks(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b < 4096 ? (b ? b : 1) : 4096);
}
static inline uint32_t drop_thresh(float p) { return dropout_thresh8(p); }

int bn_stat_parts(int M) { return (M + STAT_ROWS - 1) / STAT_ROWS; }
// upper bound on the partial rows any BN reduction over M rows writes (vectorised form uses 64-row blocks)
int bn_parts_max(int M) { return (M + 63) / 64; }

// NOTE: returns the number of partial rows written through *nparts_out (vectorised and scalar forms differ)
int launch_colstats_n(hipStream_t st, const float* Y, int M, int C, int ld, double* part, int* nparts_out) {
    if (ld == C && vec_ok(C, Y) && 256 % (C / 4) == 0) {
        *nparts_out = bn_vec_parts(M);
        hipLaunchKernelGGL(colstats_vec_kernel, dim3(*nparts_out), dim3(256), 0, st, Y, M, C, part);
        return mlsp_launch_status();
    }
    *nparts_out = bn_stat_parts(M);
    hipLaunchKernelGGL(colstats_kernel, dim3((C + 63) / 64, bn_stat_parts(M)), dim3(256), 0, st, Y, M, C, ld, part);
    return mlsp_launch_status();
}

int launch_colstats(hipStream_t st, const float* Y, int M, int C, int ld, double* part) {
    hipLaunchKernelGGL(colstats_kernel, dim3((C + 63) / 64, bn_stat_parts(M)), dim3(256), 0, st, Y, M, C, ld, part);
    return mlsp_launch_status();
}

// request of the NEXT launch_bn_finalize on this thread: also write the channels' output bounds (bn_finalize_kernel bound_out); consumed
// (cleared) by that launch -- the pattern of gemm_unfold_request
static thread_local float* tl_bound_out = nullptr;
void bn_bound_request(float* out) { tl_bound_out = out; }
int launch_bn_finalize(hipStream_t st, const double* part, int nparts, double count, int C, const float* gamma,
                       const float* beta, float* run_mean, float* run_var, float momentum, float eps, float* scale,
                       float* shift, float* save_mean, float* save_invstd) {
    float* bo = tl_bound_out;
    tl_bound_out = nullptr;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(FIN_THREADS), 0, st, part, nparts, count, C, gamma, beta,
                       run_mean, run_var, momentum, eps, scale, shift, save_mean, save_invstd, (float*)nullptr, 1, bo, (float)sqrt(count));
    return mlsp_launch_status();
}

// the same + gsum [nparts / ppg][C]: the column sums of every ppg consecutive row panels (one cloud)
int launch_bn_finalize_groups(hipStream_t st, const double* part, int nparts, double count, int C, const float* gamma,
                              const float* beta, float* run_mean, float* run_var, float momentum, float eps, float* scale,
                              float* shift, float* save_mean, float* save_invstd, float* gsum, int ppg) {
    if (!gsum || ppg <= 0 || nparts % ppg) return MLSP_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(FIN_THREADS), 0, st, part, nparts, count, C, gamma, beta,
                       run_mean, run_var, momentum, eps, scale, shift, save_mean, save_invstd, gsum, ppg);
    return mlsp_launch_status();
}

int launch_bn_eval_prepare(hipStream_t st, int C, const float* gamma, const float* beta, const float* run_mean,
                           const float* run_var, float eps, float* scale, float* shift, float* save_mean,
                           float* save_invstd) {
    hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3((C + 127) / 128), dim3(128), 0, st, C, gamma, beta, run_mean, run_var,
                       eps, scale, shift, save_mean, save_invstd);
    return mlsp_launch_status();
}

int launch_bn_act_fwd(hipStream_t st, const float* Y, float* Z, size_t rows, int C, const float* scale,
                      const float* shift, int act, float slope, float p_drop, uint64_t seed) {
    size_t total = rows * C;
    float inv_keep = dropout_inv_keep8(p_drop);
    if (rows < (size_t)1 << 30 && vec_ok(C, Y, Z) && 256 % (C / 4) == 0 && ((((uintptr_t)scale | (uintptr_t)shift) & 15) == 0)) {
        hipLaunchKernelGGL((bn_act_fwd_vec_kernel<float>), dim3(bn_vec_parts((int)rows)), dim3(256), 0, st, Y, Z, (int)rows, C, scale, shift,
                           act, slope, drop_thresh(p_drop), inv_keep, seed);
        return mlsp_launch_status();
    }
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, Y, Z, total, C, scale, shift, act,
                       slope, drop_thresh(p_drop), inv_keep, seed);
    return mlsp_launch_status();
}

static float* bn_zero_vec_consume();
int launch_bn_act_bwd(hipStream_t st, const float* dZ, const float* Y, float* dY, int M, int C, const float* scale,
                      const float* shift, const float* mean, const float* invstd, int training, int act, float slope,
                      float p_drop, uint64_t seed, double* part, float* dgamma, float* dbeta, float* mean_dz,
                      float* mean_dzy, float* gpart, int rows_per_group, int* gpart_slabs, const double* pre_stats, int pre_parts) {
    // gpart / gpart_slabs (nullable): ask for the per-group column sums of dY as a by-product; *gpart_slabs is set to the slab count
    // written per group ([G][slabs][C] floats, finish with launch_colsum_groups_fin) or to 0 when this shape does not fuse them
    // pre_stats (nullable, [pre_parts][2][C]): dZ arrives MASKED and its column sums are already there (the consumer's dgrad left them:
    // gemm.hip gemm_out_bs) -- no reduction pass, the apply pass only does the BatchNorm part
    if (gpart_slabs) *gpart_slabs = 0;
    if (pre_stats) {
        if (!(vec_ok(C, dZ, Y, dY) && 256 % (C / 4) == 0 && ((((uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)invstd) & 15) == 0)))
            return MLSP_ERR_UNSUPPORTED;
        const bool fuse_g = gpart && gpart_slabs && rows_per_group > 0 && rows_per_group % VROWS == 0 && rows_per_group / VROWS <= 16 && M % rows_per_group == 0;
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(FIN_THREADS), 0, st, pre_stats, pre_parts, (double)M, C, dgamma, dbeta, mean_dz, mean_dzy, bn_zero_vec_consume());
        hipLaunchKernelGGL((bn_act_bwd_apply_vec_kernel<float>), dim3(bn_vec_parts(M)), dim3(256), 0, st, dZ, Y, dY, M, C, scale, shift, mean, invstd,
                           training ? mean_dz : (const float*)nullptr, mean_dzy, act, slope, 0u, 1.f, seed, fuse_g ? gpart : (float*)nullptr, 1);
        if (fuse_g) *gpart_slabs = rows_per_group / VROWS;
        return mlsp_launch_status();
    }
    float inv_keep = dropout_inv_keep8(p_drop);
    uint32_t th = drop_thresh(p_drop);
    int nparts = bn_stat_parts(M);
    const bool vec = vec_ok(C, dZ, Y, dY) && 256 % (C / 4) == 0 && ((((uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)invstd) & 15) == 0);
    // pre_parts < 0 without pre_stats: dZ arrives MASKED (some consumers' dgrads applied the activation derivative / dropout mask, the
    // columns of the others are zero) but the sums are incomplete: reduce here, without applying the mask a second time
    const int premasked = pre_parts < 0 ? 1 : 0;
    if (premasked) {
        if (!vec) return MLSP_ERR_UNSUPPORTED;
        th = 0u; inv_keep = 1.f;
    }
    const bool fuse_g = vec && gpart && gpart_slabs && rows_per_group > 0 && rows_per_group % VROWS == 0 && rows_per_group / VROWS <= 16 &&
                        M % rows_per_group == 0;
    if (vec) {
        nparts = bn_vec_parts(M);
        hipLaunchKernelGGL((bn_act_bwd_reduce_vec_kernel<float>), dim3(nparts), dim3(256), 0, st, dZ, Y, M, C, scale, shift, mean, invstd,
                           act, slope, th, inv_keep, seed, part, premasked);
    } else {
        hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3((C + 63) / 64, nparts), dim3(256), 0, st, dZ, Y, M, C, scale, shift,
                           mean, invstd, act, slope, th, inv_keep, seed, part);
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(FIN_THREADS), 0, st, part, nparts, (double)M, C, dgamma,
                       dbeta, mean_dz, mean_dzy, bn_zero_vec_consume());
    size_t total = (size_t)M * C;
    if (vec) {
        hipLaunchKernelGGL((bn_act_bwd_apply_vec_kernel<float>), dim3(bn_vec_parts(M)), dim3(256), 0, st, dZ, Y, dY, M, C,
                           scale, shift, mean, invstd, training ? mean_dz : (const float*)nullptr, mean_dzy, act, slope, th,
                           inv_keep, seed, fuse_g ? gpart : (float*)nullptr, premasked);
        if (fuse_g) *gpart_slabs = rows_per_group / VROWS;
        return mlsp_launch_status();
    }
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, dZ, Y, dY, total, C, scale, shift,
                       mean, invstd, training ? mean_dz : (const float*)nullptr, mean_dzy, act, slope, th, inv_keep, seed);
    return mlsp_launch_status();
}

// The same sums turned into the coefficient rows of the on-the-fly BatchNorm backward (gemm.hip gemm_split_kernel<.., DY>):
//   dY = scale * (d' - mean_dz - (y - mean) * invstd * mean_dzy) = (d' + y * nk2 + c0) * sc,
//   nk2 = -invstd * mean_dzy,  c0 = mean * invstd * mean_dzy - mean_dz,  sc = scale;   coef: rows c0 | nk2 | sc of pitch C.
// gout (nullable): the per-cloud column sums of that dY, never formed -- sc * (sum d' + nk2 * sum y + rows * c0) from the row-panel sums of d'
// (ppg panels per cloud) and the clouds' column sums of y (gys [G][C], kept by the forward: bn_finalize_kernel gsum)
__global__ void bn_bwd_finalize_coef_kernel(const double* __restrict__ part, int nparts, double count, int C, const float* __restrict__ bn,
                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef,
                                            const float* __restrict__ gys, int ppg, int rows, float* __restrict__ gout,
                                            float* __restrict__ zero_vec = nullptr, int with_amax = 0) {
    // with_amax: behind the nparts x 2 x C sums lie nparts x C floats -- the panels' column maxima of |d'| (gemm.hip gemm_out_bs, thin.hip):
    // their maximum becomes the FOURTH coefficient row (a bound of d' per channel for the two-piece f16 products of the layer's GEMMs)
    const int c = blockIdx.x;
    double s, q;
    fin_part_sums(part, nparts, C, c, s, q);
    if (with_amax) {
        __shared__ float amred[FIN_THREADS / 64];
        const float* plane = (const float*)(part + (size_t)nparts * 2 * C);
        float m = 0.f;
        for (int i = threadIdx.x; i < nparts; i += FIN_THREADS) m = fmaxf(m, plane[(size_t)i * C + c]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) amred[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) coef[3 * C + c] = fmaxf(fmaxf(amred[0], amred[1]), fmaxf(amred[2], amred[3]));
    }
    const double k2 = (double)bn[3 * C + c] * (q / count);
    const float c0 = (float)((double)bn[2 * C + c] * k2 - s / count), nk2 = (float)(-k2), sc = bn[c];
    if (gout)
        for (int g = threadIdx.x; g < nparts / ppg; g += blockDim.x) {
            double sd = 0.0;
            for (int p = 0; p < ppg; ++p) sd += part[((size_t)(g * ppg + p) * 2) * C + c];
            gout[(size_t)g * C + c] = (float)((double)sc * (sd + (double)nk2 * (double)gys[(size_t)g * C + c] + (double)rows * (double)c0));
        }
    if (threadIdx.x != 0) return;
    if (zero_vec) zero_vec[c] = 0.f;
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
    coef[c] = c0;
    coef[C + c] = nk2;
    coef[2 * C + c] = sc;
}

// (coef: FOUR rows of pitch C -- c0 | nk2 | sc | max |d'| per channel, the last from the maxima plane behind the sums: include/mlsp_hip.h)
int launch_bn_bwd_finalize_coef_z(hipStream_t st, const double* part, int nparts, double count, int C, const float* bn_save, float* dgamma,
                                  float* dbeta, float* coef, float* zero_vec) {
    hipLaunchKernelGGL(bn_bwd_finalize_coef_kernel, dim3(C), dim3(FIN_THREADS), 0, st, part, nparts, count, C, bn_save, dgamma, dbeta, coef,
                       (const float*)nullptr, 1, 0, (float*)nullptr, zero_vec, 1);
    return mlsp_launch_status();
}
int launch_bn_bwd_finalize_coef(hipStream_t st, const double* part, int nparts, double count, int C, const float* bn_save, float* dgamma,
                                float* dbeta, float* coef) {
    return launch_bn_bwd_finalize_coef_z(st, part, nparts, count, C, bn_save, dgamma, dbeta, coef, nullptr);
}
int launch_bn_bwd_finalize_coef_groups(hipStream_t st, const double* part, int nparts, double count, int C, const float* bn_save, float* dgamma,
                                       float* dbeta, float* coef, const float* gys, int ppg, int rows, float* gout) {
    if (!gys || !gout || ppg <= 0 || nparts % ppg) return MLSP_ERR_ARG;
    hipLaunchKernelGGL(bn_bwd_finalize_coef_kernel, dim3(C), dim3(FIN_THREADS), 0, st, part, nparts, count, C, bn_save, dgamma, dbeta, coef, gys,
                       ppg, rows, gout, (float*)nullptr, 1);
    return mlsp_launch_status();
}

int launch_bn_bwd_finalize_z(hipStream_t st, const double* part, int nparts, double count, int C, float* dgamma, float* dbeta,
                             float* mean_dz, float* mean_dzy, float* zero_vec) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(FIN_THREADS), 0, st, part, nparts, count, C, dgamma, dbeta,
                       mean_dz, mean_dzy, zero_vec);
    return mlsp_launch_status();
}
int launch_bn_bwd_finalize(hipStream_t st, const double* part, int nparts, double count, int C, float* dgamma, float* dbeta,
                           float* mean_dz, float* mean_dzy) {
    return launch_bn_bwd_finalize_z(st, part, nparts, count, C, dgamma, dbeta, mean_dz, mean_dzy, nullptr);
}

// out[c] = sum over partial blocks of the column sums (first plane of colstats partials)
__global__ void colsum_finalize_kernel(const double* __restrict__ part, int nparts, int C, float* __restrict__ out) {
    const int c = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[((size_t)i * 2) * C + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[c] = (float)s;
}

// column sums over all M rows of a [M][C] matrix (bias gradients); part: [bn_stat_parts(M)][2][C] doubles
// vectorised per-group column sums: block = (64-row slab of a group); partial sums combined by a second tiny pass
template <typename TY>
__global__ __launch_bounds__(256) void colsum_groups_vec_kernel(const TY* __restrict__ X, int C, int rows_per_group, int slabs,
                                                                float* __restrict__ part) {
    __shared__ float shd[256 * 4];
    const int tid = threadIdx.x, tpr = C >> 2, nrg = 256 / tpr;
    const int cg = tid % tpr, rg = tid / tpr, c = cg * 4;
    const int g = blockIdx.y, slab = blockIdx.x;
    const int rows_per_slab = (rows_per_group + slabs - 1) / slabs;
    const int r0 = slab * rows_per_slab, r1 = min(rows_per_group, r0 + rows_per_slab);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (rg < nrg) {
        // four rows in flight per thread, four accumulators (a single chain waited for each 16-byte load: 2.4 TB/s on the [32768, 512]
        // input of conv5's backward), combined in a fixed order
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, s3 = {0.f, 0.f, 0.f, 0.f};
        const TY* base = X + (size_t)g * rows_per_group * C + c;
        int r = r0 + rg;
        // (whole-matrix sums only: the per-cloud form keeps the single chain that the fused BatchNorm-backward pass restates element by
        // element -- bn_act_bwd_apply_vec_kernel's per-group partial sums)
        if (gridDim.y == 1)
        for (; r + 3 * nrg < r1; r += 4 * nrg) {
            const f32x4 a0 = ld4<TY>(base + (size_t)r * C), a1 = ld4<TY>(base + (size_t)(r + nrg) * C);
            const f32x4 a2 = ld4<TY>(base + (size_t)(r + 2 * nrg) * C), a3 = ld4<TY>(base + (size_t)(r + 3 * nrg) * C);
            s = s + a0; s1 = s1 + a1; s2 = s2 + a2; s3 = s3 + a3;
        }
        for (; r < r1; r += nrg) s = s + ld4<TY>(base + (size_t)r * C);
        s = (s + s1) + (s2 + s3);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) shd[tid * 4 + e] = s[e];
    __syncthreads();
    if (tid < tpr) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < nrg; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += shd[(q * tpr + tid) * 4 + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) part[((size_t)g * slabs + slab) * C + tid * 4 + e] = a[e];
    }
}
// 16 lanes per output: lane l adds slabs l, l + 16, ... (four loads in flight), the sixteen partial sums are added in lane order (one
// thread walked all the slabs before: up to 1,024 dependent loads, 10-12 us for 512 outputs)
__global__ __launch_bounds__(256) void colsum_groups_fin_kernel(const float* __restrict__ part, int G, int C, int slabs, float* __restrict__ out) {
    __shared__ float sh[256];
    const int tid = threadIdx.x, l = tid & 15;
    const int t = blockIdx.x * 16 + (tid >> 4);
    float s = 0.f;
    if (t < G * C) {
        const int g = t / C, c = t % C;
        const float* p = part + (size_t)g * slabs * C + c;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int u = l;
        for (; u + 48 < slabs; u += 64) {
            const float a0 = p[(size_t)u * C], a1 = p[(size_t)(u + 16) * C], a2 = p[(size_t)(u + 32) * C], a3 = p[(size_t)(u + 48) * C];
            s += a0; s1 += a1; s2 += a2; s3 += a3;
        }
        for (; u < slabs; u += 16) s += p[(size_t)u * C];
        s = (s + s1) + (s2 + s3);
    }
    sh[tid] = s;
    __syncthreads();
    if (l == 0 && t < G * C) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += sh[tid + q];
        out[t] = a;
    }
}

int launch_colsum_groups(hipStream_t st, const float* X, int G, int rows_per_group, int C, float* out, float* scratch = nullptr) {
    if (scratch && vec_ok(C, X) && 256 % (C / 4) == 0 && rows_per_group >= 256) {
        const int slabs = 16;                       // scratch: [G][16][C] floats
        hipLaunchKernelGGL((colsum_groups_vec_kernel<float>), dim3(slabs, G), dim3(256), 0, st, X, C, rows_per_group, slabs, scratch);
        hipLaunchKernelGGL(colsum_groups_fin_kernel, dim3((G * C + 15) / 16), dim3(256), 0, st, scratch, G, C, slabs, out);
        return mlsp_launch_status();
    }
    hipLaunchKernelGGL(colsum_groups_kernel, dim3((C + 63) / 64, G), dim3(256), 0, st, X, C, rows_per_group, out);
    return mlsp_launch_status();
}

// Per-cloud column sums of that dY without forming it: sc * (sum d' + nk2 * sum y + rows * c0) from the consumer's row-panel sums of d'
// (stats [panels][2][C], ppg panels per cloud) and the cloud's column sums of y (ys [G][slabs][C], colsum_groups_vec_kernel).
__global__ void bn_dy_gbias_kernel(const double* __restrict__ stats, int ppg, const float* __restrict__ ys, int slabs, const float* __restrict__ coef,
                                   int G, int C, int rows, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * C) return;
    const int g = i / C, c = i - g * C;
    double sd = 0.0, sy = 0.0;
    for (int p = 0; p < ppg; ++p) sd += stats[((size_t)(g * ppg + p) * 2) * C + c];
    for (int s = 0; s < slabs; ++s) sy += (double)ys[((size_t)g * slabs + s) * C + c];
    out[i] = (float)((double)coef[2 * C + c] * (sd + (double)coef[C + c] * sy + (double)rows * (double)coef[c]));
}

// scratch: [G][16][C] floats; stats: [M / panel rows][2][C]
// ysum (nullable, [G][C]): the clouds' column sums of y kept by the forward (bn_finalize_kernel gsum): no pass over Y here
int launch_bn_dy_gbias(hipStream_t st, const float* Y, int G, int rows_per_group, int C, const double* stats, int panel_rows, const float* coef,
                       float* scratch, float* out, const float* ysum) {
    if (!(vec_ok(C, Y) && 256 % (C / 4) == 0 && rows_per_group >= 256 && panel_rows > 0 && rows_per_group % panel_rows == 0)) return MLSP_ERR_UNSUPPORTED;
    const int slabs = ysum ? 1 : 16;
    if (!ysum) hipLaunchKernelGGL((colsum_groups_vec_kernel<float>), dim3(slabs, G), dim3(256), 0, st, Y, C, rows_per_group, slabs, scratch);
    hipLaunchKernelGGL(bn_dy_gbias_kernel, dim3((G * C + 255) / 256), dim3(256), 0, st, stats, rows_per_group / panel_rows, ysum ? ysum : scratch, slabs, coef,
                       G, C, rows_per_group, out);
    return mlsp_launch_status();
}

// second half of launch_colsum_groups alone: scratch [G][slabs][C] written by bn_act_bwd_apply_vec_kernel (launch_bn_act_bwd, gpart)
int launch_colsum_groups_fin(hipStream_t st, const float* scratch, int G, int C, int slabs, float* out) {
    hipLaunchKernelGGL(colsum_groups_fin_kernel, dim3((G * C + 15) / 16), dim3(256), 0, st, scratch, G, C, slabs, out);
    return mlsp_launch_status();
}

// column sums over all M rows of a [M][C] matrix (bias gradients, Gram-matrix terms); part: [bn_stat_parts(M)][2][C] doubles
int launch_colsum(hipStream_t st, const float* X, int M, int C, double* part, float* out) {
    int nparts = bn_stat_parts(M);
    if (vec_ok(C, X) && 256 % (C / 4) == 0 && M >= 4096) {
        // streaming shape: 16-byte column-stationary partial sums over 4*nparts row slabs (the fp64 partial buffer reused as floats)
        const int slabs = 4 * nparts;
        float* scratch = (float*)part;
        hipLaunchKernelGGL((colsum_groups_vec_kernel<float>), dim3(slabs, 1), dim3(256), 0, st, X, C, M, slabs, scratch);
        hipLaunchKernelGGL(colsum_groups_fin_kernel, dim3((C + 15) / 16), dim3(256), 0, st, scratch, 1, C, slabs, out);
        return mlsp_launch_status();
    }
    hipLaunchKernelGGL(colstats_kernel, dim3((C + 63) / 64, nparts), dim3(256), 0, st, X, M, C, C, part);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(C), dim3(64), 0, st, part, nparts, C, out);
    return mlsp_launch_status();
}

// column partial sums of dz' = dZ * act'(Y*scale + shift) and dz' * yhat only (the first pass of launch_bn_act_bwd), vectorised form:
// writes bn_vec_parts(M) rows of [2][C] doubles.  Requires C % 4 == 0, 256 % (C/4) == 0 and 16-byte aligned operands.
int launch_bn_act_bwd_partials_vec(hipStream_t st, const float* dZ, const float* Y, int M, int C, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, int act, float slope, double* part) {
    if (!(vec_ok(C, dZ, Y) && 256 % (C / 4) == 0 && ((((uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)invstd) & 15) == 0)))
        return MLSP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((bn_act_bwd_reduce_vec_kernel<float>), dim3(bn_vec_parts(M)), dim3(256), 0, st, dZ, Y, M, C, scale, shift, mean, invstd, act,
                       slope, 0u, 1.f, (uint64_t)0, part, 0);
    return mlsp_launch_status();
}

int launch_colmax_fwd(hipStream_t st, const float* Z, int B, int N, int C, float* out, int* arg) {
    hipLaunchKernelGGL(colmax_fwd_kernel, dim3((C + 63) / 64, B), dim3(256), 0, st, Z, N, C, out, arg);
    return mlsp_launch_status();
}

int launch_colmax_bwd(hipStream_t st, const float* dOut, const int* arg, int B, int N, int C, float* dZ) {
    hipError_t e = hipMemsetAsync(dZ, 0, (size_t)B * N * C * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(colmax_bwd_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, dOut, arg, B, N, C, dZ);
    return mlsp_launch_status();
}

int launch_segmax_fwd(hipStream_t st, const float* Z, int P, int k, int C, float* out, uint8_t* argk) {
    hipLaunchKernelGGL(segmax_fwd_kernel, dim3(ew_blocks((size_t)P * C)), dim3(256), 0, st, Z, P, k, C, out, argk);
    return mlsp_launch_status();
}

// ---- BatchNorm + activation + max over the k rows of every group, fused (the last conv of a set-abstraction MLP, pointnet_util.py:
// 188-195) -- the activated [G*k, C] tensor is never written.  act(scale*y + shift) is monotone in y with the sign of scale, so the group
// extreme of the PRE-BN values (max for scale >= 0, min otherwise; first slot attaining it) decides:  out = act(scale*ysel + shift).
// thread = (group, channel quad); a group's k rows are read as coalesced 16-byte quads.
__global__ __launch_bounds__(256) void segsel_act_fwd_kernel(const float* __restrict__ Y, int G, int k, int C, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int act, float slope, float* __restrict__ out,
                                                             float* __restrict__ ysel, uint8_t* __restrict__ argk) {
    const int tpr = C >> 2;
    const size_t total = (size_t)G * tpr;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t g = t / tpr;
        const int c = (int)(t % tpr) * 4;
        const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
        const float* y0 = Y + (g * k) * C + c;
        f32x4 best = *(const f32x4*)y0;
        int bs[4] = {0, 0, 0, 0};
        for (int s0 = 1; s0 < k; s0 += 4) {                                // four rows in flight
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(y0 + (size_t)(s0 + u < k ? s0 + u : 0) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (s0 + u < k) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool take = sc[e] >= 0.f ? v[u][e] > best[e] : v[u][e] < best[e];
                        best[e] = take ? v[u][e] : best[e]; bs[e] = take ? s0 + u : bs[e];
                    }
                }
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = lrelu_or_relu(fmaf(best[e], sc[e], sh[e]), act, slope);
        *(f32x4*)(out + g * C + c) = o;
        *(f32x4*)(ysel + g * C + c) = best;
        *(uint32_t*)(argk + g * C + c) = (uint32_t)bs[0] | ((uint32_t)bs[1] << 8) | ((uint32_t)bs[2] << 16) | ((uint32_t)bs[3] << 24);
    }
}

// dY[(g,s)][c] = scale * ( [s == arg] * dOut[g][c] * act'(scale*ysel + shift) - m1 - (y - mean) * invstd * m2 )     (eval: m1 = m2 = 0)
// thread = (row, channel quad): reads Y, the group's arg / dOut / ysel quads, writes dY.
__global__ __launch_bounds__(256) void segsel_bwd_apply_kernel(const float* __restrict__ dOut, const float* __restrict__ Y, const float* __restrict__ ysel,
                                                               const uint8_t* __restrict__ argk, size_t M, int k, int C,
                                                               const float* __restrict__ bn, const float* __restrict__ m1v,
                                                               const float* __restrict__ m2v, int act, float slope, float* __restrict__ dY) {
    const int tpr = C >> 2;
    const size_t total = M * tpr;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t r = t / tpr;
        const int c = (int)(t % tpr) * 4;
        const size_t g = r / k;
        const int s = (int)(r % k);
        const f32x4 sc = *(const f32x4*)(bn + c), sh = *(const f32x4*)(bn + C + c);
        const f32x4 y = *(const f32x4*)(Y + r * C + c);
        const uint32_t a4 = *(const uint32_t*)(argk + g * C + c);
        f32x4 o;
        if (m1v) {
            const f32x4 mu = *(const f32x4*)(bn + 2 * C + c), is = *(const f32x4*)(bn + 3 * C + c);
            const f32x4 m1 = *(const f32x4*)(m1v + c), m2 = *(const f32x4*)(m2v + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = -m1[e] - (y[e] - mu[e]) * is[e] * m2[e];
        } else {
            o = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (((a4 & 255u) == (uint32_t)s) | (((a4 >> 8) & 255u) == (uint32_t)s) | (((a4 >> 16) & 255u) == (uint32_t)s) | ((a4 >> 24) == (uint32_t)s)) {
            const f32x4 d = *(const f32x4*)(dOut + g * C + c), ys = *(const f32x4*)(ysel + g * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (((a4 >> (8 * e)) & 255u) == (uint32_t)s) {
                    const float a = fmaf(ys[e], sc[e], sh[e]);
                    const float dp = act == 0 ? 1.f : (a > 0.f ? 1.f : (act == 2 ? slope : 0.f));
                    o[e] += d[e] * dp;
                }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] *= sc[e];
        *(f32x4*)(dY + r * C + c) = o;
    }
}

int launch_segsel_act_fwd(hipStream_t st, const float* Y, int G, int k, int C, const float* scale, const float* shift, int act, float slope,
                          float* out, float* ysel, uint8_t* argk) {
    if (C % 4 || k < 1 || k > 255 || ((((uintptr_t)Y | (uintptr_t)out | (uintptr_t)ysel | (uintptr_t)scale | (uintptr_t)shift) & 15) != 0) ||
        (((uintptr_t)argk) & 3)) return MLSP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(segsel_act_fwd_kernel, dim3(ew_blocks((size_t)G * (C / 4))), dim3(256), 0, st, Y, G, k, C, scale, shift, act, slope, out, ysel, argk);
    return mlsp_launch_status();
}
int launch_segsel_bwd_apply(hipStream_t st, const float* dOut, const float* Y, const float* ysel, const uint8_t* argk, size_t M, int k, int C,
                            const float* bn, const float* m1, const float* m2, int act, float slope, float* dY) {
    hipLaunchKernelGGL(segsel_bwd_apply_kernel, dim3(ew_blocks(M * (C / 4))), dim3(256), 0, st, dOut, Y, ysel, argk, M, k, C, bn, m1, m2, act, slope, dY);
    return mlsp_launch_status();
}

int launch_segmax_bwd(hipStream_t st, const float* dOut, const uint8_t* argk, int P, int k, int C, float* dZ) {
    hipLaunchKernelGGL(segmax_bwd_kernel, dim3(ew_blocks((size_t)P * k * C)), dim3(256), 0, st, dOut, argk, P, k, C, dZ);
    return mlsp_launch_status();
}


// ---- bf16 activation storage (BASELINE.json configs[4]): the same column-stationary passes on bf16 Y / Z / dZ / dY ------------
// (fp32 arithmetic and fp64 partial sums as above; only the loads and stores are 2 bytes per element).  Vectorised shapes only.
static inline bool vec_ok_b16(int C, const void* a, const void* b = nullptr, const void* c = nullptr) {
    return C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 7) == 0);
}
int launch_bn_act_fwd_b16(hipStream_t st, const void* Y, void* Z, int rows, int C, const float* scale, const float* shift, int act,
                          float slope, float p_drop, uint64_t seed) {
    if (!vec_ok_b16(C, Y, Z) || ((((uintptr_t)scale | (uintptr_t)shift) & 15) != 0)) return MLSP_ERR_UNSUPPORTED;
    const float inv_keep = dropout_inv_keep8(p_drop);
    hipLaunchKernelGGL((bn_act_fwd_vec_kernel<__bf16>), dim3(bn_vec_parts(rows)), dim3(256), 0, st, (const __bf16*)Y, (__bf16*)Z, rows, C,
                       scale, shift, act, slope, drop_thresh(p_drop), inv_keep, seed);
    return mlsp_launch_status();
}
// A [C] vector the NEXT BatchNorm-backward finalizer launched on this thread also zero-fills (the gradient of a bias in front of a
// batch-statistics BatchNorm): bn_zero_vec_request() before launch_bn_act_bwd / launch_bn_act_bwd_b16, bn_zero_vec_take() afterwards says
// whether a finalizer took it (the caller memsets otherwise).
static thread_local float* tl_zero_vec = nullptr;
static thread_local bool tl_zero_done = false;
void bn_zero_vec_request(float* v) { tl_zero_vec = v; tl_zero_done = false; }
bool bn_zero_vec_take() { const bool d = tl_zero_done; tl_zero_vec = nullptr; tl_zero_done = false; return d; }
static float* bn_zero_vec_consume() { float* v = tl_zero_vec; if (v) { tl_zero_vec = nullptr; tl_zero_done = true; } return v; }

int launch_bn_act_bwd_b16(hipStream_t st, const void* dZ, const void* Y, void* dY, int M, int C, const float* scale, const float* shift,
                          const float* mean, const float* invstd, int training, int act, float slope, float p_drop, uint64_t seed,
                          double* part, float* dgamma, float* dbeta, float* mean_dz, float* mean_dzy) {
    if (!vec_ok_b16(C, dZ, Y, dY) || ((((uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)invstd) & 15) != 0))
        return MLSP_ERR_UNSUPPORTED;
    const float inv_keep = dropout_inv_keep8(p_drop);
    const uint32_t th = drop_thresh(p_drop);
    const int nparts = bn_vec_parts(M);
    hipLaunchKernelGGL((bn_act_bwd_reduce_vec_kernel<__bf16>), dim3(nparts), dim3(256), 0, st, (const __bf16*)dZ, (const __bf16*)Y, M, C, scale,
                       shift, mean, invstd, act, slope, th, inv_keep, seed, part, 0);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(FIN_THREADS), 0, st, part, nparts, (double)M, C, dgamma, dbeta, mean_dz, mean_dzy, bn_zero_vec_consume());
    hipLaunchKernelGGL((bn_act_bwd_apply_vec_kernel<__bf16>), dim3(nparts), dim3(256), 0, st, (const __bf16*)dZ, (const __bf16*)Y, (__bf16*)dY,
                       M, C, scale, shift, mean, invstd, training ? mean_dz : (const float*)nullptr, mean_dzy, act, slope, th, inv_keep, seed, (float*)nullptr, 0);
    return mlsp_launch_status();
}
int launch_colsum_groups_b16(hipStream_t st, const void* X, int G, int rows_per_group, int C, float* out, float* scratch) {
    if (!scratch || !vec_ok_b16(C, X) || rows_per_group < 256) return MLSP_ERR_UNSUPPORTED;
    const int slabs = 16;
    hipLaunchKernelGGL((colsum_groups_vec_kernel<__bf16>), dim3(slabs, G), dim3(256), 0, st, (const __bf16*)X, C, rows_per_group, slabs, scratch);
    hipLaunchKernelGGL(colsum_groups_fin_kernel, dim3((G * C + 15) / 16), dim3(256), 0, st, scratch, G, C, slabs, out);
    return mlsp_launch_status();
}
