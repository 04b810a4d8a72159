// Several Linear (+bias) layers SIDE BY SIDE under ONE BatchNorm + activation (+dropout) pass.
//
// The three masked-local-structure heads (PointDA/Models.py:165-285) are three stacks of the same depth on the same points:
//   position / normal head   conv2 256->256 + bn2 + ReLU + dropout | conv3 256->128 + bn3 + ReLU        (:193-196, :227-230)
//   cardinality head         mlp1  512->256 + BN + LeakyReLU + dropout | mlp2 256->256 + BN + LeakyReLU + dropout  (:274-279)
// BatchNorm, the activation and dropout are per-channel operations, so layer d of all heads can live in ONE [M][sum Cout] matrix:
// each head's Linear writes its column slice (its own GEMM launch, BatchNorm sums out of the GEMM epilogue into the slice's
// columns of one partial buffer), and ONE statistics finalisation + ONE streaming pass serve all of them.  Backward likewise: one
// reduction, one finalisation, one apply pass, then per head the dgrad into its column slice of dX and the wgrad.
// Per channel the activation is max(a, slope_c * a) (slope 0 = ReLU, 0.2 = LeakyReLU, 1 = none) and dropout is on or off
// (chan [2][C]: slope row, dropout-switch row); all dropout channels share one rate.
// Versus one mlsp_pointmlp_* call per head: 2 finalisations + 2 streaming launches fewer per layer forward, 6 fewer backward.
#include "common.h"
#include "../../include/mlsp_hip.h"
#include <cstdlib>

int launch_gemm(hipStream_t st, bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                float* C, int ldc, const float* bias, const float* gbias, int rows_per_group, float* slab, size_t slab_floats,
                double* stat_part, const float* sel_gamma, float* sel_val, int* sel_row, bool accumulate, const GemmXf* xf, int stat_ld,
                const GemmGroups* grp = nullptr, const GemmBs* bs = nullptr, const GemmDy* dy = nullptr);
int gemm_bs_parts(int M, int N, int K, int lda, int ldb, int ldc);
int gemm_panel_rows(int M, int N, int K);
bool gemm_xf_supported(bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, int which);
bool gemm_xf_on_split(bool ta, bool tb, int M, int N, int K, int which);
int gemm_precision_mode();
int gemm_stat_parts(int M, int N, int K);
size_t gemm_slab_floats(int M, int N, int K);
int bn_vec_parts(int M);
int launch_colstats_n(hipStream_t st, const float* Y, int M, int C, int ld, double* part, int* nparts_out);
int launch_bn_finalize(hipStream_t st, const double* part, int nparts, double count, int C, const float* gamma,
                       const float* beta, float* run_mean, float* run_var, float momentum, float eps, float* scale,
                       float* shift, float* save_mean, float* save_invstd);
int launch_bn_eval_prepare(hipStream_t st, int C, const float* gamma, const float* beta, const float* run_mean,
                           const float* run_var, float eps, float* scale, float* shift, float* save_mean, float* save_invstd);
int launch_bn_bwd_finalize_coef(hipStream_t st, const double* part, int nparts, double count, int C, const float* bn_save, float* dgamma,
                                float* dbeta, float* coef);
bool gemm_dy_supported(bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb);
int launch_bn_bwd_finalize(hipStream_t st, const double* part, int nparts, double count, int C, float* dgamma, float* dbeta,
                           float* mean_dz, float* mean_dzy);
int launch_bn_bwd_finalize_z(hipStream_t st, const double* part, int nparts, double count, int C, float* dgamma, float* dbeta,
                             float* mean_dz, float* mean_dzy, float* zero_vec);
int launch_bn_bwd_finalize_coef_z(hipStream_t st, const double* part, int nparts, double count, int C, const float* bn_save, float* dgamma,
                                  float* dbeta, float* coef, float* zero_vec);
int launch_colsum(hipStream_t st, const float* X, int M, int C, double* part, float* out);

#define MROWS 64          // rows per workgroup of the streaming passes (== VROWS of bn.hip: bn_vec_parts)

// ---- streaming passes with per-channel activation parameters: a thread owns 4 consecutive channels and walks rows -------------------
// tpr = C / 4 threads cover a row (16-byte accesses), nrg = 256 / tpr rows per step; threads beyond nrg * tpr idle (C = 768: 64 of 256)
__global__ __launch_bounds__(256) void multi_act_fwd_kernel(const float* __restrict__ Y, float* __restrict__ Z, int M, int C, int rpb,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ chan, uint32_t thresh, float inv_keep,
                                                            uint64_t seed) {
    const int tid = threadIdx.x, tpr = C >> 2, nrg = 256 / tpr;
    const int cg = tid % tpr, rg = tid / tpr, c = cg * 4;
    if (rg >= nrg) return;
    const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
    const f32x4 sl = *(const f32x4*)(chan + c), dr = *(const f32x4*)(chan + C + c);
    const bool anyd = thresh && (dr[0] != 0.f || dr[1] != 0.f || dr[2] != 0.f || dr[3] != 0.f);
    const int r0 = blockIdx.x * rpb, r1 = min(M, r0 + rpb);
    for (int rb = r0 + rg; rb < r1; rb += 4 * nrg) {          // four rows in flight per thread
        f32x4 y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb + u * nrg;
            if (r < r1) y[u] = *(const f32x4*)(Y + (size_t)r * C + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb + u * nrg;
            if (r >= r1) break;
            const size_t i = (size_t)r * C + c;
            const uint32_t hq = anyd ? dropout_hash4(seed, i >> 2) : 0u;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = fmaf(y[u][e], sc[e], sh[e]);
                a = a > 0.f ? a : a * sl[e];
                if (thresh && dr[e] != 0.f) a = ((hq >> (8 * e)) & 255u) >= thresh ? a * inv_keep : 0.f;
                o[e] = a;
            }
            *(f32x4*)(Z + i) = o;
        }
    }
}

// gradient w.r.t. the BatchNorm output of element e of an aligned quad (hash hq of the quad)
__device__ __forceinline__ float multi_dz_prime(float dz, float y, float sc, float sh, float sl, float dr, uint32_t thresh, float inv_keep,
                                                uint32_t hq, int e) {
    if (thresh && dr != 0.f) dz = ((hq >> (8 * e)) & 255u) >= thresh ? dz * inv_keep : 0.f;
    const float a = fmaf(y, sc, sh);
    if (!(a > 0.f)) dz *= sl;
    return dz;
}

__global__ __launch_bounds__(256) void multi_bwd_reduce_kernel(const float* __restrict__ dZ, const float* __restrict__ Y, int M, int C,
                                                               const float* __restrict__ bn, const float* __restrict__ chan,
                                                               uint32_t thresh, float inv_keep, uint64_t seed, double* __restrict__ part,
                                                               int premasked) {
    // premasked: as in multi_bwd_apply_kernel (dZ already carries the activation derivative and the dropout mask)
    __shared__ double shd[256 * 8];
    const int tid = threadIdx.x, tpr = C >> 2, nrg = 256 / tpr;
    const int cg = tid % tpr, rg = tid / tpr, c = cg * 4;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (rg < nrg) {
        const f32x4 sc = *(const f32x4*)(bn + c), sh = *(const f32x4*)(bn + C + c);
        const f32x4 mu = *(const f32x4*)(bn + 2 * C + c), is = *(const f32x4*)(bn + 3 * C + c);
        const f32x4 sl = *(const f32x4*)(chan + c), dr = *(const f32x4*)(chan + C + c);
        const bool anyd = !premasked && thresh && (dr[0] != 0.f || dr[1] != 0.f || dr[2] != 0.f || dr[3] != 0.f);
        const int r0 = blockIdx.x * MROWS, r1 = min(M, r0 + MROWS);
        for (int rb = r0 + rg; rb < r1; rb += 4 * nrg) {      // four rows in flight per thread; summed in row order
            f32x4 y[4], dz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = rb + u * nrg;
                if (r < r1) { y[u] = *(const f32x4*)(Y + (size_t)r * C + c); dz[u] = *(const f32x4*)(dZ + (size_t)r * C + c); }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = rb + u * nrg;
                if (r >= r1) break;
                const size_t i = (size_t)r * C + c;
                const uint32_t hq = anyd ? dropout_hash4(seed, i >> 2) : 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = premasked ? dz[u][e] : multi_dz_prime(dz[u][e], y[u][e], sc[e], sh[e], sl[e], dr[e], thresh, inv_keep, hq, e);
                    s[e] += d; q[e] += (double)d * ((y[u][e] - mu[e]) * is[e]);
                }
            }
        }
    }
    // sum over the row groups, in group order; one partial row [2][C] per workgroup
#pragma unroll
    for (int e = 0; e < 4; ++e) { shd[tid * 8 + e] = s[e]; shd[tid * 8 + 4 + e] = q[e]; }
    __syncthreads();
    if (tid < tpr) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < nrg; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += shd[(g * tpr + tid) * 8 + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            part[((size_t)blockIdx.x * 2 + 0) * C + tid * 4 + e] = a[e];
            part[((size_t)blockIdx.x * 2 + 1) * C + tid * 4 + e] = a[4 + e];
        }
    }
}

// dY = scale * (dz' - mean_dz - yhat * mean_dzy)   (mean_dz == null: eval mode, dY = scale * dz')
__global__ __launch_bounds__(256) void multi_bwd_apply_kernel(const float* __restrict__ dZ, const float* __restrict__ Y, float* __restrict__ dY,
                                                              int M, int C, int rpb, const float* __restrict__ bn, const float* __restrict__ chan,
                                                              const float* __restrict__ mean_dz, const float* __restrict__ mean_dzy,
                                                              uint32_t thresh, float inv_keep, uint64_t seed, int premasked) {
    // premasked: dZ already carries the activation derivative and the dropout mask (the consumer's dgrad applied them where it produced
    // the gradient: gemm.hip gemm_out_bs / thin.hip): only the BatchNorm part is left
    const int tid = threadIdx.x, tpr = C >> 2, nrg = 256 / tpr;
    const int cg = tid % tpr, rg = tid / tpr, c = cg * 4;
    if (rg >= nrg) return;
    const f32x4 sc = *(const f32x4*)(bn + c), sh = *(const f32x4*)(bn + C + c);
    const f32x4 sl = *(const f32x4*)(chan + c), dr = *(const f32x4*)(chan + C + c);
    const bool anyd = !premasked && thresh && (dr[0] != 0.f || dr[1] != 0.f || dr[2] != 0.f || dr[3] != 0.f);
    f32x4 mu = {0, 0, 0, 0}, k1 = {0, 0, 0, 0}, k2 = {0, 0, 0, 0};
    if (mean_dz) {
        mu = *(const f32x4*)(bn + 2 * C + c);
        k1 = *(const f32x4*)(mean_dz + c);
        const f32x4 is = *(const f32x4*)(bn + 3 * C + c), mz = *(const f32x4*)(mean_dzy + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) k2[e] = is[e] * mz[e];
    }
    const int r0 = blockIdx.x * rpb, r1 = min(M, r0 + rpb);
    for (int rb = r0 + rg; rb < r1; rb += 4 * nrg) {          // four rows in flight per thread
        f32x4 y[4], dz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb + u * nrg;
            if (r < r1) { y[u] = *(const f32x4*)(Y + (size_t)r * C + c); dz[u] = *(const f32x4*)(dZ + (size_t)r * C + c); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb + u * nrg;
            if (r >= r1) break;
            const size_t i = (size_t)r * C + c;
            const uint32_t hq = anyd ? dropout_hash4(seed, i >> 2) : 0u;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d = premasked ? dz[u][e] : multi_dz_prime(dz[u][e], y[u][e], sc[e], sh[e], sl[e], dr[e], thresh, inv_keep, hq, e);
                d = d - k1[e] - (y[u][e] - mu[e]) * k2[e];
                o[e] = sc[e] * d;
            }
            *(f32x4*)(dY + i) = o;
        }
    }
}

// ---- the operand transform as a streaming pass (fallback of the chained layers for shapes the GEMM kernels do not transform) ---------
// out [M][C] contiguous = act(X * scale + shift) with the producer's dropout; X [M][..] row pitch ldx is a column slice (first column
// xf.col) of the producer's [M][xf.ld] matrix; xf.scale / xf.shift point at the slice's first channel.
__global__ __launch_bounds__(256) void xf_materialize_kernel(const float* __restrict__ X, int ldx, int M, int C, XfDev xf, float* __restrict__ out, int ldo, int vec) {
    if (vec) {
        const int cq = C >> 2;
        const size_t total = (size_t)M * cq;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const int r = (int)(i / cq), q = (int)(i - (size_t)r * cq);
            const f32x4 v = *(const f32x4*)(X + (size_t)r * ldx + 4 * q);
            const f32x4 sc = *(const f32x4*)(xf.scale + 4 * q), sh = *(const f32x4*)(xf.shift + 4 * q);
            const uint32_t qi = (uint32_t)(((uint64_t)r * xf.ld + xf.col + 4 * q) >> 2);
            *(f32x4*)(out + (size_t)r * ldo + 4 * q) = xf_apply_quad(v, sc, sh, xf.slope, xf.thresh, xf.inv_keep, xf.thresh ? mix32(qi ^ xf.xH) : 0u);
        }
    } else {
        const size_t total = (size_t)M * C;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const int r = (int)(i / C), c = (int)(i - (size_t)r * C);
            float a = fmaf(X[(size_t)r * ldx + c], xf.scale[c], xf.shift[c]);
            a = fmaxf(a, a * xf.slope);
            if (xf.thresh) {
                const uint64_t e = (uint64_t)r * xf.ld + xf.col + c;
                a = ((mix32((uint32_t)(e >> 2) ^ xf.xH) >> (8 * ((uint32_t)e & 3))) & 255u) >= xf.thresh ? a * xf.inv_keep : 0.f;
            }
            out[(size_t)r * ldo + c] = a;
        }
    }
}
int launch_xf_materialize_ld(hipStream_t st, const float* X, int ldx, int M, int C, const GemmXf& xf, float* out, int ldo) {
    if (!X || !out || M <= 0 || C <= 0 || ldo < C || (double)M * xf.ld >= 17179869184.0) return MLSP_ERR_ARG;
    const int vec = (C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && xf.ld % 4 == 0 && xf.col % 4 == 0 &&
                     (((uintptr_t)X | (uintptr_t)out | (uintptr_t)xf.scale | (uintptr_t)xf.shift) & 15) == 0) ? 1 : 0;
    const size_t work = vec ? (size_t)M * (C / 4) : (size_t)M * C;
    size_t blocks = (work + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(xf_materialize_kernel, dim3((unsigned)blocks), dim3(256), 0, st, X, ldx, M, C, xf_dev(xf), out, ldo, vec);
    return mlsp_launch_status();
}
int launch_xf_materialize(hipStream_t st, const float* X, int ldx, int M, int C, const GemmXf& xf, float* out) {
    return launch_xf_materialize_ld(st, X, ldx, M, C, xf, out, C);
}

// ---- host side --------------------------------------------------------------------------------------------------------------------
#define MCHECK(x) do { int _r = (x); if (_r != MLSP_OK) return _r; } while (0)

// deferred input of a segment (mlsp_defer_t, include/mlsp_hip.h): the transform a GEMM applies to that segment's operand
static bool multi_defer_ok(const mlsp_defer_t& d, const mlsp_seg_t& g) {
    return d.bn_save && d.ld > 0 && d.col == g.x_col && d.col + g.Cin <= d.ld && d.p_drop >= 0.f && d.p_drop < 1.f && d.act >= 0 && d.act <= 2;
}
static GemmXf multi_xf(const mlsp_defer_t& in, int which, int batch_stats) {
    GemmXf x;
    x.scale = in.bn_save + in.col; x.shift = in.bn_save + in.ld + in.col; x.act = in.act; x.slope = in.slope; x.thresh = dropout_thresh8(in.p_drop);
    x.inv_keep = dropout_inv_keep8(in.p_drop); x.seed = in.seed; x.ld = in.ld; x.col = in.col; x.which = which;
    if (batch_stats) { x.mean = in.bn_save + 2 * in.ld + in.col; x.invstd = in.bn_save + 3 * in.ld + in.col; }     // (api.hip chain_xf)
    return x;
}
static bool multi_defer_same(const mlsp_defer_t& a, const mlsp_defer_t& b) {
    return a.bn_save == b.bn_save && a.ld == b.ld && a.act == b.act && a.slope == b.slope && a.p_drop == b.p_drop && a.seed == b.seed;
}
static bool multi_defer_fusable(const mlsp_defer_t& in) { return !(in.act == 2 && !(in.slope >= 0.f && in.slope <= 1.f)); }

// rows per workgroup of the element-wise passes: enough workgroups for ~8 per CU (a thread keeps four rows in flight)
static int multi_rows_per_block(int M, int C) {
    const int nrg = 256 / (C / 4);
    int rpb = 4 * nrg;                                         // one unrolled step
    while (rpb < 64 && (long)(M + rpb - 1) / rpb > 4096) rpb *= 2;
    return rpb;
}

static int multi_check(const float* X, int ldx, int M, const mlsp_seg_t* segs, int nseg, int& Ctot, int& xw) {
    if (!X || !segs || nseg < 1 || nseg > 8 || M <= 32) return MLSP_ERR_ARG;
    Ctot = 0; xw = 0;
    for (int s = 0; s < nseg; ++s) {
        const mlsp_seg_t& g = segs[s];
        if (!g.W || g.Cin <= 0 || g.Cout <= 0 || g.ldw < g.Cin || g.x_col < 0 || g.x_col + g.Cin > ldx) return MLSP_ERR_ARG;
        if (g.Cout % 4 || g.x_col % 4 || g.Cin % 4) return MLSP_ERR_UNSUPPORTED;          // 16-byte column slices
        Ctot += g.Cout;
        if (g.x_col + g.Cin > xw) xw = g.x_col + g.Cin;
    }
    if (Ctot > 1024) return MLSP_ERR_UNSUPPORTED;
    return MLSP_OK;
}

// Length of the run of segments starting at s that ONE block-diagonal GEMM launch can take (gemm.hip GemmGroups): same shape and
// weight pitch, no bias, input slices back to back, widths a multiple of the 128-wide tile, interior 16-byte-aligned operands.
static int multi_group_run(const float* X, int ldx, int M, const mlsp_seg_t* segs, int nseg, int s, const mlsp_defer_t* in = nullptr) {
    static const bool off = getenv("MLSP_NO_GROUPED_GEMM") != nullptr;          // read-once A/B switch
    const mlsp_seg_t& a = segs[s];
    if (off || gemm_precision_mode() == 1 || a.bias || a.Cin % 128 || a.Cout % 128 || M % 128 || ldx % 4 || a.ldw % 4 || (((uintptr_t)X | (uintptr_t)a.W) & 15)) return 1;
    int n = 1;
    while (s + n < nseg && n < 4) {
        const mlsp_seg_t& b = segs[s + n];
        if (b.bias || b.Cin != a.Cin || b.Cout != a.Cout || b.ldw != a.ldw || b.x_col != a.x_col + n * a.Cin || ((uintptr_t)b.W & 15)) break;
        if (in && !multi_defer_same(in[s], in[s + n])) break;                   // one transform per launch
        ++n;
    }
    return n;
}

extern "C" {

int mlsp_multimlp_supported(int M, const mlsp_seg_t* segs, int nseg, int precision) {
    if (precision < 0 || precision > 3) return 0;
    GemmPrecisionScope prec_scope_(precision);
    int Ctot = 0, xw = 0;
    if (!segs || nseg < 1) return 0;
    int ldx = 0;
    for (int s = 0; s < nseg; ++s) ldx = segs[s].x_col + segs[s].Cin > ldx ? segs[s].x_col + segs[s].Cin : ldx;
    static const float dummy = 0.f;
    if (multi_check(&dummy, ldx, M, segs, nseg, Ctot, xw) != MLSP_OK) return 0;
    return 1;
}

int mlsp_multimlp_fwd_f32(const float* X, int ldx, int M, const mlsp_seg_t* segs, int nseg, const mlsp_defer_t* in, const float* gamma, const float* beta,
                          float* run_mean, float* run_var, float momentum, float eps, int training, const float* chan, float p_drop,
                          uint64_t seed, float* Y, float* Z, float* bn_save, int precision, void* ws, size_t ws_bytes, mlsp_stream_t st) {
    PREC_SCOPE(precision);
    int Ctot, xw;
    MCHECK(multi_check(X, ldx, M, segs, nseg, Ctot, xw));
    if (!gamma || !beta || !chan || !Y || !bn_save || p_drop < 0.f || p_drop >= 1.f) return MLSP_ERR_ARG;      // Z == NULL: activation deferred to the consumers
    if (!training && (!run_mean || !run_var)) return MLSP_ERR_ARG;
    if (!mlsp_multimlp_supported(M, segs, nseg, precision)) return MLSP_ERR_UNSUPPORTED;
    if ((((uintptr_t)Y | (uintptr_t)Z | (uintptr_t)chan | (uintptr_t)bn_save) & 15) != 0) return MLSP_ERR_UNSUPPORTED;
    if (in) for (int s = 0; s < nseg; ++s) if (!multi_defer_ok(in[s], segs[s])) return MLSP_ERR_ARG;
    Workspace w(ws, ws_bytes);
    // BatchNorm sums out of the GEMM epilogues when every segment uses the same row-panel height (their partial rows then line up)
    // launches: runs of identical region-head segments go out as ONE block-diagonal GEMM (twice the tiles per launch: the second
    // generation of tiles covers the first one's output pass), the others one by one
    int run[8];
    bool fuse[8] = {false, false, false, false, false, false, false, false};    // deferred input: the run's GEMM transforms its A operand itself
    bool any_mat = false;
    for (int s = 0; s < nseg; s += run[s]) {
        run[s] = multi_group_run(X, ldx, M, segs, nseg, s, in);
        if (in) {
            const mlsp_seg_t& g = segs[s];
            if (run[s] > 1 && !gemm_xf_on_split(false, true, M, run[s] * g.Cout, g.Cin, 1)) run[s] = 1;
            fuse[s] = multi_defer_fusable(in[s]) && gemm_xf_supported(false, true, M, run[s] * g.Cout, g.Cin, X + g.x_col, ldx, g.W, g.ldw, 1);
            any_mat |= !fuse[s];
        }
        for (int t = 1; t < run[s]; ++t) run[s + t] = 0;
    }
    bool fused = training != 0;
    const int bm = gemm_panel_rows(M, run[0] * segs[0].Cout, segs[0].Cin);
    for (int s = 0; s < nseg && fused; s += run[s])
        fused = gemm_stat_parts(M, run[s] * segs[s].Cout, segs[s].Cin) > 0 && gemm_panel_rows(M, run[s] * segs[s].Cout, segs[s].Cin) == bm;
    int nparts = fused ? (M + bm - 1) / bm : bn_vec_parts(M);
    const int npmax = nparts > bn_vec_parts(M) ? nparts : bn_vec_parts(M);
    double* part = w.take<double>((size_t)(npmax > (M + 255) / 256 ? npmax : (M + 255) / 256) * 2 * Ctot);
    size_t sf = 0;                                             // (small M: a segment's GEMM may split K; then the statistics are a separate pass)
    for (int s = 0; s < nseg; s += run[s]) { const size_t f = gemm_slab_floats(M, run[s] * segs[s].Cout, segs[s].Cin); sf = f > sf ? f : sf; }
    float* slab = sf ? w.take<float>(sf) : nullptr;
    float* Xa = any_mat ? w.take<float>((size_t)M * ldx) : nullptr;     // activated copies of the slices whose GEMM cannot transform (same layout as X)
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    int ycol = 0;
    for (int s = 0; s < nseg; s += run[s]) {
        const mlsp_seg_t& g = segs[s];
        GemmGroups grp = {run[s], 1, g.Cin, 0, {nullptr, nullptr, nullptr, nullptr}};
        for (int t = 0; t < run[s]; ++t) grp.Bg[t] = segs[s + t].W;
        const float* Xs = X;
        GemmXf xf_s; const GemmXf* xf = nullptr;
        if (in) {
            xf_s = multi_xf(in[s], 1, training);
            if (fuse[s]) xf = &xf_s;
            else {
                for (int t = 0; t < run[s]; ++t) {              // (a run without a fused transform: slice by slice into the activated copy)
                    const GemmXf xt = multi_xf(in[s + t], 1, training);
                    MCHECK(launch_xf_materialize_ld(st, X + segs[s + t].x_col, ldx, M, segs[s + t].Cin, xt, Xa + segs[s + t].x_col, ldx));
                }
                Xs = Xa;
            }
        }
        MCHECK(launch_gemm(st, false, true, M, run[s] * g.Cout, g.Cin, Xs + g.x_col, ldx, g.W, g.ldw, Y + ycol, Ctot, g.bias, nullptr, 0, slab, sf,
                           fused ? part + ycol : nullptr, nullptr, nullptr, nullptr, false, xf, Ctot, run[s] > 1 ? &grp : nullptr));
        ycol += run[s] * g.Cout;
    }
    float* scale = bn_save, *shift = bn_save + Ctot, *mean = bn_save + 2 * Ctot, *invstd = bn_save + 3 * Ctot;
    if (training) {
        if (!fused) MCHECK(launch_colstats_n(st, Y, M, Ctot, Ctot, part, &nparts));
        MCHECK(launch_bn_finalize(st, part, nparts, (double)M, Ctot, gamma, beta, run_mean, run_var, momentum, eps, scale, shift, mean, invstd));
    } else {
        MCHECK(launch_bn_eval_prepare(st, Ctot, gamma, beta, run_mean, run_var, eps, scale, shift, mean, invstd));
    }
    if (!Z) return MLSP_OK;                                    // deferred: the consumers apply scale / shift / activation / dropout themselves
    const float pd = training ? p_drop : 0.f;
    const int rpb = multi_rows_per_block(M, Ctot);
    hipLaunchKernelGGL(multi_act_fwd_kernel, dim3((M + rpb - 1) / rpb), dim3(256), 0, st, Y, Z, M, Ctot, rpb, scale, shift, chan,
                       dropout_thresh8(pd), dropout_inv_keep8(pd), seed);
    return mlsp_launch_status();
}

// Row panels the fused statistics pass of mlsp_multimlp_bwd_f32(in_stats != NULL) writes: M / 128 when EVERY segment's dgrad (single or
// block-diagonal, as the backward will launch them) runs on a kernel with that pass, else 0.
int mlsp_multimlp_bwd_stats_parts(int M, const mlsp_seg_t* segs, int nseg, int ldx, int lddx, int precision) {
    if (precision < 0 || precision > 3 || !segs || nseg < 1 || nseg > 8) return 0;
    GemmPrecisionScope prec_scope_(precision);
    int Ctot = 0;
    for (int s = 0; s < nseg; ++s) Ctot += segs[s].Cout;
    static const float dummy = 0.f;
    int run = 1;
    for (int s = 0; s < nseg; s += run) {
        run = multi_group_run(&dummy, ldx, M, segs, nseg, s);
        for (int t = 0; t < s; ++t)
            if (!(segs[t].x_col + segs[t].Cin <= segs[s].x_col || segs[s].x_col + run * segs[s].Cin <= segs[t].x_col)) return 0;     // overlapping inputs: beta = 1
        // (the backward may launch the run as one block-diagonal product or segment by segment: both shapes must take the pass)
        if (gemm_bs_parts(M, run * segs[s].Cin, segs[s].Cout, Ctot, segs[s].ldw, lddx) != M / 128 ||
            gemm_bs_parts(M, segs[s].Cin, segs[s].Cout, Ctot, segs[s].ldw, lddx) != M / 128) return 0;
    }
    return M / 128;
}

int mlsp_multimlp_bwd_f32(const float* dZ, const float* X, int ldx, int M, const mlsp_seg_t* segs, int nseg, const mlsp_defer_t* in, const float* Y,
                          const float* bn_save, int training, const float* chan, float p_drop, uint64_t seed, float* dX, int lddx,
                          float* const* dW, float* dbias, float* dgamma, float* dbeta, double* in_stats, const double* pre_stats, int pre_parts,
                          int precision, void* ws, size_t ws_bytes, mlsp_stream_t st) {
    // in_stats (consumer role, needs `in`): every segment's input gradient is stored MASKED by its producer's activation derivative /
    // dropout and the producer's BatchNorm-backward column sums are left in in_stats [M / 128][2][in->ld] (gemm.hip gemm_out_bs).
    // pre_stats (producer role): dZ arrives masked with its sums in pre_stats [pre_parts][2][Ctot]: no reduction pass here.
    // pre_stats == NULL with pre_parts < 0: dZ arrives masked but WITHOUT complete sums (only some of the consumers took part in this
    // backward pass; the other columns are zero): this call reduces them itself and does not apply the mask a second time.
    PREC_SCOPE(precision);
    int Ctot, xw;
    MCHECK(multi_check(X, ldx, M, segs, nseg, Ctot, xw));
    if (!dZ || !Y || !bn_save || !chan || !dW || !dgamma || !dbeta || (dX && lddx < xw)) return MLSP_ERR_ARG;
    if (in_stats && (!in || !dX)) return MLSP_ERR_ARG;
    if (pre_stats && pre_parts <= 0) return MLSP_ERR_ARG;
    const int premasked = (pre_stats || pre_parts < 0) ? 1 : 0;
    if (!mlsp_multimlp_supported(M, segs, nseg, precision)) return MLSP_ERR_UNSUPPORTED;
    if ((((uintptr_t)Y | (uintptr_t)dZ | (uintptr_t)chan | (uintptr_t)bn_save) & 15) != 0) return MLSP_ERR_UNSUPPORTED;
    if (in) for (int s = 0; s < nseg; ++s) if (!multi_defer_ok(in[s], segs[s])) return MLSP_ERR_ARG;
    Workspace w(ws, ws_bytes);
    float* dY = w.take<float>((size_t)M * Ctot);
    const int nparts = bn_vec_parts(M);
    double* part = w.take<double>((size_t)nparts * 2 * Ctot);
    float* mean_dz = w.take<float>(Ctot);
    float* mean_dzy = w.take<float>(Ctot);
    int run[8];                                                 // block-diagonal launches: as in the forward; the wgrad needs the run's dW back to back
    bool fuse[8] = {false, false, false, false, false, false, false, false};    // deferred input: the run's weight-gradient GEMM transforms its B operand (X) itself
    bool any_mat = false;
    for (int s = 0; s < nseg; s += run[s]) {
        run[s] = multi_group_run(X, ldx, M, segs, nseg, s, in);
        for (int t = 1; t < run[s]; ++t)
            if (!dW[s + t] || dW[s + t] != dW[s] + (size_t)t * segs[s].Cout * segs[s].Cin) { run[s] = 1; break; }
        if (in) {
            const mlsp_seg_t& g = segs[s];
            if (run[s] > 1 && !gemm_xf_on_split(true, false, run[s] * g.Cout, g.Cin, M, 2)) run[s] = 1;
            fuse[s] = multi_defer_fusable(in[s]) && gemm_xf_supported(true, false, run[s] * g.Cout, g.Cin, M, dY, Ctot, X + g.x_col, ldx, 2);
            any_mat |= !fuse[s];
        }
        for (int t = 1; t < run[s]; ++t) run[s + t] = 0;
    }
    size_t sf = 0;
    for (int s = 0; s < nseg; s += run[s]) {
        const size_t f = gemm_slab_floats(run[s] * segs[s].Cout, segs[s].Cin, M), f2 = dX ? gemm_slab_floats(M, run[s] * segs[s].Cin, segs[s].Cout) : 0;
        sf = f > sf ? f : sf; sf = f2 > sf ? f2 : sf;
    }
    float* slab = sf ? w.take<float>(sf) : nullptr;
    float* Xa = any_mat ? w.take<float>((size_t)M * ldx) : nullptr;
    float* coef = w.take<float>((size_t)4 * Ctot);           // c0 | nk2 | sc | max |d'| (bn.hip bn_bwd_finalize_coef_kernel)
    if (!w.ok()) return MLSP_ERR_WORKSPACE;
    const float pd = training ? p_drop : 0.f;
    const uint32_t th = dropout_thresh8(pd);
    const float ik = dropout_inv_keep8(pd);
    // masked gradient + its sums in hand: when every dgrad / weight-gradient launch of this layer can form dY = (d' + y * nk2 + c0) * sc in
    // its A operand loads (gemm.hip gemm_split_kernel<.., DY>) the apply pass and the dY tensor are skipped altogether
    static const bool dy_off = getenv("MLSP_BWD_DY_OFF") != nullptr;        // read-once A/B switch (tools/ab)
    bool use_dy = pre_stats && training && !dy_off && !any_mat;
    for (int s = 0, yc = 0; s < nseg && use_dy; yc += run[s] * segs[s].Cout, s += run[s]) {
        const mlsp_seg_t& g = segs[s];
        if (dX) use_dy = gemm_dy_supported(false, false, M, run[s] * g.Cin, g.Cout, dZ + yc, Ctot, g.W, g.ldw);
        use_dy = use_dy && gemm_dy_supported(true, false, run[s] * g.Cout, g.Cin, M, dZ + yc, Ctot, X + g.x_col, ldx);
    }
    float* zb = (dbias && training) ? dbias : nullptr;      // a bias in front of a batch-statistics BatchNorm: analytically zero gradient, written by the finalizer
    if (use_dy) {
        MCHECK(launch_bn_bwd_finalize_coef_z(st, pre_stats, pre_parts, (double)M, Ctot, bn_save, dgamma, dbeta, coef, zb));
    } else if (pre_stats) {
        MCHECK(launch_bn_bwd_finalize_z(st, pre_stats, pre_parts, (double)M, Ctot, dgamma, dbeta, mean_dz, mean_dzy, zb));
    } else {
        hipLaunchKernelGGL(multi_bwd_reduce_kernel, dim3(nparts), dim3(256), 0, st, dZ, Y, M, Ctot, bn_save, chan, th, ik, seed, part, premasked);
        MCHECK(launch_bn_bwd_finalize_z(st, part, nparts, (double)M, Ctot, dgamma, dbeta, mean_dz, mean_dzy, zb));
    }
    const int rpb = multi_rows_per_block(M, Ctot);
    if (!use_dy) {
        hipLaunchKernelGGL(multi_bwd_apply_kernel, dim3((M + rpb - 1) / rpb), dim3(256), 0, st, dZ, Y, dY, M, Ctot, rpb, bn_save, chan,
                           training ? mean_dz : (const float*)nullptr, mean_dzy, th, ik, seed, premasked);
        MCHECK(mlsp_launch_status());
    }
    const float* gY = use_dy ? dZ : dY;                           // the GEMMs' A operand: d' (+ y, coefficients) or the formed dY
    int ycol = 0;
    for (int s = 0; s < nseg; s += run[s]) {
        const mlsp_seg_t& g = segs[s];
        const int G = run[s];
        for (int t = 0; t < G; ++t) if (!dW[s + t]) return MLSP_ERR_ARG;
        GemmGroups grp = {G, 1, g.Cout, 0, {nullptr, nullptr, nullptr, nullptr}};
        for (int t = 0; t < G; ++t) grp.Bg[t] = segs[s + t].W;
        const GemmDy dy_s = {Y + ycol, coef + ycol, Ctot, 1};
        if (dX) {
            bool acc = false;                                  // an earlier segment on the same input columns: add (partial overlaps: rejected)
            for (int t = 0; t < s; ++t) {
                const bool same = segs[t].x_col == g.x_col && segs[t].Cin == g.Cin;
                const bool apart = segs[t].x_col + segs[t].Cin <= g.x_col || g.x_col + G * g.Cin <= segs[t].x_col;
                if (!same && !apart) return MLSP_ERR_UNSUPPORTED;
                acc |= same;
            }
            if (acc && G > 1) return MLSP_ERR_UNSUPPORTED;
            GemmBs bs_s; const GemmBs* bs = nullptr;
            if (in_stats) {
                const mlsp_defer_t& d = in[s];
                bs_s = {X + g.x_col, ldx, d.bn_save + d.col, d.ld, d.act, d.slope, dropout_thresh8(d.p_drop), dropout_inv_keep8(d.p_drop), d.seed,
                        d.ld, d.col, in_stats + d.col, d.ld};
                bs_s.amax = (float*)(in_stats + (size_t)(M / 128) * 2 * d.ld) + d.col;       // the maxima plane behind the [M / 128][2][ld] sums
                bs = &bs_s;
                for (int t = 1; t < G; ++t) if (!multi_defer_same(in[s], in[s + t])) return MLSP_ERR_UNSUPPORTED;     // (one producer description per launch)
            }
            MCHECK(launch_gemm(st, false, false, M, G * g.Cin, g.Cout, gY + ycol, Ctot, g.W, g.ldw, dX + g.x_col, lddx, nullptr, nullptr, 0, slab, sf,
                               nullptr, nullptr, nullptr, nullptr, acc, nullptr, 0, G > 1 ? &grp : nullptr, bs, use_dy ? &dy_s : nullptr));
        }
        const float* Xs = X;
        GemmXf xf_s; const GemmXf* xf = nullptr;
        if (in) {
            xf_s = multi_xf(in[s], 2, training);
            if (fuse[s]) xf = &xf_s;
            else {
                for (int t = 0; t < G; ++t) {
                    const GemmXf xt = multi_xf(in[s + t], 2, training);
                    MCHECK(launch_xf_materialize_ld(st, X + segs[s + t].x_col, ldx, M, segs[s + t].Cin, xt, Xa + segs[s + t].x_col, ldx));
                }
                Xs = Xa;
            }
        }
        GemmGroups grw = {G, 2, 0, g.Cin, {nullptr, nullptr, nullptr, nullptr}};
        MCHECK(launch_gemm(st, true, false, G * g.Cout, g.Cin, M, gY + ycol, Ctot, Xs + g.x_col, ldx, dW[s], g.Cin, nullptr, nullptr, 0, slab, sf,
                           nullptr, nullptr, nullptr, nullptr, false, xf, 0, G > 1 ? &grw : nullptr, nullptr, use_dy ? &dy_s : nullptr));
        ycol += G * g.Cout;
    }
    if (dbias && !training) MCHECK(launch_colsum(st, dY, M, Ctot, part, dbias));      // (training: zeroed by the finalizer above)
    return MLSP_OK;
}

}  // extern "C"
