// Per-point 1x1 conv + BatchNorm + activation followed by a max over the N points of each cloud:
//   conv5 + bn5 + LeakyReLU + adaptive_max_pool1d        PointDA/Models.py:132-136
//   T-Net conv2d3 + torch.max(dim=2)                      PointDA/model_utils.py:116-117
// out[b][c] = max_n act(BN(Y[b*N+n][c])),  Y = X W^T.
//
// Forward: act(BN(.)) is monotone per channel, so the max over n is taken on Y itself (max for scale >= 0, min
// otherwise) -- the activated [P,Cout] tensor is never written.
// Backward: only B*Cout entries of the activated tensor carry an incoming gradient, and BatchNorm's backward
// is affine in Y = X W^T, so the dense [P,Cout] gradient never has to exist:
//     dY[p][c] = g_bc [p == arg_bc] - A_c - Bc_c (Y[p][c] - mean_c)
//     dW = S - A (x) sum_x - diag(Bc) (W G - mean (x) sum_x),   G = X^T X   (Gram matrix, one [Cin,Cin] GEMM over P)
//     dX = X (-M) - 1 (x) r + rows scattered from g,            M = W^T diag(Bc) W,  r = W^T (A - Bc*mean)
// i.e. two [Cin x Cin]-sized contractions over the points instead of two [Cout x Cin]-sized ones (conv5: 2x fewer
// FLOPs, conv3: 8x), no BN streaming passes over [P,Cout], no dense scatter target.
#include "common.h"
#include <math.h>

// column extreme over the N rows of each cloud: sel = max (gamma >= 0) or min; arg = first row attaining it
__global__ __launch_bounds__(256) void colsel_kernel(const float* __restrict__ Y, const float* __restrict__ gamma, int N, int C,
                                                     float* __restrict__ ysel, int* __restrict__ arg) {
    __shared__ float sv[4][64];
    __shared__ int si[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, b = blockIdx.y;
    const bool use_max = c < C ? gamma[c] >= 0.f : true;
    float best = use_max ? -INFINITY : INFINITY;
    int bi = 0x7fffffff;
    if (c < C)
        for (int n = w; n < N; n += 4) {
            float v = Y[((size_t)b * N + n) * C + c];
            bool take = use_max ? (v > best) : (v < best);
            if (take) { best = v; bi = n; }
        }
    sv[w][lane] = best; si[w][lane] = bi;
    __syncthreads();
    if (w == 0 && c < C) {
        for (int u = 1; u < 4; ++u) {
            float v = sv[u][lane]; int i = si[u][lane];
            bool better = use_max ? (v > best) : (v < best);
            if (better || (v == best && i < bi)) { best = v; bi = i; }
        }
        ysel[(size_t)b * C + c] = best;
        arg[(size_t)b * C + c] = bi == 0x7fffffff ? 0 : bi;
    }
}

// reduce the per-panel extremes written by the GEMM epilogue over the N/128 panels of each cloud
// (bn != null: the BatchNorm + activation of the selected value -- colsel_out_kernel's arithmetic -- in the same pass)
__global__ void colsel_panels_kernel(const float* __restrict__ pv, const int* __restrict__ pr, const float* __restrict__ gamma,
                                     int B, int N, int C, int panels_per_cloud, float* __restrict__ ysel, int* __restrict__ arg,
                                     const float* __restrict__ bn, int act, float slope, float* __restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    int b = t / C, c = t % C;
    const bool use_max = gamma[c] >= 0.f;
    float best = use_max ? -INFINITY : INFINITY;
    int brow = 0x7fffffff;
    for (int u = 0; u < panels_per_cloud; ++u) {
        size_t q = (size_t)(b * panels_per_cloud + u) * C + c;
        float v = pv[q]; int r = pr[q];
        bool better = use_max ? (v > best) : (v < best);
        if (better || (v == best && r < brow)) { best = v; brow = r; }
    }
    ysel[t] = best;
    arg[t] = brow - b * N;
    if (bn) out[t] = lrelu_or_relu(fmaf(best, bn[c], bn[C + c]), act, slope);
}

// out = act(scale*ysel + shift)      [B][C]
__global__ void colsel_out_kernel(const float* __restrict__ ysel, const float* __restrict__ bn, int total, int C, int act,
                                  float slope, float* __restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int c = t % C;
    out[t] = lrelu_or_relu(fmaf(ysel[t], bn[c], bn[C + c]), act, slope);
}

// Backward coefficients, one block per channel c (B is small):
//   dz_bc = dOut*act'(out); dgamma = sum_b dz*yhat_sel; dbeta = sum_b dz; g_bc = scale*dz
//   coef[0][c] = A = scale*mean_dz; coef[1][c] = Bc = scale*invstd*mean_dzy; coef[2][c] = A - Bc*mean; coef[3][c] = -Bc
//   (training == 0: A = Bc = 0)
__global__ __launch_bounds__(64) void colmax_bwd_coef_kernel(const float* __restrict__ dOut, const float* __restrict__ out,
                                                             const float* __restrict__ ysel, const float* __restrict__ bn,
                                                             int B, int C, double count, int act, float slope, int training,
                                                             float* __restrict__ g, float* __restrict__ coef,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const float sc = bn[c], mu = bn[2 * C + c], is = bn[3 * C + c];
    double s = 0.0, q = 0.0;
    for (int b = lane; b < B; b += 64) {
        size_t t = (size_t)b * C + c;
        float d = dOut[t];
        if (act && !(out[t] > 0.f)) d *= (act == 1 ? 0.f : slope);
        g[t] = sc * d;
        s += d; q += (double)d * ((ysel[t] - mu) * is);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (lane == 0) {
        dbeta[c] = (float)s; dgamma[c] = (float)q;
        float A = training ? sc * (float)(s / count) : 0.f;
        float Bc = training ? sc * is * (float)(q / count) : 0.f;
        coef[c] = A; coef[C + c] = Bc; coef[2 * C + c] = A - Bc * mu; coef[3 * C + c] = -Bc;
    }
}

// Wb[c][:] = rowscale[c] * W[c][:]
__global__ void scale_rows_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ rowscale, int Cout, int Cin,
                                  float* __restrict__ Wb) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Cout * Cin) return;
    int c = t / Cin, i = t % Cin;
    Wb[t] = rowscale[c] * W[(size_t)c * ldw + i];
}

// negr[i] = - sum_c v[c] * W[c][i]        block: 64 columns x 16 row groups (one wave each), LDS reduce in fixed order
// (Wb != null: the workgroups past the first ceil(Cin / 64) are scale_rows_kernel's -- Wb[c][:] = rowscale[c] * W[c][:] -- so that the two
// weight-sized preparations of the colmax backward's input-gradient half are one launch)
__global__ __launch_bounds__(1024) void wt_vec_neg_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ v, int Cout,
                                                          int Cin, float* __restrict__ negr, const float* __restrict__ rowscale,
                                                          float* __restrict__ Wb) {
    __shared__ float red[16][64];
    const int nvb = (Cin + 63) / 64;
    if ((int)blockIdx.x >= nvb) {
        const int t = ((int)blockIdx.x - nvb) * 1024 + threadIdx.x;
        if (t < Cout * Cin) { const int c = t / Cin, i = t % Cin; Wb[t] = rowscale[c] * W[(size_t)c * ldw + i]; }
        return;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (i < Cin) {
        int c = w;
        for (; c + 16 * 7 < Cout; c += 16 * 8) {          // eight independent loads in flight per lane (the chain itself is fixed-order)
            float wv[8], vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { wv[u] = W[(size_t)(c + 16 * u) * ldw + i]; vv[u] = v[c + 16 * u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(vv[u], wv[u], acc);
        }
        for (; c < Cout; c += 16) acc = fmaf(v[c], W[(size_t)c * ldw + i], acc);
    }
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0 && i < Cin) {
        float s = 0.f;
        for (int u = 0; u < 16; ++u) s += red[u][lane];
        negr[i] = -s;
    }
}

// S[c][:] = sum_b g[b][c] * X[b*N + arg[b][c]][:]       block = (c, 256-wide chunk of Cin), fixed order over b
__global__ __launch_bounds__(256) void colmax_gather_rows_kernel(const float* __restrict__ g, const int* __restrict__ arg,
                                                                 const float* __restrict__ X, int ldx, int B, int N, int Cout,
                                                                 int Cin, float* __restrict__ S) {
    const int c = blockIdx.x;
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= Cin) return;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const size_t t = (size_t)b * Cout + c;
        acc = fmaf(g[t], X[((size_t)b * N + arg[t]) * ldx + i], acc);
    }
    S[(size_t)c * Cin + i] = acc;
}

// dW = S - A (x) sx - Bc (WG - mean (x) sx)
__global__ void colmax_dw_kernel(const float* __restrict__ S, const float* __restrict__ WG, const float* __restrict__ sx,
                                 const float* __restrict__ coef, const float* __restrict__ bn, int Cout, int Cin,
                                 float* __restrict__ dW) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Cout * Cin) return;
    int c = t / Cin, i = t % Cin;
    float A = coef[c], Bc = coef[Cout + c], mu = bn[2 * Cout + c];
    dW[t] = S[t] - A * sx[i] - Bc * (WG[t] - mu * sx[i]);
}

// dX[b*N + p][:] += sum_{c : arg[b][c] == p} g[b][c] * W[c][:]
// CSR_SPLIT workgroups per cloud; each sorts the cloud's Cout (row, channel) pairs by row in LDS (lists ascending in
// channel -> fixed summation order) and accumulates its slice of the rows.  The arg-max of many channels lands on the same
// few "critical" points, so list lengths are heavily skewed: rows with up to CSR_LONG entries are handled one per wave,
// longer ones by all waves of the workgroup together (entries strided over the waves, partial rows added in wave order).
#define CSR_SPLIT 16
#define CSR_LONG 8
__global__ __launch_bounds__(1024) void colmax_scatter_rows_kernel(const float* __restrict__ g, const int* __restrict__ arg,
                                                                  const float* __restrict__ W, int ldw, int N, int Cout, int Cin,
                                                                  float* __restrict__ dX, int lddx) {
    extern __shared__ int csm[];
    int* cnt = csm;                 // [N]
    int* off = csm + N;             // [N+1]
    int* lst = off + N + 1;         // [Cout] channels sorted by (row, channel)
    int* ab = lst + Cout;           // [Cout] arg of this cloud
    int* tmp = ab + Cout;           // [Cout] channels grouped by row, unsorted inside a row
    float* red = (float*)(tmp + Cout);   // [16 waves][256] partial row slices of the cooperative path
    const int b = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    for (int p = tid; p < N; p += nt) cnt[p] = 0;
    for (int c = tid; c < Cout; c += nt) ab[c] = arg[(size_t)b * Cout + c];
    __syncthreads();
    for (int c = tid; c < Cout; c += nt) atomicAdd(&cnt[ab[c]], 1);
    __syncthreads();
    if (tid < 64) {                 // exclusive scan of cnt by one wave
        int chunk = (N + 63) / 64, beg = tid * chunk, end = min(N, beg + chunk), s = 0;
        for (int p = beg; p < end; ++p) s += cnt[p];
        int incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (tid >= o) incl += t; }
        int run = incl - s;
        for (int p = beg; p < end; ++p) { off[p] = run; run += cnt[p]; }
        if (tid == 63) off[N] = incl;
    }
    __syncthreads();
    for (int p = tid; p < N; p += nt) cnt[p] = 0;
    __syncthreads();
    for (int c = tid; c < Cout; c += nt) {      // group by row (arrival order), then rank inside the row by channel
        const int p = ab[c];
        tmp[off[p] + atomicAdd(&cnt[p], 1)] = c;
    }
    __syncthreads();
    for (int c = tid; c < Cout; c += nt) {
        const int p = ab[c], e0 = off[p], e1 = off[p + 1];
        int rank = 0;
        for (int e = e0; e < e1; ++e) rank += (tmp[e] < c);
        lst[e0 + rank] = c;
    }
    __syncthreads();
    const int rows_per_part = (N + CSR_SPLIT - 1) / CSR_SPLIT;
    const int p0 = part * rows_per_part, p1 = min(N, p0 + rows_per_part);
    const float* gb = g + (size_t)b * Cout;
    // short lists: one wave per row
    for (int p = p0 + wave; p < p1; p += nw) {
        const int e0 = off[p], e1 = off[p + 1];
        if (e0 == e1 || e1 - e0 > CSR_LONG) continue;
        float* o = dX + ((size_t)b * N + p) * lddx;
        for (int i0 = 0; i0 < Cin; i0 += 256) {
            float acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                 // the read of the read-modify-write goes out with the W loads
                int i = i0 + lane + 64 * u;
                acc[u] = i < Cin ? o[i] : 0.f;
            }
            for (int e = e0; e < e1; ++e) {
                const int c = lst[e];
                const float gv = gb[c];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int i = i0 + lane + 64 * u;
                    if (i < Cin) acc[u] = fmaf(gv, W[(size_t)c * ldw + i], acc[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int i = i0 + lane + 64 * u;
                if (i < Cin) o[i] = acc[u];
            }
        }
    }
    // long lists: the whole workgroup per row (control flow is uniform: p and the list bounds come from LDS)
    for (int p = p0; p < p1; ++p) {
        const int e0 = off[p], e1 = off[p + 1];
        if (e1 - e0 <= CSR_LONG) continue;
        float* o = dX + ((size_t)b * N + p) * lddx;
        for (int i0 = 0; i0 < Cin; i0 += 256) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int e = e0 + wave; e < e1; e += nw) {
                const int c = lst[e];
                const float gv = gb[c];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int i = i0 + lane + 64 * u;
                    if (i < Cin) acc[u] = fmaf(gv, W[(size_t)c * ldw + i], acc[u]);
                }
            }
            __syncthreads();                               // previous slice's readers are done with red
#pragma unroll
            for (int u = 0; u < 4; ++u) red[wave * 256 + lane + 64 * u] = acc[u];
            __syncthreads();
            if (tid < 256 && i0 + tid < Cin) {
                float s = o[i0 + tid];
                for (int w = 0; w < nw; ++w) s += red[w * 256 + tid];
                o[i0 + tid] = s;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
int launch_colsel(hipStream_t st, const float* Y, const float* gamma, int B, int N, int C, float* ysel, int* arg) {
    hipLaunchKernelGGL(colsel_kernel, dim3((C + 63) / 64, B), dim3(256), 0, st, Y, gamma, N, C, ysel, arg);
    return mlsp_launch_status();
}
int launch_colsel_panels(hipStream_t st, const float* pv, const int* pr, const float* gamma, int B, int N, int C, int panel_rows,
                         float* ysel, int* arg, const float* bn, int act, float slope, float* out) {
    hipLaunchKernelGGL(colsel_panels_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, pv, pr, gamma, B, N, C, N / panel_rows, ysel,
                       arg, bn, act, slope, out);
    return mlsp_launch_status();
}
int launch_colsel_out(hipStream_t st, const float* ysel, const float* bn, int B, int C, int act, float slope, float* out) {
    hipLaunchKernelGGL(colsel_out_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, ysel, bn, B * C, C, act, slope, out);
    return mlsp_launch_status();
}
int launch_colmax_bwd_coef(hipStream_t st, const float* dOut, const float* out, const float* ysel, const float* bn, int B, int C,
                           double count, int act, float slope, int training, float* g, float* coef, float* dgamma, float* dbeta) {
    hipLaunchKernelGGL(colmax_bwd_coef_kernel, dim3(C), dim3(64), 0, st, dOut, out, ysel, bn, B, C, count, act, slope, training, g,
                       coef, dgamma, dbeta);
    return mlsp_launch_status();
}
int launch_scale_rows(hipStream_t st, const float* W, int ldw, const float* rowscale, int Cout, int Cin, float* Wb) {
    hipLaunchKernelGGL(scale_rows_kernel, dim3((Cout * Cin + 255) / 256), dim3(256), 0, st, W, ldw, rowscale, Cout, Cin, Wb);
    return mlsp_launch_status();
}
int launch_wt_vec_neg(hipStream_t st, const float* W, int ldw, const float* v, int Cout, int Cin, float* negr) {
    hipLaunchKernelGGL(wt_vec_neg_kernel, dim3((Cin + 63) / 64), dim3(1024), 0, st, W, ldw, v, Cout, Cin, negr, (const float*)nullptr, (float*)nullptr);
    return mlsp_launch_status();
}
// negr = -W^T v and Wb = diag(rowscale) W in one launch
int launch_wt_vec_neg_scale_rows(hipStream_t st, const float* W, int ldw, const float* v, const float* rowscale, int Cout, int Cin, float* negr,
                                 float* Wb) {
    hipLaunchKernelGGL(wt_vec_neg_kernel, dim3((Cin + 63) / 64 + (Cout * Cin + 1023) / 1024), dim3(1024), 0, st, W, ldw, v, Cout, Cin, negr,
                       rowscale, Wb);
    return mlsp_launch_status();
}
int launch_colmax_gather_rows(hipStream_t st, const float* g, const int* arg, const float* X, int ldx, int B, int N, int Cout,
                              int Cin, float* S) {
    hipLaunchKernelGGL(colmax_gather_rows_kernel, dim3(Cout, (Cin + 255) / 256), dim3(256), 0, st, g, arg, X, ldx, B, N, Cout, Cin, S);
    return mlsp_launch_status();
}
int launch_colmax_dw(hipStream_t st, const float* S, const float* WG, const float* sx, const float* coef, const float* bn, int Cout,
                     int Cin, float* dW) {
    hipLaunchKernelGGL(colmax_dw_kernel, dim3((Cout * Cin + 255) / 256), dim3(256), 0, st, S, WG, sx, coef, bn, Cout, Cin, dW);
    return mlsp_launch_status();
}
int launch_colmax_scatter_rows(hipStream_t st, const float* g, const int* arg, const float* W, int ldw, int B, int N, int Cout,
                               int Cin, float* dX, int lddx) {
    size_t lds = ((size_t)2 * N + 1 + 3 * Cout + 16 * 256) * sizeof(int);
    if (lds > 150 * 1024) return MLSP_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = mlsp_lds_limit((const void*)colmax_scatter_rows_kernel, lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(colmax_scatter_rows_kernel, dim3(CSR_SPLIT, B), dim3(1024), lds, st, g, arg, W, ldw, N, Cout, Cin, dX, lddx);
    return mlsp_launch_status();
}
