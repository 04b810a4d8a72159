// EdgeConv on the kNN graph, algebraically folded, plus the materialising graph-feature gather.
//
// Reference (PointDA/model_utils.py:18-42 + conv_2d :45-63 + `.max(dim=-1)` Models.py:115-129):
//     feat_e = [x_j - x_i ; x_i]            for every edge e = (i, j = idx[i][s])
//     y_e    = W feat_e                     1x1 conv, W = [Wa | Wb]   ([Cout, 2C])
//     out_i  = max_s LeakyReLU(BN(y_e))     BN statistics over all B*N*k edges
//
// Folding (hard parts 2 and 3):
//     y_e = Wa x_j + (Wb - Wa) x_i = u_j + v_i       with  [u|v] = x [Wa ; Wb-Wa]^T   (ONE per-point GEMM)
//     act(BN(.)) is monotone per channel, so  max_s act(BN(y_e)) = act(BN(v_i + max_s u_j))  when the BN scale
//     gamma*invstd >= 0, and  act(BN(v_i + min_s u_j))  otherwise;  sum_e y, sum_e y^2 follow from
//     s1_i = sum_s u_j and s2_i = sum_s u_j^2.  The [B,2C,N,k] and [B,Cout,N,k] tensors never exist; the
//     per-edge work is a row gather of u (L2-resident: one cloud's u is <= 1 MB), k x fewer MACs.
// Backward uses the same closed forms; the scatter-add onto neighbours becomes a gather over the
// reverse index built by knn.hip (deterministic, no float atomics).
#include "common.h"
#include <math.h>
#include <cstdlib>

#define EDGE_PTS_PER_WAVE 16

// Wd = [Wa ; Wb - Wa]  ([2*Cout, C]) from the reference-layout weight W [Cout, 2C].  amax (nullable, <= 256 workgroups): the 256 partial
// maxima of |Wd| for the f16x3 product that reads it (gemm.hip amax_reserve) -- one per workgroup, the rest zero.
__global__ __launch_bounds__(256) void build_wd_kernel(const float* __restrict__ W, int Cout, int C, float* __restrict__ Wd, float* __restrict__ amax) {
    __shared__ float sm[4];
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.f;
    if (t < Cout * C) {
        int o = t / C, c = t % C;
        float wa = W[(size_t)o * 2 * C + c], wb = W[(size_t)o * 2 * C + C + c];
        Wd[(size_t)o * C + c] = wa;
        Wd[(size_t)(Cout + o) * C + c] = wb - wa;
        m = fmaxf(fabsf(wa), fabsf(wb - wa));
    }
    if (!amax) return;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) amax[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    if (blockIdx.x == 0 && threadIdx.x >= gridDim.x) amax[threadIdx.x] = 0.f;
}
// the same + the eval-mode BatchNorm vectors of up to two BatchNorm stages of the calling entry point (scale | shift | mean | invstd from the
// running statistics: bn_eval_prepare_kernel's arithmetic) -- an eval-mode EdgeConv / T-Net forward folds its weight and prepares its
// BatchNorm in ONE launch (PointSegDA's layers have no BatchNorm: they run as identity-BN eval layers, three + two such launches a step)
struct EvalPrep { int C; const float* gamma; const float* beta; const float* rm; const float* rv; float eps; float* save; };
__device__ __forceinline__ void eval_prep_one(const EvalPrep& e, int c) {
    const float invstd = 1.0f / sqrtf(e.rv[c] + e.eps);
    const float sc = e.gamma[c] * invstd;
    e.save[c] = sc; e.save[e.C + c] = e.beta[c] - e.rm[c] * sc; e.save[2 * e.C + c] = e.rm[c]; e.save[3 * e.C + c] = invstd;
}
__global__ void build_wd_eval_kernel(const float* __restrict__ W, int Cout, int C, float* __restrict__ Wd, EvalPrep a, EvalPrep b) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < a.C) eval_prep_one(a, t);
    if (t < b.C) eval_prep_one(b, t);
    if (t >= Cout * C) return;
    int o = t / C, c = t % C;
    float wa = W[(size_t)o * 2 * C + c], wb = W[(size_t)o * 2 * C + C + c];
    Wd[(size_t)o * C + c] = wa;
    Wd[(size_t)(Cout + o) * C + c] = wb - wa;
}
// dW from dWd:  dWa = dWd_u - dWd_v,  dWb = dWd_v
__global__ void unbuild_wd_kernel(const float* __restrict__ dWd, int Cout, int C, float* __restrict__ dW) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Cout * C) return;
    int o = t / C, c = t % C;
    float du = dWd[(size_t)o * C + c], dv = dWd[(size_t)(Cout + o) * C + c];
    dW[(size_t)o * 2 * C + c] = du - dv;
    dW[(size_t)o * 2 * C + C + c] = dv;
}

// per point: selected extreme of u over the neighbours, its slot, s1; per channel: partial sums of y and y^2
__global__ __launch_bounds__(256) void edge_reduce_kernel(const float* __restrict__ uv, const int* __restrict__ idx,
                                                          const float* __restrict__ gamma, int P, int N, int Cout, int k,
                                                          float* __restrict__ msel, uint8_t* __restrict__ argsel,
                                                          float* __restrict__ s1out, double* __restrict__ part) {
    extern __shared__ double shd[];   // [2][4][Cout]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ld = 2 * Cout;
    const int p0 = (blockIdx.x * 4 + w) * EDGE_PTS_PER_WAVE;
    for (int c = lane; c < Cout; c += 64) {
        const bool use_max = gamma[c] >= 0.f;
        double ps = 0.0, pq = 0.0;
        for (int pi = 0; pi < EDGE_PTS_PER_WAVE; ++pi) {
            const int i = p0 + pi;               // wave-uniform
            if (i >= P) break;
            const int base = (i / N) * N;        // first point of this cloud
            const int* irow = idx + (size_t)i * k;
            float best = 0.f, s1 = 0.f, s2 = 0.f;
            int bs = 0;
            for (int s = 0; s < k; ++s) {
                const int j = base + irow[s];    // scalar load
                float u = uv[(size_t)j * ld + c];
                s1 += u; s2 = fmaf(u, u, s2);
                bool take = (s == 0) || (use_max ? (u > best) : (u < best));
                best = take ? u : best; bs = take ? s : bs;
            }
            float v = uv[(size_t)i * ld + Cout + c];
            msel[(size_t)i * Cout + c] = best;
            argsel[(size_t)i * Cout + c] = (uint8_t)bs;
            s1out[(size_t)i * Cout + c] = s1;
            ps += (double)s1 + (double)k * v;
            pq += (double)s2 + 2.0 * (double)v * s1 + (double)k * v * v;
        }
        shd[(0 * 4 + w) * Cout + c] = ps;
        shd[(1 * 4 + w) * Cout + c] = pq;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Cout; c += 256) {
        double a = 0.0, b = 0.0;
        for (int u = 0; u < 4; ++u) { a += shd[(0 * 4 + u) * Cout + c]; b += shd[(1 * 4 + u) * Cout + c]; }
        part[((size_t)blockIdx.x * 2 + 0) * Cout + c] = a;
        part[((size_t)blockIdx.x * 2 + 1) * Cout + c] = b;
    }
}


// Vectorised form for Cout in {64,128,256}: LR = Cout/4 lanes cover one neighbour row with 16-byte loads and the
// wave's 64/LR lane groups take different neighbours of the same point at once (4x fewer load instructions,
// 16 B per lane in flight); groups are combined with a (value, slot) butterfly that keeps the first-occurrence rule.
template <int LR>
__global__ __launch_bounds__(256) void edge_reduce_vec_kernel(const float* __restrict__ uv, const int* __restrict__ idx,
                                                              const float* __restrict__ gamma, int P, int N, int k,
                                                              float* __restrict__ msel, uint8_t* __restrict__ argsel,
                                                              float* __restrict__ s1out, double* __restrict__ part) {
    constexpr int Cout = LR * 4, NP = 64 / LR;
    __shared__ double shd[2][4][Cout];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane / LR, c = (lane % LR) * 4;
    const int ld = 2 * Cout;
    int p0 = (blockIdx.x * 4 + w) * EDGE_PTS_PER_WAVE;
    if (N % (4 * EDGE_PTS_PER_WAVE) == 0) {              // whole clouds per XCD (see xcd_cloud_map)
        int cloud, chunk;
        xcd_cloud_map(blockIdx.x, N / (4 * EDGE_PTS_PER_WAVE), P / N, cloud, chunk);
        p0 = cloud * N + (chunk * 4 + w) * EDGE_PTS_PER_WAVE;
    }
    const f32x4 g4 = *(const f32x4*)(gamma + c);
    bool use_max[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) use_max[e] = g4[e] >= 0.f;
    double ps[4] = {0, 0, 0, 0}, pq[4] = {0, 0, 0, 0};
    for (int pi = 0; pi < EDGE_PTS_PER_WAVE; ++pi) {
        const int i = p0 + pi;
        if (i >= P) break;
        const int base = (i / N) * N;
        const int* irow = idx + (size_t)i * k;
        float best[4], s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
        int bs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { best[e] = use_max[e] ? -INFINITY : INFINITY; bs[e] = 255; }
        // one coalesced load of the point's index row, then every neighbour row address comes from a lane shuffle:
        // the k row gathers are independent and issue back-to-back instead of chaining index -> row per neighbour
        const int jv = lane < k ? base + irow[lane] : base;
        for (int s0 = 0; s0 < k; s0 += NP) {            // wave-uniform trip count: every lane takes part in the shuffle
            const int s = s0 + sub;
            const bool ok = s < k;
            const int j = __shfl(jv, ok ? s : 0, 64);
            const f32x4 u = *(const f32x4*)(uv + (size_t)j * ld + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ue = ok ? u[e] : 0.f;
                s1[e] += ue; s2[e] = fmaf(ue, ue, s2[e]);
                bool take = ok && (use_max[e] ? (u[e] > best[e]) : (u[e] < best[e]));
                best[e] = take ? u[e] : best[e]; bs[e] = take ? s : bs[e];
            }
        }
#pragma unroll
        for (int o = LR; o < 64; o <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float ob = __shfl_xor(best[e], o, 64);
                int os = __shfl_xor(bs[e], o, 64);
                s1[e] += __shfl_xor(s1[e], o, 64);
                s2[e] += __shfl_xor(s2[e], o, 64);
                bool better = use_max[e] ? (ob > best[e]) : (ob < best[e]);
                bool take = better || (ob == best[e] && os < bs[e]);
                best[e] = take ? ob : best[e]; bs[e] = take ? os : bs[e];
            }
        }
        if (sub == 0) {
            const f32x4 v = *(const f32x4*)(uv + (size_t)i * ld + Cout + c);
            f32x4 b4, s4;
            uint32_t a4 = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                b4[e] = best[e]; s4[e] = s1[e]; a4 |= (uint32_t)(bs[e] & 255) << (8 * e);
                ps[e] += (double)s1[e] + (double)k * v[e];
                pq[e] += (double)s2[e] + 2.0 * (double)v[e] * s1[e] + (double)k * v[e] * v[e];
            }
            *(f32x4*)(msel + (size_t)i * Cout + c) = b4;
            *(f32x4*)(s1out + (size_t)i * Cout + c) = s4;
            *(uint32_t*)(argsel + (size_t)i * Cout + c) = a4;
        }
    }
    if (sub == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { shd[0][w][c + e] = ps[e]; shd[1][w][c + e] = pq[e]; }
    }
    __syncthreads();
    for (int cc = threadIdx.x; cc < Cout; cc += 256) {
        part[((size_t)blockIdx.x * 2 + 0) * Cout + cc] = shd[0][0][cc] + shd[0][1][cc] + shd[0][2][cc] + shd[0][3][cc];
        part[((size_t)blockIdx.x * 2 + 1) * Cout + cc] = shd[1][0][cc] + shd[1][1][cc] + shd[1][2][cc] + shd[1][3][cc];
    }
}

// LDS-resident form (whole clouds, N <= 4096, k <= 32, Cout % 16 == 0): a workgroup owns one (cloud, 16-channel slice,
// point chunk).  The slice of u for the WHOLE cloud (N x 16 floats, 64 KB at N = 1024) and the chunk's neighbour lists
// (u16) are staged in LDS once; the k row gathers of every point then run at LDS rate instead of one L2 round trip per
// neighbour row.  A thread owns (point, channel quad) and walks its k neighbours in slot order, so the first-occurrence
// rule needs no cross-lane merge.  Partial BN statistics: one fp64 row per (cloud, chunk), each slice writes its columns.
#ifndef ELDS_CS
#define ELDS_CS 8           // channels of a slice: 32 KB of LDS at N = 1024 (four workgroups per CU; 16 measured 2 % slower)
#endif
template <int KMAX, bool EXACT>     // EXACT: k == KMAX, the neighbour loop is straight-line (all index / row reads of a point in flight)
__global__ __launch_bounds__(256) void edge_reduce_lds_kernel(const float* __restrict__ uv, const int* __restrict__ idx,
                                                              const float* __restrict__ gamma, int B, int N, int k, int Cout, int psplit,
                                                              float* __restrict__ msel, uint8_t* __restrict__ argsel,
                                                              float* __restrict__ s1out, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float esm[];
    constexpr int CS = ELDS_CS, QPP = CS / 4, PL = 256 / QPP;
    float* Us = esm;                                           // [N][CS]
    unsigned short* Is = (unsigned short*)(esm + (size_t)N * CS);   // [chunk points][k]
    const int tid = threadIdx.x;
    const int nsl = Cout / CS, bpc = nsl * psplit;
    int b, r;
    xcd_cloud_map(blockIdx.x, bpc, B, b, r);                   // all workgroups of a cloud on one XCD: uv rows come from its L2
    const int sl = r % nsl, ch = r / nsl;
    const int c0 = sl * CS, ld = 2 * Cout;
    const int pper = (N + psplit - 1) / psplit, pbeg = ch * pper, pend = min(N, pbeg + pper);
    const int q = tid % QPP, pl = tid / QPP;
    const float* ub = uv + (size_t)b * N * ld;
    for (int row = pl; row < N; row += PL)
        *(f32x4*)(Us + row * CS + 4 * q) = *(const f32x4*)(ub + (size_t)row * ld + c0 + 4 * q);
    const int* ib = idx + ((size_t)b * N + pbeg) * k;
    for (int t = tid; t < (pend - pbeg) * k; t += 256) Is[t] = (unsigned short)ib[t];
    __syncthreads();
    const f32x4 g4 = *(const f32x4*)(gamma + c0 + 4 * q);
    bool use_max[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) use_max[e] = g4[e] >= 0.f;
    double ps[4] = {0, 0, 0, 0}, pq[4] = {0, 0, 0, 0};
    for (int il = pl; il < pend - pbeg; il += PL) {
        const unsigned short* irow = Is + il * k;
        // min-channels (negative BN scale) are searched as the max of -u: one compare per value, strict > keeps the first slot
        // sums and sign flip as PACKED fp32 operations (two channels per instruction: this loop is VALU-bound, 24 -> 18 instructions
        // per neighbour); t = u * (+-1) is exact, the sums are the same IEEE operations in the same order as the scalar form
        float best[4];
        f32x2 s1a = {0.f, 0.f}, s1b = {0.f, 0.f}, s2a = {0.f, 0.f}, s2b = {0.f, 0.f};
        const f32x2 sga = {use_max[0] ? 1.f : -1.f, use_max[1] ? 1.f : -1.f}, sgb = {use_max[2] ? 1.f : -1.f, use_max[3] ? 1.f : -1.f};
        int bs[4] = {0, 0, 0, 0};
        int jr[KMAX];
#pragma unroll
        for (int s = 0; s < KMAX; ++s) jr[s] = (EXACT || s < k) ? (int)irow[s] : 0;
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            if (EXACT || s < k) {
                const f32x4 u = *(const f32x4*)(Us + jr[s] * CS + 4 * q);
                const f32x2 ua = {u[0], u[1]}, ub = {u[2], u[3]};
                s1a += ua; s1b += ub;
                s2a = __builtin_elementwise_fma(ua, ua, s2a); s2b = __builtin_elementwise_fma(ub, ub, s2b);
                const f32x2 ta = ua * sga, tb = ub * sgb;
                const float t[4] = {ta[0], ta[1], tb[0], tb[1]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool take = (s == 0) || (t[e] > best[e]);
                    best[e] = take ? t[e] : best[e]; bs[e] = take ? s : bs[e];
                }
            }
        }
        const float s1[4] = {s1a[0], s1a[1], s1b[0], s1b[1]}, s2[4] = {s2a[0], s2a[1], s2b[0], s2b[1]};
#pragma unroll
        for (int e = 0; e < 4; ++e) best[e] = use_max[e] ? best[e] : -best[e];
        const size_t i = (size_t)b * N + pbeg + il;
        const f32x4 v = *(const f32x4*)(uv + i * ld + Cout + c0 + 4 * q);
        f32x4 b4, s4;
        uint32_t a4 = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            b4[e] = best[e]; s4[e] = s1[e]; a4 |= (uint32_t)(bs[e] & 255) << (8 * e);
            ps[e] += (double)s1[e] + (double)k * v[e];
            pq[e] += (double)s2[e] + 2.0 * (double)v[e] * s1[e] + (double)k * v[e] * v[e];
        }
        *(f32x4*)(msel + i * Cout + c0 + 4 * q) = b4;
        *(f32x4*)(s1out + i * Cout + c0 + 4 * q) = s4;
        *(uint32_t*)(argsel + i * Cout + c0 + 4 * q) = a4;
    }
    __syncthreads();                                           // Us is dead: reuse it for the fp64 column reduction
    double* red = (double*)esm;                                // [2][PL][CS]
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[(0 * PL + pl) * CS + 4 * q + e] = ps[e]; red[(1 * PL + pl) * CS + 4 * q + e] = pq[e]; }
    __syncthreads();
    if (tid < 2 * CS) {
        const int which = tid / CS, c = tid % CS;
        double a = 0.0;
        for (int u = 0; u < PL; ++u) a += red[(which * PL + u) * CS + c];
        part[((size_t)(b * psplit + ch) * 2 + which) * Cout + c0 + c] = a;
    }
}

// ---- round 5: the same reduction with WIDE slices, one workgroup of 1024 threads per CU -----------------------------------------------
// Where edge_reduce_lds_kernel's time went (rocprofv3, B = 32, N = 1024, k = 20: 23 / 36 / 70 us at Cout = 64 / 128 / 256 for ~4 / 8 / 16 us
// of vector work): 8-channel slices are 32-byte pieces of 512 .. 2048-byte rows -- a quarter of every cache line fetched is used -- and
// with the point range cut in four (to keep the u16 neighbour lists small) every workgroup stages its slice of the WHOLE cloud for a
// quarter of the points: 134 MB of staging per launch at Cout = 256, line traffic 4x that.  Here a slice is CS = 16 channels (64-byte
// half lines), staged ONCE per (cloud, slice) when the grid allows, 64 KB per workgroup of 512 threads, two workgroups per CU (N <= 1024;
// one of 1024 threads for N <= 2048); a thread = (point, channel quad) walks the k neighbours of its point in slot order.  The neighbour
// indices are not staged: the lanes of a point read the same 80 bytes from global (one broadcast line).  The selection runs on t = +-u (sign
// of the BatchNorm scale folded in at staging: the max search is one compare, the sums are the same IEEE operations on negated values).
template <int KMAX, bool EXACT, int CS, int NT>
__global__ __launch_bounds__(NT, 4) void edge_reduce_wide_kernel(const float* __restrict__ uv, const int* __restrict__ idx,
                                                                const float* __restrict__ gamma, int B, int N, int k, int Cout, int psplit,
                                                                float* __restrict__ msel, uint8_t* __restrict__ argsel,
                                                                float* __restrict__ s1out, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float esm[];
    constexpr int QPP = CS / 4, PL = NT / QPP;                 // quads per point, points per pass
    float* Us = esm;                                           // [N][CS]  (+-u)
    const int tid = threadIdx.x;
    const int nsl = Cout / CS, bpc = nsl * psplit;
    int b, r;
    xcd_cloud_map(blockIdx.x, bpc, B, b, r);                   // all workgroups of a cloud on one XCD: uv rows come from its L2
    const int sl = r % nsl, ch = r / nsl;
    const int c0 = sl * CS, ld = 2 * Cout;
    const int pper = (N + psplit - 1) / psplit, pbeg = ch * pper, pend = min(N, pbeg + pper);
    const int q = tid % QPP, pl = tid / QPP;
    const float* ub = uv + (size_t)b * N * ld;
    const f32x4 g4 = *(const f32x4*)(gamma + c0 + 4 * q);
    f32x4 sg;
#pragma unroll
    for (int e = 0; e < 4; ++e) sg[e] = g4[e] >= 0.f ? 1.f : -1.f;
    for (int row = pl; row < N; row += 4 * PL) {               // four rows in flight per thread
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (row + u * PL < N) v[u] = *(const f32x4*)(ub + (size_t)(row + u * PL) * ld + c0 + 4 * q);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (row + u * PL < N) *(f32x4*)(Us + (row + u * PL) * CS + 4 * q) = v[u] * sg;
    }
    __syncthreads();
    double ps[4] = {0, 0, 0, 0}, pq[4] = {0, 0, 0, 0};
    constexpr int KC = (KMAX % 20 == 0) ? 20 : KMAX;           // neighbours per batch of index loads + row gathers (k = 40: two batches, 128 registers hold)
    for (int il = pbeg + pl; il < pend; il += PL) {
        const size_t i = (size_t)b * N + il;
        const int* irow = idx + i * k;
        const f32x4 v = *(const f32x4*)(uv + i * ld + Cout + c0 + 4 * q);
        float best[4];
        f32x2 s1a = {0.f, 0.f}, s1b = {0.f, 0.f}, s2a = {0.f, 0.f}, s2b = {0.f, 0.f};
        int bs[4] = {0, 0, 0, 0};
#pragma unroll 1
        for (int s0 = 0; s0 < KMAX; s0 += KC) {                // (not unrolled: the second batch's index loads must not be hoisted over the first)
            int jr[KC];
            if (EXACT && (KC % 4) == 0) {                      // k * 4 bytes per row, rows 16-byte aligned when k % 4 == 0
#pragma unroll
                for (int s4 = 0; s4 < KC / 4; ++s4) {
                    const int4 t = *(const int4*)(irow + s0 + 4 * s4);
                    jr[4 * s4] = t.x; jr[4 * s4 + 1] = t.y; jr[4 * s4 + 2] = t.z; jr[4 * s4 + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int s = 0; s < KC; ++s) jr[s] = (EXACT || s0 + s < k) ? irow[s0 + s] : 0;
            }
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                if (EXACT || s0 + s < k) {
                    const f32x4 t = *(const f32x4*)(Us + jr[s] * CS + 4 * q);
                    const f32x2 ta = {t[0], t[1]}, tb = {t[2], t[3]};
                    s1a += ta; s1b += tb;
                    s2a = __builtin_elementwise_fma(ta, ta, s2a); s2b = __builtin_elementwise_fma(tb, tb, s2b);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool take = (s0 + s == 0) || (t[e] > best[e]);
                        best[e] = take ? t[e] : best[e]; bs[e] = take ? s0 + s : bs[e];
                    }
                }
            }
        }
        const float s1[4] = {s1a[0] * sg[0], s1a[1] * sg[1], s1b[0] * sg[2], s1b[1] * sg[3]}, s2[4] = {s2a[0], s2a[1], s2b[0], s2b[1]};
        f32x4 b4, s4;
        uint32_t a4 = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            b4[e] = best[e] * sg[e]; s4[e] = s1[e]; a4 |= (uint32_t)(bs[e] & 255) << (8 * e);
            ps[e] += (double)s1[e] + (double)k * v[e];
            pq[e] += (double)s2[e] + 2.0 * (double)v[e] * s1[e] + (double)k * v[e] * v[e];
        }
        *(f32x4*)(msel + i * Cout + c0 + 4 * q) = b4;
        *(f32x4*)(s1out + i * Cout + c0 + 4 * q) = s4;
        *(uint32_t*)(argsel + i * Cout + c0 + 4 * q) = a4;
    }
    __syncthreads();                                           // Us is dead: reuse it for the fp64 column reduction
    double* red = (double*)esm;                                // [2][PL][CS]
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[(0 * PL + pl) * CS + 4 * q + e] = ps[e]; red[(1 * PL + pl) * CS + 4 * q + e] = pq[e]; }
    __syncthreads();
    if (tid < 2 * CS) {
        const int which = tid / CS, c = tid % CS;
        double a = 0.0;
        for (int u = 0; u < PL; ++u) a += red[(which * PL + u) * CS + c];
        part[((size_t)(b * psplit + ch) * 2 + which) * Cout + c0 + c] = a;
    }
}

// By-product bound of duv for the f16x3 products that read it (gemm.hip amax_reserve): every WORKGROUP raises one of 256 partial maxima
// (non-negative floats order like their bit patterns; the slot was zeroed by edge_bwd_reduce_vec_kernel, earlier on the stream).  One
// atomic per workgroup: one per wave -- 32768 of them on 256 addresses -- cost the gather 15 us per launch.  Every thread of the
// workgroup must call it (barrier inside).
__device__ __forceinline__ void edge_amax_raise(float* __restrict__ amax, float m) {
    __shared__ float wm[4];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax((unsigned int*)amax + (blockIdx.x & 255), __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

// vectorised reverse gather (same lane layout): du_j over rev(j)
template <int LR>
__global__ __launch_bounds__(256) void edge_bwd_gather_vec_kernel(const float* __restrict__ gz, const uint8_t* __restrict__ argsel,
                                                                  const float* __restrict__ uv, const int* __restrict__ rev_off,
                                                                  const int* __restrict__ rev_ent, int P, int N,
                                                                  const float* __restrict__ scale, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ mean_dz,
                                                                  const float* __restrict__ mean_dzy, float* __restrict__ duv,
                                                                  float* __restrict__ duv_amax) {
    constexpr int Cout = LR * 4, NP = 64 / LR;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int j = blockIdx.x * 4 + w;
    if (N % 4 == 0) {
        int cloud, chunk;
        xcd_cloud_map(blockIdx.x, N / 4, P / N, cloud, chunk);
        j = cloud * N + chunk * 4 + w;
    }
    float pm = 0.f;
    if (j < P) {                                         // (wave-uniform; no early return: the workgroup meets in edge_amax_raise)
        const int sub = lane / LR, c = (lane % LR) * 4;
        const int base = (j / N) * N;
        const int e0 = rev_off[j], e1 = rev_off[j + 1];
        const int ld = 2 * Cout;
        float A[4] = {0, 0, 0, 0}, Bc[4] = {0, 0, 0, 0};
        if (mean_dz) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float sc = scale[c + e]; A[e] = sc * mean_dz[c + e]; Bc[e] = sc * invstd[c + e] * mean_dzy[c + e]; }
        }
        float acc[4] = {0, 0, 0, 0};
        for (int ec = e0; ec < e1; ec += 64) {              // chunks of 64 reverse entries: one coalesced load, then shuffles
            const int nent = min(64, e1 - ec);
            const int entv = lane < nent ? rev_ent[ec + lane] : 0;
            for (int t0 = 0; t0 < nent; t0 += NP) {          // wave-uniform trip count (see edge_reduce_vec_kernel)
                const int t = t0 + sub;
                const bool ok = t < nent;
                const int ent = __shfl(entv, ok ? t : 0, 64);
                const int i = base + (ent >> 8), slot = ent & 255;
                const f32x4 g = *(const f32x4*)(gz + (size_t)i * Cout + c);
                const uint32_t a4 = *(const uint32_t*)(argsel + (size_t)i * Cout + c);
                const f32x4 v = *(const f32x4*)(uv + (size_t)i * ld + Cout + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float ge = (((a4 >> (8 * e)) & 255) == (uint32_t)slot) ? g[e] : 0.f;
                    acc[e] += ok ? (ge - Bc[e] * v[e]) : 0.f;
                }
            }
        }
#pragma unroll
        for (int o = LR; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
        if (sub == 0) {
            const f32x4 u = *(const f32x4*)(uv + (size_t)j * ld + c);
            f32x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o4[e] = acc[e] - (float)(e1 - e0) * (A[e] + Bc[e] * (u[e] - mean[c + e])); pm = fmaxf(pm, fabsf(o4[e])); }
            *(f32x4*)(duv + (size_t)j * ld + c) = o4;
        }
    }
    if (duv_amax) edge_amax_raise(duv_amax, pm);
}

// out = act(scale*(msel + v) + shift)
__global__ __launch_bounds__(256) void edge_select_act_kernel(const float* __restrict__ msel, const float* __restrict__ uv,
                                                              int P, int Cout, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int act, float slope,
                                                              float* __restrict__ out, int ldo) {
    size_t total = (size_t)P * Cout;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        size_t i = t / Cout; int c = (int)(t % Cout);
        float y = msel[t] + uv[i * 2 * Cout + Cout + c];
        out[i * ldo + c] = lrelu_or_relu(fmaf(y, scale[c], shift[c]), act, slope);
    }
}

// backward pass 1: column partial sums of dz and dz*yhat_sel (only the selected edge of each
// (point, channel) carries an incoming gradient)
__global__ __launch_bounds__(256) void edge_bwd_reduce_kernel(const float* __restrict__ dOut, const float* __restrict__ out,
                                                              const float* __restrict__ msel, const float* __restrict__ uv,
                                                              int P, int Cout, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, int act, float slope,
                                                              double* __restrict__ part, int lddo, int ldo) {
    __shared__ double sh[2][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * 512, r1 = min(P, r0 + 512);
    double s = 0.0, q = 0.0;
    if (c < Cout) {
        float mu = mean[c], is = invstd[c];
        for (int r = r0 + w; r < r1; r += 4) {
            size_t t = (size_t)r * Cout + c;
            float d = dOut[(size_t)r * lddo + c];
            if (act && !(out[(size_t)r * ldo + c] > 0.f)) d *= (act == 1 ? 0.f : slope);
            float yh = (msel[t] + uv[(size_t)r * 2 * Cout + Cout + c] - mu) * is;
            s += d; q += (double)d * yh;
        }
    }
    sh[0][w][lane] = s; sh[1][w][lane] = q;
    __syncthreads();
    if (w == 0 && c < Cout) {
        part[((size_t)blockIdx.y * 2 + 0) * Cout + c] = sh[0][0][lane] + sh[0][1][lane] + sh[0][2][lane] + sh[0][3][lane];
        part[((size_t)blockIdx.y * 2 + 1) * Cout + c] = sh[1][0][lane] + sh[1][1][lane] + sh[1][2][lane] + sh[1][3][lane];
    }
}

// backward pass 2 (per point):  gz = scale*dz ;  dv_i = gz - k*A - Bc*(s1 + k*v - k*mean)
//   A = scale*mean_dz, Bc = scale*invstd*mean_dzy  (both 0 in eval mode: pass mean_dz = null)
__global__ __launch_bounds__(256) void edge_bwd_point_kernel(const float* __restrict__ dOut, const float* __restrict__ out,
                                                             const float* __restrict__ uv, const float* __restrict__ s1,
                                                             int P, int Cout, int k, const float* __restrict__ scale,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ mean_dz,
                                                             const float* __restrict__ mean_dzy, int act, float slope,
                                                             float* __restrict__ gz, float* __restrict__ duv, int lddo, int ldo) {
    size_t total = (size_t)P * Cout;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        size_t i = t / Cout; int c = (int)(t % Cout);
        float d = dOut[i * lddo + c];
        if (act && !(out[i * ldo + c] > 0.f)) d *= (act == 1 ? 0.f : slope);
        float sc = scale[c];
        float g = sc * d;
        gz[t] = g;
        float dv = g;
        if (mean_dz) {
            float A = sc * mean_dz[c], Bc = sc * invstd[c] * mean_dzy[c];
            float v = uv[i * 2 * Cout + Cout + c];
            dv = g - (float)k * A - Bc * (s1[t] + (float)k * (v - mean[c]));
        }
        duv[i * 2 * Cout + Cout + c] = dv;
    }
}


// ---- vectorised (16 B per lane) forms of the per-point passes; Cout % 4 == 0, Cout <= 1024, 256 % (Cout/4) == 0 ----
#define EVROWS 64
__global__ __launch_bounds__(256) void edge_bwd_reduce_vec_kernel(const float* __restrict__ dOut, const float* __restrict__ out,
                                                                  const float* __restrict__ msel, const float* __restrict__ uv,
                                                                  int P, int Cout, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, int act, float slope,
                                                                  double* __restrict__ part, int lddo, int ldo,
                                                                  float* __restrict__ duv_amax) {
    __shared__ double shd[256 * 8];
    const int tid = threadIdx.x, tpr = Cout >> 2, nrg = 256 / tpr;
    if (duv_amax && blockIdx.x == 0) duv_amax[tid] = 0.f;       // the 256 partial maxima the two passes below raise (edge_amax_raise)
    const int cg = tid % tpr, rg = tid / tpr, c = cg * 4;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const f32x4 mu = *(const f32x4*)(mean + c), is = *(const f32x4*)(invstd + c);
    const int r0 = blockIdx.x * EVROWS, r1 = min(P, r0 + EVROWS);
    for (int r = r0 + rg; r < r1; r += nrg) {
        const size_t t = (size_t)r * Cout + c;
        const f32x4 d4 = *(const f32x4*)(dOut + (size_t)r * lddo + c), o4 = *(const f32x4*)(out + (size_t)r * ldo + c);
        const f32x4 m4 = *(const f32x4*)(msel + t);
        const f32x4 v4 = *(const f32x4*)(uv + (size_t)r * 2 * Cout + Cout + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float d = d4[e];
            if (act && !(o4[e] > 0.f)) d *= (act == 1 ? 0.f : slope);
            s[e] += d; q[e] += (double)d * ((m4[e] + v4[e] - mu[e]) * is[e]);
        }
    }
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
            part[((size_t)blockIdx.x * 2 + 0) * Cout + tid * 4 + e] = a[e];
            part[((size_t)blockIdx.x * 2 + 1) * Cout + tid * 4 + e] = a[4 + e];
        }
    }
}

__global__ __launch_bounds__(256) void edge_select_act_vec_kernel(const float* __restrict__ msel, const float* __restrict__ uv,
                                                                  size_t total4, int C4, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, int act, float slope,
                                                                  float* __restrict__ out, int ldo) {
    for (size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x; v < total4; v += (size_t)gridDim.x * blockDim.x) {
        const size_t i = v / C4; const int c = (int)(v % C4) * 4;
        const f32x4 m = *(const f32x4*)(msel + v * 4), vv = *(const f32x4*)(uv + i * 8 * C4 + 4 * C4 + c);
        const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = lrelu_or_relu(fmaf(m[e] + vv[e], sc[e], sh[e]), act, slope);
        *(f32x4*)(out + i * ldo + c) = o;
    }
}

__global__ __launch_bounds__(256) void edge_bwd_point_vec_kernel(const float* __restrict__ dOut, const float* __restrict__ out,
                                                                 const float* __restrict__ uv, const float* __restrict__ s1,
                                                                 size_t total4, int C4, int k, const float* __restrict__ scale,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ mean_dz, const float* __restrict__ mean_dzy,
                                                                 int act, float slope, float* __restrict__ gz,
                                                                 float* __restrict__ duv, int lddo, int ldo,
                                                                 float* __restrict__ duv_amax) {
    const float fk = (float)k;
    float pm = 0.f;
    for (size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x; v < total4; v += (size_t)gridDim.x * blockDim.x) {
        const size_t i = v / C4; const int c = (int)(v % C4) * 4;
        const f32x4 d4 = *(const f32x4*)(dOut + i * lddo + c), o4 = *(const f32x4*)(out + i * ldo + c);
        const f32x4 sc = *(const f32x4*)(scale + c);
        f32x4 g, dv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float d = d4[e];
            if (act && !(o4[e] > 0.f)) d *= (act == 1 ? 0.f : slope);
            g[e] = sc[e] * d; dv[e] = g[e];
        }
        if (mean_dz) {
            const f32x4 vv = *(const f32x4*)(uv + i * 8 * C4 + 4 * C4 + c), s4 = *(const f32x4*)(s1 + v * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float A = sc[e] * mean_dz[c + e], Bc = sc[e] * invstd[c + e] * mean_dzy[c + e];
                dv[e] = g[e] - fk * A - Bc * (s4[e] + fk * (vv[e] - mean[c + e]));
            }
        }
        *(f32x4*)(gz + v * 4) = g;
        *(f32x4*)(duv + i * 8 * C4 + 4 * C4 + c) = dv;
#pragma unroll
        for (int e = 0; e < 4; ++e) pm = fmaxf(pm, fabsf(dv[e]));
    }
    if (duv_amax) edge_amax_raise(duv_amax, pm);
}

// backward pass 3 (reverse gather, wave per destination j):
//   du_j = sum_{(i,slot) in rev(j)} ( [argsel[i]==slot]*gz_i - Bc*v_i ) - deg_j*(A + Bc*(u_j - mean))
__global__ __launch_bounds__(256) void edge_bwd_gather_kernel(const float* __restrict__ gz, const uint8_t* __restrict__ argsel,
                                                              const float* __restrict__ uv, const int* __restrict__ rev_off,
                                                              const int* __restrict__ rev_ent, int P, int N, int Cout,
                                                              const float* __restrict__ scale, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ mean_dz,
                                                              const float* __restrict__ mean_dzy, float* __restrict__ duv) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = blockIdx.x * 4 + w;     // wave-uniform
    if (j >= P) return;
    const int base = (j / N) * N;
    const int e0 = rev_off[j], e1 = rev_off[j + 1];
    const int ld = 2 * Cout;
    for (int c = lane; c < Cout; c += 64) {
        float A = 0.f, Bc = 0.f;
        if (mean_dz) { float sc = scale[c]; A = sc * mean_dz[c]; Bc = sc * invstd[c] * mean_dzy[c]; }
        float acc = 0.f;
        for (int e = e0; e < e1; ++e) {
            int ent = rev_ent[e];                        // scalar load
            int i = base + (ent >> 8), slot = ent & 255;
            size_t t = (size_t)i * Cout + c;
            float g = (argsel[t] == slot) ? gz[t] : 0.f;
            acc += g - Bc * uv[(size_t)i * ld + Cout + c];
        }
        float u = uv[(size_t)j * ld + c];
        duv[(size_t)j * ld + c] = acc - (float)(e1 - e0) * (A + Bc * (u - mean[c]));
    }
}

// ---------------------------------------------------------------------------------------------
// Materialising graph feature (public get_graph_feature API and the T-Net per-edge stage):
//   F[(i,s)][0:C] = x_j - x_i ;  F[(i,s)][C:2C] = x_i          F is [P*k][2C] (edge-major)
__global__ __launch_bounds__(256) void graph_feature_fwd_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                                int P, int N, int C, int k, float* __restrict__ F) {
    size_t total = (size_t)P * k * C;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(t % C);
        size_t e = t / C;
        size_t i = e / k;
        int j = (int)(i / N) * N + idx[e];
        float xi = x[i * C + c];
        F[e * 2 * C + c] = x[(size_t)j * C + c] - xi;
        F[e * 2 * C + C + c] = xi;
    }
}

// dx_i = sum_s (dF_ctr[(i,s)] - dF_nbr[(i,s)]) + sum_{(i',slot) in rev(i)} dF_nbr[(i',slot)]
__global__ __launch_bounds__(256) void graph_feature_bwd_kernel(const float* __restrict__ dF, const int* __restrict__ rev_off,
                                                                const int* __restrict__ rev_ent, int P, int N, int C, int k,
                                                                float* __restrict__ dx) {
    size_t total = (size_t)P * C;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(t % C);
        size_t i = t / C;
        int base = (int)(i / N) * N;
        float acc = 0.f;
        for (int s = 0; s < k; ++s) {
            const float* r = dF + (i * k + s) * 2 * C;
            acc += r[C + c] - r[c];
        }
        for (int e = rev_off[i]; e < rev_off[i + 1]; ++e) {
            int ent = rev_ent[e];
            size_t src = (size_t)(base + (ent >> 8)) * k + (ent & 255);
            acc += dF[src * 2 * C + c];
        }
        dx[t] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
static inline int ew_blocks2(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b < 4096 ? (b ? b : 1) : 4096);
}

// number of partial rows launch_edge_bwd_reduce writes for this shape
int edge_bwd_reduce_parts(int P, int Cout, const void* a, const void* b, const void* c, const void* d, const void* e, const void* f,
                          int lddo, int ldo) {
    bool vec = Cout % 4 == 0 && Cout <= 1024 && 256 % (Cout / 4) == 0 && lddo % 4 == 0 && ldo % 4 == 0 &&
               ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e | (uintptr_t)f) & 15) == 0);
    return vec ? (P + EVROWS - 1) / EVROWS : (P + 511) / 512;
}
int edge_reduce_parts(int P) { return (P + 4 * EDGE_PTS_PER_WAVE - 1) / (4 * EDGE_PTS_PER_WAVE); }

bool build_wd_leaves_bound(int Cout, int C) { return (Cout * C + 255) / 256 <= 256; }
int launch_build_wd(hipStream_t st, const float* W, int Cout, int C, float* Wd, float* amax) {
    if (amax && (Cout * C + 255) / 256 > 256) return MLSP_ERR_ARG;          // (callers ask build_wd_leaves_bound first)
    hipLaunchKernelGGL(build_wd_kernel, dim3((Cout * C + 255) / 256), dim3(256), 0, st, W, Cout, C, Wd, amax);
    return mlsp_launch_status();
}
// Ca / Cb = 0: that stage is absent
int launch_build_wd_eval(hipStream_t st, const float* W, int Cout, int C, float* Wd, int Ca, const float* ga, const float* ba, const float* rma,
                         const float* rva, float* sva, int Cb, const float* gb, const float* bb, const float* rmb, const float* rvb, float* svb,
                         float eps) {
    const EvalPrep a = {Ca, ga, ba, rma, rva, eps, sva}, b = {Cb, gb, bb, rmb, rvb, eps, svb};
    int n = Cout * C; if (Ca > n) n = Ca; if (Cb > n) n = Cb;
    hipLaunchKernelGGL(build_wd_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, st, W, Cout, C, Wd, a, b);
    return mlsp_launch_status();
}
int launch_unbuild_wd(hipStream_t st, const float* dWd, int Cout, int C, float* dW) {
    hipLaunchKernelGGL(unbuild_wd_kernel, dim3((Cout * C + 255) / 256), dim3(256), 0, st, dWd, Cout, C, dW);
    return mlsp_launch_status();
}
// *nparts_used (optional) receives the number of partial-statistics rows written (<= edge_reduce_parts(P))
int launch_edge_reduce(hipStream_t st, const float* uv, const int* idx, const float* gamma, int P, int N, int Cout, int k,
                       float* msel, uint8_t* argsel, float* s1, double* part, int* nparts_used) {
    const bool al = (((uintptr_t)uv | (uintptr_t)msel | (uintptr_t)s1 | (uintptr_t)gamma) & 15) == 0 && (((uintptr_t)argsel) & 3) == 0;
    if (nparts_used) *nparts_used = edge_reduce_parts(P);
#ifndef EDGE_NO_LDS
    static const bool no_wide = getenv("MLSP_EDGE_NO_WIDE") != nullptr;       // read-once A/B switch (tools/ab)
    if (!no_wide && al && P % N == 0 && N % 4 == 0 && (k == 20 || k == 40 || k <= 32) && Cout % 32 == 0 && N <= 2048) {
        // wide slices: CS = 32 channels, one 1024-thread workgroup per CU while the cloud's slice fits (N <= 1024: 128 KB), else 16
        static const int force_cs = getenv("MLSP_EDGE_WIDE_CS") ? atoi(getenv("MLSP_EDGE_WIDE_CS")) : 0;     // read-once A/B switch: 16 -> 16-channel slices, 512 threads, two workgroups per CU
        // measured (B = 32, N = 1024, k = 20, five launches of a step): 32 channels x 1024 threads 143 us, 16 x 512 (two workgroups per CU: one
        // stages while the other gathers) 129 us, 8 x 256 141 us; the round-4 kernel 182 us
        const int B = P / N, CS = force_cs == 8 ? 8 : (force_cs == 32 && N <= 1024) ? 32 : 16, nsl = Cout / CS;
        const int NT = CS == 8 ? 256 : (CS == 16 && N <= 1024) ? 512 : 1024;
        const int want = NT == 256 ? 1024 : NT == 512 ? 512 : 256;
        int psplit = 1;                                        // cut the point range only to give every CU its workgroup(s)
        while (psplit < 8 && B * nsl * psplit < want && N % (psplit * 2) == 0) psplit *= 2;
        const size_t lds = (size_t)N * CS * sizeof(float);
        const size_t red = (size_t)2 * (NT / (CS / 4)) * CS * sizeof(double);
        if ((lds > red ? lds : red) <= 150 * 1024 && B * psplit <= edge_reduce_parts(P)) {
            const size_t ldsz = lds > red ? lds : red;
#define EW_GO(KM, EX) do { auto kern = CS == 32 ? edge_reduce_wide_kernel<KM, EX, 32, 1024> : CS == 8 ? edge_reduce_wide_kernel<KM, EX, 8, 256> : NT == 512 ? edge_reduce_wide_kernel<KM, EX, 16, 512> : edge_reduce_wide_kernel<KM, EX, 16, 1024>; \
                if (ldsz > 64 * 1024) { hipError_t e_ = mlsp_lds_limit((const void*)kern, ldsz); if (e_ != hipSuccess) return (int)e_; } \
                hipLaunchKernelGGL(kern, dim3(B * nsl * psplit), dim3(NT), ldsz, st, uv, idx, gamma, B, N, k, Cout, psplit, msel, argsel, s1, part); } while (0)
            if (k == 20) EW_GO(20, true); else if (k <= 20) EW_GO(20, false); else if (k <= 32) EW_GO(32, false); else EW_GO(40, true);
#undef EW_GO
            if (nparts_used) *nparts_used = B * psplit;
            return mlsp_launch_status();
        }
    }
    if (al && Cout % ELDS_CS == 0 && P % N == 0 && N <= 4096 && k <= 40 && N % 4 == 0) {
        const int B = P / N, nsl = Cout / ELDS_CS;
        int psplit = 1;                                        // enough workgroups for two per CU, small neighbour-list stage
        while (psplit < 8 && (B * nsl * psplit < 512 || (N / psplit) * k * 2 > 16 * 1024) && N % (psplit * 2) == 0) psplit *= 2;
        const size_t lds = (size_t)N * ELDS_CS * sizeof(float) + align_up((size_t)((N + psplit - 1) / psplit) * k * 2, 16);
        if (lds <= 150 * 1024 && B * psplit <= edge_reduce_parts(P) && lds >= (size_t)2 * (256 / (ELDS_CS / 4)) * ELDS_CS * sizeof(double)) {
            auto kern = k == 20 ? edge_reduce_lds_kernel<20, true> : k <= 20 ? edge_reduce_lds_kernel<20, false> : k <= 32 ? edge_reduce_lds_kernel<32, false>
                      : k == 40 ? edge_reduce_lds_kernel<40, true> : edge_reduce_lds_kernel<40, false>;     // k = 40: BASELINE.json configs[4]
            if (lds > 64 * 1024) {
                hipError_t e = mlsp_lds_limit((const void*)kern, lds);
                if (e != hipSuccess) return (int)e;
            }
            hipLaunchKernelGGL(kern, dim3(B * nsl * psplit), dim3(256), lds, st, uv, idx, gamma, B, N, k, Cout, psplit, msel, argsel, s1, part);
            if (nparts_used) *nparts_used = B * psplit;
            return mlsp_launch_status();
        }
    }
#endif
    if (al && (Cout == 64 || Cout == 128 || Cout == 256)) {
        dim3 g(edge_reduce_parts(P)), b(256);
        if (Cout == 64) hipLaunchKernelGGL((edge_reduce_vec_kernel<16>), g, b, 0, st, uv, idx, gamma, P, N, k, msel, argsel, s1, part);
        else if (Cout == 128) hipLaunchKernelGGL((edge_reduce_vec_kernel<32>), g, b, 0, st, uv, idx, gamma, P, N, k, msel, argsel, s1, part);
        else hipLaunchKernelGGL((edge_reduce_vec_kernel<64>), g, b, 0, st, uv, idx, gamma, P, N, k, msel, argsel, s1, part);
        return mlsp_launch_status();
    }
    size_t lds = (size_t)8 * Cout * sizeof(double);
    if (lds > 64 * 1024) return MLSP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(edge_reduce_kernel, dim3(edge_reduce_parts(P)), dim3(256), lds, st, uv, idx, gamma, P, N, Cout, k,
                       msel, argsel, s1, part);
    return mlsp_launch_status();
}
int launch_edge_select_act(hipStream_t st, const float* msel, const float* uv, int P, int Cout, const float* scale,
                           const float* shift, int act, float slope, float* out, int ldo) {
    if (Cout % 4 == 0 && ldo % 4 == 0 && ((((uintptr_t)msel | (uintptr_t)uv | (uintptr_t)out | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0)) {
        size_t t4 = (size_t)P * Cout / 4;
        hipLaunchKernelGGL(edge_select_act_vec_kernel, dim3(ew_blocks2(t4)), dim3(256), 0, st, msel, uv, t4, Cout / 4, scale, shift,
                           act, slope, out, ldo);
        return mlsp_launch_status();
    }
    hipLaunchKernelGGL(edge_select_act_kernel, dim3(ew_blocks2((size_t)P * Cout)), dim3(256), 0, st, msel, uv, P, Cout, scale,
                       shift, act, slope, out, ldo);
    return mlsp_launch_status();
}
static bool edge_bwd_reduce_vec_ok(const float* dOut, const float* out, const float* msel, const float* uv, int Cout, const float* mean,
                                   const float* invstd, int lddo, int ldo) {
    return Cout % 4 == 0 && Cout <= 1024 && 256 % (Cout / 4) == 0 && lddo % 4 == 0 && ldo % 4 == 0 &&
           ((((uintptr_t)dOut | (uintptr_t)out | (uintptr_t)msel | (uintptr_t)uv | (uintptr_t)mean | (uintptr_t)invstd) & 15) == 0);
}
static bool edge_bwd_point_vec_ok(const float* dOut, const float* out, const float* uv, const float* s1, int Cout, const float* scale,
                                  const float* gz, const float* duv, int lddo, int ldo) {
    return Cout % 4 == 0 && lddo % 4 == 0 && ldo % 4 == 0 &&
           ((((uintptr_t)dOut | (uintptr_t)out | (uintptr_t)uv | (uintptr_t)s1 | (uintptr_t)gz | (uintptr_t)duv | (uintptr_t)scale) & 15) == 0);
}
static bool edge_bwd_gather_vec_ok(const float* gz, const uint8_t* argsel, const float* uv, int Cout, const float* duv) {
    return (((uintptr_t)uv | (uintptr_t)gz | (uintptr_t)duv) & 15) == 0 && (((uintptr_t)argsel) & 3) == 0 && (Cout == 64 || Cout == 128 || Cout == 256);
}
// all three backward passes take their vectorised form for these operands: they can leave the bound of duv as a by-product
// (`duv_amax` of the launchers below; the scalar forms ignore it)
bool edge_bwd_leaves_duv_bound(const float* dOut, const float* out, const float* msel, const float* uv, const float* s1, const uint8_t* argsel,
                               int Cout, const float* scale, const float* mean, const float* invstd, const float* gz, const float* duv,
                               int lddo, int ldo) {
    return edge_bwd_reduce_vec_ok(dOut, out, msel, uv, Cout, mean, invstd, lddo, ldo) &&
           edge_bwd_point_vec_ok(dOut, out, uv, s1, Cout, scale, gz, duv, lddo, ldo) && edge_bwd_gather_vec_ok(gz, argsel, uv, Cout, duv);
}
int launch_edge_bwd_reduce(hipStream_t st, const float* dOut, const float* out, const float* msel, const float* uv, int P,
                           int Cout, const float* mean, const float* invstd, int act, float slope, double* part, int lddo, int ldo,
                           float* duv_amax) {
    if (edge_bwd_reduce_vec_ok(dOut, out, msel, uv, Cout, mean, invstd, lddo, ldo)) {
        // NOTE: writes (P+1023)/1024 partial rows -- callers size `part` for (P+511)/512 and pass the count below
        hipLaunchKernelGGL(edge_bwd_reduce_vec_kernel, dim3((P + EVROWS - 1) / EVROWS), dim3(256), 0, st, dOut, out, msel, uv, P, Cout,
                           mean, invstd, act, slope, part, lddo, ldo, duv_amax);
        return mlsp_launch_status();
    }
    hipLaunchKernelGGL(edge_bwd_reduce_kernel, dim3((Cout + 63) / 64, (P + 511) / 512), dim3(256), 0, st, dOut, out, msel, uv,
                       P, Cout, mean, invstd, act, slope, part, lddo, ldo);
    return mlsp_launch_status();
}
int launch_edge_bwd_point(hipStream_t st, const float* dOut, const float* out, const float* uv, const float* s1, int P,
                          int Cout, int k, const float* scale, const float* mean, const float* invstd, const float* mean_dz,
                          const float* mean_dzy, int act, float slope, float* gz, float* duv, int lddo, int ldo, float* duv_amax) {
    if (edge_bwd_point_vec_ok(dOut, out, uv, s1, Cout, scale, gz, duv, lddo, ldo)) {
        size_t t4 = (size_t)P * Cout / 4;
        hipLaunchKernelGGL(edge_bwd_point_vec_kernel, dim3(ew_blocks2(t4)), dim3(256), 0, st, dOut, out, uv, s1, t4, Cout / 4, k, scale,
                           mean, invstd, mean_dz, mean_dzy, act, slope, gz, duv, lddo, ldo, duv_amax);
        return mlsp_launch_status();
    }
    hipLaunchKernelGGL(edge_bwd_point_kernel, dim3(ew_blocks2((size_t)P * Cout)), dim3(256), 0, st, dOut, out, uv, s1, P, Cout,
                       k, scale, mean, invstd, mean_dz, mean_dzy, act, slope, gz, duv, lddo, ldo);
    return mlsp_launch_status();
}
int launch_edge_bwd_gather(hipStream_t st, const float* gz, const uint8_t* argsel, const float* uv, const int* rev_off,
                           const int* rev_ent, int P, int N, int Cout, const float* scale, const float* mean,
                           const float* invstd, const float* mean_dz, const float* mean_dzy, float* duv, float* duv_amax) {
    if (edge_bwd_gather_vec_ok(gz, argsel, uv, Cout, duv)) {
        dim3 g((P + 3) / 4), b(256);
        if (Cout == 64) hipLaunchKernelGGL((edge_bwd_gather_vec_kernel<16>), g, b, 0, st, gz, argsel, uv, rev_off, rev_ent, P, N, scale, mean, invstd, mean_dz, mean_dzy, duv, duv_amax);
        else if (Cout == 128) hipLaunchKernelGGL((edge_bwd_gather_vec_kernel<32>), g, b, 0, st, gz, argsel, uv, rev_off, rev_ent, P, N, scale, mean, invstd, mean_dz, mean_dzy, duv, duv_amax);
        else hipLaunchKernelGGL((edge_bwd_gather_vec_kernel<64>), g, b, 0, st, gz, argsel, uv, rev_off, rev_ent, P, N, scale, mean, invstd, mean_dz, mean_dzy, duv, duv_amax);
        return mlsp_launch_status();
    }
    hipLaunchKernelGGL(edge_bwd_gather_kernel, dim3((P + 3) / 4), dim3(256), 0, st, gz, argsel, uv, rev_off, rev_ent, P, N,
                       Cout, scale, mean, invstd, mean_dz, mean_dzy, duv);
    return mlsp_launch_status();
}
int launch_graph_feature_fwd(hipStream_t st, const float* x, const int* idx, int P, int N, int C, int k, float* F) {
    hipLaunchKernelGGL(graph_feature_fwd_kernel, dim3(ew_blocks2((size_t)P * k * C)), dim3(256), 0, st, x, idx, P, N, C, k, F);
    return mlsp_launch_status();
}
int launch_graph_feature_bwd(hipStream_t st, const float* dF, const int* rev_off, const int* rev_ent, int P, int N, int C,
                             int k, float* dx) {
    hipLaunchKernelGGL(graph_feature_bwd_kernel, dim3(ew_blocks2((size_t)P * C)), dim3(256), 0, st, dF, rev_off, rev_ent, P, N,
                       C, k, dx);
    return mlsp_launch_status();
}
