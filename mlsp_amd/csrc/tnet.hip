// Fused per-edge stage of the input transform net (T-Net), dgcnn branch:
//   PointDA/model_utils.py:111-115   x = conv2d1(x0) ; x = conv2d2(x) ; x = x.max(dim=-1)
// i.e.  t_i = max_s LeakyReLU(BN2( W2 . LeakyReLU(BN1( W1 . [x_j - x_i ; x_i] )) ))   over the k edges of point i.
//
// The first conv folds exactly like EdgeConv (edge.hip): hpre_e = u_j + v_i with [u|v] = x [W1a ; W1b-W1a]^T.
// The second conv (64 -> 128 per edge) cannot be folded; it is the one genuine per-edge contraction of the
// network and runs here on the matrix cores with everything LDS-resident:
//   workgroup tile = TP points x k edges (<= 128 rows; 6 x 20 = 120 at k = 20)
//   Hs [64][129]  hpre, k-major     (gathered from the L2-resident u rows; BN1 + LeakyReLU applied at fragment read)
//   Ws [128][65]  W2                (odd stride: conflict-free for both the Z and the dH products)
//   Zs [128][129] Z = H W2^T        (then the per-point max/min over its k rows, BN2 statistics)
// Neither [E,64] nor [E,128] ever reaches HBM in the forward pass.  Backward recomputes H and Z per tile, forms
// dZ in LDS from the closed-form BN2 backward, and runs dH = dZ W2 and dW2 += dZ^T H on the matrix cores; only
// dh' = dH * act'(a) ([E,64]) is written out because BN1's backward needs its global sums before it can be
// folded onto the points (tnet_edge_bwd2_kernel, a gather over the reverse neighbour index).
#include "common.h"
#include <math.h>
#include <cstdlib>
#include <type_traits>

#define TN_C1 64
#define TN_C2 128
#define TN_ROWS 128
#define TN_SH 129
#define TN_SZ 129
#define TN_SW 65
#define TN_MAXTP 8
#define TN_LDS_FLOATS (TN_C1 * TN_SH + TN_C2 * TN_SW + TN_ROWS * TN_SZ + 4 * TN_C1 + 4 * TN_C2 + TN_ROWS + TN_MAXTP * TN_C2 + TN_MAXTP * TN_C2 / 4)

__device__ __forceinline__ float lrelu(float a, float slope) { return a > 0.f ? a : a * slope; }

// Tiles never straddle clouds (tpc = ceil(N / TP) tiles per cloud) and the persistent workgroups of one XCD (blockIdx % 8)
// walk only the clouds c with c % 8 == that label, so the neighbour rows a tile gathers stay in that XCD's L2.
// m-th tile of this workgroup -> first point index and point count; returns false when the workgroup is done.
__device__ __forceinline__ bool tn_tile(int m, int B, int N, int TP, int& pt0, int& npts) {
    const int tpc = (N + TP - 1) / TP;
    int cloud, tic;
    if ((B & 7) == 0 && (gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7, q = blockIdx.x >> 3, per = gridDim.x >> 3;
        const int L = q + m * per;
        if (L >= (B >> 3) * tpc) return false;
        cloud = x + 8 * (L / tpc); tic = L % tpc;
    } else {
        const int L = blockIdx.x + m * gridDim.x;
        if (L >= B * tpc) return false;
        cloud = L / tpc; tic = L % tpc;
    }
    pt0 = cloud * N + tic * TP;
    npts = min(TP, N - tic * TP);
    return true;
}

// Gather of hpre = u_j + v_i for the rows of one tile (512 threads: 4 per row), split so that it can run one tile AHEAD of
// the MFMA work: the neighbour index of tile m+2 and the u/v rows of tile m+1 are in flight while tile m is computed (a
// single workgroup per CU: nothing else would hide the two dependent L2 round trips of index -> row).
struct TnRows { f32x4 u[4], v[4]; };

// global source row of this thread's (point, slot) in a tile, -1 when the row is padding
__device__ __forceinline__ int tn_row_index(const int* __restrict__ idx, int pt0, int npts, int k, int N, int tid) {
    const int row = tid >> 2;
    const int pt = row / k, s = row - pt * k;
    if (pt >= npts) return -1;
    const int i = pt0 + pt;
    return (i / N) * N + idx[(size_t)i * k + s];
}
__device__ __forceinline__ void tn_load_rows(TnRows& r, const float* __restrict__ uv, int j, int pt0, int k, int tid) {
    const int row = tid >> 2, qt = tid & 3;
    if (j >= 0) {
        const int i = pt0 + row / k;
        const f32x4* ur = (const f32x4*)(uv + (size_t)j * 2 * TN_C1 + 16 * qt);
        const f32x4* vr = (const f32x4*)(uv + (size_t)i * 2 * TN_C1 + TN_C1 + 16 * qt);
#pragma unroll
        for (int q = 0; q < 4; ++q) { r.u[q] = ur[q]; r.v[q] = vr[q]; }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { r.u[q] = f32x4{0.f, 0.f, 0.f, 0.f}; r.v[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
}
__device__ __forceinline__ void tn_store_h(float* __restrict__ Hs, const TnRows& r, int tid) {
    const int row = tid >> 2, qt = tid & 3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 hv = r.u[q] + r.v[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) Hs[(16 * qt + 4 * q + e) * TN_SH + row] = hv[e];
    }
}

// Z tile = H W2^T  (H = LeakyReLU(scale1*hpre + shift1), zero on invalid rows) -> Zs[row][o].
// 8 waves: wave (wm = 0..3, wn = 0..1) owns rows 32*wm.. and columns 64*wn..
__device__ __forceinline__ void tn_compute_z(const float* __restrict__ Hs, const float* __restrict__ Ws,
                                             const float* __restrict__ S1, float* __restrict__ Zs, int nvalid, float slope,
                                             int wm, int wn, int l31, int h) {
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int r0 = wm * 32 + l31;
    const bool v0 = r0 < nvalid;
    const int o0 = wn * 64 + l31, o1 = o0 + 32;
#pragma unroll 8
    for (int t = 0; t < TN_C1 / 2; ++t) {
        const int c = 2 * t + h;
        const float sc = S1[c], sh = S1[TN_C1 + c];
        float a0 = v0 ? lrelu(fmaf(Hs[c * TN_SH + r0], sc, sh), slope) : 0.f;
        float b0 = Ws[o0 * TN_SW + c], b1 = Ws[o1 * TN_SW + c];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            Zs[row * TN_SZ + wn * 64 + j * 32 + l31] = acc[j][r];
        }
}

struct TnetFwdArgs {
    const float* uv; const int* idx; const float* bn1; const float* W2; const float* gamma2;
    float* zsel; uint8_t* argsel; double* part;    // part: [gridDim.x][2][128]
    int P, N, k, TP, ntiles; float slope;
};

__global__ __launch_bounds__(512) void tnet_edge_fwd_kernel(TnetFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Hs = sm;
    float* Ws = Hs + TN_C1 * TN_SH;
    float* Zs = Ws + TN_C2 * TN_SW;
    float* S1 = Zs + TN_ROWS * TN_SZ;               // scale1[64], shift1[64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    for (int e = tid; e < TN_C2 * TN_C1; e += 512) Ws[(e >> 6) * TN_SW + (e & 63)] = p.W2[e];
    if (tid < 2 * TN_C1) S1[tid] = p.bn1[tid];
    const int o = tid & 127;
    const bool use_max = p.gamma2[o] >= 0.f;
    double ssum = 0.0, ssq = 0.0;
    int pt0 = 0, npts = 0, pt0n = 0, nptsn = 0, pt0nn = 0, nptsnn = 0;
    const int Bc = p.P / p.N;
    bool have = tn_tile(0, Bc, p.N, p.TP, pt0, npts);
    TnRows rows;
    if (have) tn_load_rows(rows, p.uv, tn_row_index(p.idx, pt0, npts, p.k, p.N, tid), pt0, p.k, tid);
    bool haven = have && tn_tile(1, Bc, p.N, p.TP, pt0n, nptsn);
    int jn = haven ? tn_row_index(p.idx, pt0n, nptsn, p.k, p.N, tid) : -1;
    for (int m = 0; have; ++m) {
        __syncthreads();
        tn_store_h(Hs, rows, tid);
        __syncthreads();
        // tile m+1's rows and tile m+2's neighbour index travel under the MFMA + epilogue of tile m
        if (haven) tn_load_rows(rows, p.uv, jn, pt0n, p.k, tid);
        const bool havenn = haven && tn_tile(m + 2, Bc, p.N, p.TP, pt0nn, nptsnn);
        const int jnn = havenn ? tn_row_index(p.idx, pt0nn, nptsnn, p.k, p.N, tid) : -1;
        tn_compute_z(Hs, Ws, S1, Zs, npts * p.k, p.slope, wm, wn, l31, h);
        __syncthreads();
        for (int item = tid; item < npts * TN_C2; item += 512) {
            const int pt = item >> 7;
            const size_t i = (size_t)pt0 + pt;
            const float* z = Zs + (pt * p.k) * TN_SZ + o;
            float best = z[0], s1 = z[0], s2 = z[0] * z[0];
            int bs = 0;
            for (int s0 = 1; s0 < p.k; s0 += 4) {            // four LDS reads in flight (one wave per SIMD pair: latency shows)
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = z[min(s0 + u, p.k - 1) * TN_SZ];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (s0 + u < p.k) {
                        s1 += v[u]; s2 = fmaf(v[u], v[u], s2);
                        const bool take = use_max ? (v[u] > best) : (v[u] < best);
                        best = take ? v[u] : best; bs = take ? s0 + u : bs;
                    }
                }
            }
            p.zsel[i * TN_C2 + o] = best;
            p.argsel[i * TN_C2 + o] = (uint8_t)bs;
            ssum += s1; ssq += s2;
        }
        have = haven; pt0 = pt0n; npts = nptsn;
        haven = havenn; pt0n = pt0nn; nptsn = nptsnn; jn = jnn;
    }
    __syncthreads();
    double* red = (double*)Zs;                       // [2][512]
    red[tid] = ssum; red[512 + tid] = ssq;
    __syncthreads();
    if (tid < TN_C2) {
        p.part[((size_t)blockIdx.x * 2 + 0) * TN_C2 + tid] = red[tid] + red[tid + 128] + red[tid + 256] + red[tid + 384];
        p.part[((size_t)blockIdx.x * 2 + 1) * TN_C2 + tid] = red[512 + tid] + red[512 + tid + 128] + red[512 + tid + 256] + red[512 + tid + 384];
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Forward, register-resident (round 2).  The Z tile never touches LDS: a point's K edge rows are laid out so that the per-point
// max / arg-max and the BN2 sums come straight out of the MFMA accumulators.
//   workgroup = 4 waves, tile = 160 rows = 5 MFMA row blocks = 160 / K points (K = 20: 8 points, K = 40: 4 points); K % 4 == 0 puts
//   every aligned group of 4 rows inside ONE point, and a 32x32 accumulator holds its rows in exactly such groups
//   (register 4q..4q+3 of lane half h <-> rows 8q + 4h .. +3), so "which point / which slot" is a compile-time function of
//   (block, q) and the lane half.  160 rows is the smallest row count that is a multiple of both 32 and K in {20, 40}.
//   wave w owns output columns 32w..32w+31 for all 5 row blocks: 80 accumulator registers, its W2 fragment (32 registers) stays in
//   registers for the whole kernel.  LDS holds only the ACTIVATED H tile [160][64] (row-major, pitch 66 floats: the gathered rows
//   are stored as they arrive and one ds_read_b64 feeds two MFMA steps, conflict-free): 42 KB -> three workgroups per CU, whose
//   gather / MFMA / epilogue phases overlap each other (the 141 KB, one-workgroup-per-CU kernel above had nothing to overlap with).
#define TF_ROWS 160
#define TF_PITCH 66
template <int K>
__global__ __launch_bounds__(256, 2) void tnet_edge_fwd2_kernel(TnetFwdArgs p) {
    constexpr int PT = TF_ROWS / K;                  // points per tile
    constexpr int GP = K / 4;                        // aligned 4-row groups per point
    __shared__ __attribute__((aligned(16))) float Hs[TF_ROWS * TF_PITCH];
    __shared__ float S1[2 * TN_C1];
    __shared__ __attribute__((aligned(16))) float Vs[PT * TN_C1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int o = 32 * wave + l31;
    if (tid < 2 * TN_C1) S1[tid] = p.bn1[tid];
    float2 w2[16];                                    // B fragments: W2[o][4m + 2h], W2[o][4m + 2h + 1]
#pragma unroll
    for (int m = 0; m < 16; ++m) w2[m] = *(const float2*)(p.W2 + (size_t)o * TN_C1 + 4 * m + 2 * h);
    const bool use_max = p.gamma2[o] >= 0.f;
    double ssum = 0.0, ssq = 0.0;
    const int Bc = p.P / p.N;
    int pt0 = 0, npts = 0, pt0n = 0, nptsn = 0;
    __syncthreads();
    // neighbour indices of the rows this thread gathers in a tile (-1: padding); tile m+1's are fetched under tile m's MFMA phase so
    // that only ONE dependent global round trip (the rows themselves) sits on a tile's critical path
    // The index loads are UNCONDITIONAL and their values stay raw until the next tile's gather uses them (okm: which of the three are real
    // rows, jbase: the cloud's first point): `ok ? base + idx[..] : -1` makes the compiler wait for the load -- and for every row load issued
    // before it -- right there, three memory round trips in a row per tile.
    auto tile_rows = [&](int t0, int np, int (&jraw)[3], int& okm_) {
        okm_ = 0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int item = tid + 256 * u, row = item >> 2;
            const int pt = row / K, s = row - pt * K;
            const bool ok = item < TF_ROWS * 4 && pt < np;
            okm_ |= ok ? 1 << u : 0;
#ifdef TF_PROBE_NOGATHER
            jraw[u] = t0 + pt - (t0 / p.N) * p.N;
#else
            jraw[u] = p.idx[ok ? (size_t)(t0 + pt) * K + s : (size_t)0];
#endif
        }
    };
    bool have = tn_tile(0, Bc, p.N, PT, pt0, npts);
    int jraw[3] = {0, 0, 0}, okm = 0, jbase = 0;
    if (have) { tile_rows(pt0, npts, jraw, okm); jbase = (pt0 / p.N) * p.N; }
    for (int m = 0; have; ++m) {
        int jrow[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) jrow[u] = (okm >> u) & 1 ? jbase + jraw[u] : -1;
        // ---- gather: 160 rows x 4 quarter rows (16 channels) of u_j over 256 threads, everything in flight before the first use;
        //      the centre term v_i is the same for the K rows of a point: PT rows staged through LDS (Vs) instead of 160
        f32x4 ur[3][4];
        f32x4 vstage = {0.f, 0.f, 0.f, 0.f};
        if (tid < PT * 16 && (tid >> 4) < npts) vstage = *(const f32x4*)(p.uv + (size_t)(pt0 + (tid >> 4)) * 2 * TN_C1 + TN_C1 + 4 * (tid & 15));
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int qt = (tid + 256 * u) & 3;
            if (jrow[u] >= 0) {
                const f32x4* up = (const f32x4*)(p.uv + (size_t)jrow[u] * 2 * TN_C1 + 16 * qt);
#pragma unroll
                for (int e = 0; e < 4; ++e) ur[u][e] = up[e];
            }
        }
        const bool haven = tn_tile(m + 1, Bc, p.N, PT, pt0n, nptsn);
        int jnext[3], okn;
        tile_rows(haven ? pt0n : pt0, haven ? nptsn : 0, jnext, okn);
        __syncthreads();                              // every wave is done reading the previous tile's H (and Vs)
        if (tid < PT * 16) *(f32x4*)(Vs + 4 * tid) = vstage;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int item = tid + 256 * u, row = item >> 2, qt = item & 3;
            if (item < TF_ROWS * 4) {
                float* dst = Hs + row * TF_PITCH + 16 * qt;
                const float* vs = Vs + (row / K) * TN_C1 + 16 * qt;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 vv = *(const f32x4*)(vs + 4 * e);
                    float hv[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int ch = 16 * qt + 4 * e + c;
                        hv[c] = jrow[u] >= 0 ? lrelu(fmaf(ur[u][e][c] + vv[c], S1[ch], S1[TN_C1 + ch]), p.slope) : 0.f;
                    }
                    *(float2*)(dst + 4 * e) = make_float2(hv[0], hv[1]);
                    *(float2*)(dst + 4 * e + 2) = make_float2(hv[2], hv[3]);
                }
            }
        }
        __syncthreads();
        // ---- Z = H W2^T for this wave's 32 columns, one row block after the other; the register epilogue of block b - 1 (per point
        //      max / min over its K rows with the first arg-max, BN2 sums: ~6 vector instructions per element) is issued BETWEEN the
        //      MFMAs of block b.  A vector instruction that has to squeeze in between another wave's back-to-back MFMAs waits about one
        //      MFMA slot (gemm.hip, gemm_out_fast); inside the issuing wave's own MFMA stream it is free: the matrix pipe is busy for 64
        //      clocks per MFMA and the wave's next instructions issue meanwhile.  Only the last block's epilogue stays exposed.
        //      (Accumulation order per block and the order of the sums are those of the all-blocks-at-once form: bitwise the same.)
        //      Measured: forward op 228 -> 219 us.  Also moving BN1 + activation from the staging pass to the fragment reads (6 more
        //      vector instructions and three more LDS reads per MFMA pair) loses that again: 233 us.
        f32x16 acc[5];
        float best[PT];
        int bslot[PT];
#pragma unroll
        for (int q = 0; q < PT; ++q) { best[q] = use_max ? -INFINITY : INFINITY; bslot[q] = 0; }
        float s1 = 0.f, s2 = 0.f;
        auto epi = [&](int b, int r) {                                 // (b, r) are compile-time after unrolling
            const int q = r >> 2, e = r & 3;
            const int G0 = 8 * b + 2 * q;                              // 4-row group index of lane half 0 (half 1: G0 + 1)
            const int p0 = G0 / GP, r0 = G0 % GP;
            const bool cross = (r0 == GP - 1);                         // half 1's group belongs to the next point
            const float v = acc[b][r];
            s1 += v; s2 = fmaf(v, v, s2);
            if (!cross) {
                const int slot = 4 * (r0 + h) + e;
                const bool take = use_max ? (v > best[p0]) : (v < best[p0]);
                best[p0] = take ? v : best[p0]; bslot[p0] = take ? slot : bslot[p0];
            } else {
                {
                    const bool take = h == 0 && (use_max ? (v > best[p0]) : (v < best[p0]));
                    best[p0] = take ? v : best[p0]; bslot[p0] = take ? 4 * (GP - 1) + e : bslot[p0];
                }
                if (p0 + 1 < PT) {
                    const bool take = h == 1 && (use_max ? (v > best[p0 + 1]) : (v < best[p0 + 1]));
                    best[p0 + 1] = take ? v : best[p0 + 1]; bslot[p0 + 1] = take ? e : bslot[p0 + 1];
                }
            }
        };
#ifndef TF_PROBE_NOMFMA
#pragma unroll
        for (int b = 0; b < 5; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
            const float* hrow = Hs + (32 * b + l31) * TF_PITCH + 2 * h;
            float2 a_cur = *(const float2*)hrow;
#pragma unroll
            for (int mm = 0; mm < 16; ++mm) {
                float2 a_nxt = a_cur;
                if (mm + 1 < 16) a_nxt = *(const float2*)(hrow + 4 * (mm + 1));
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, w2[mm].x, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, w2[mm].y, acc[b], 0, 0, 0);
                if (b > 0) epi(b - 1, mm);
                __builtin_amdgcn_sched_barrier(0);
                a_cur = a_nxt;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) epi(4, r);
#else
#pragma unroll
        for (int b = 0; b < 5; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
            acc[b][0] = Hs[(32 * b + l31) * TF_PITCH + h] * w2[b].x;
        }
#pragma unroll
        for (int b = 0; b < 5; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) epi(b, r);
#endif
        ssum += s1; ssq += s2;
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            const float ob = __shfl_xor(best[q], 32, 64);
            const int os = __shfl_xor(bslot[q], 32, 64);
            const bool better = use_max ? (ob > best[q]) : (ob < best[q]);
            if (better || (ob == best[q] && os < bslot[q])) { best[q] = ob; bslot[q] = os; }
            if (h == 0 && q < npts) {
                p.zsel[(size_t)(pt0 + q) * TN_C2 + o] = best[q];
                p.argsel[(size_t)(pt0 + q) * TN_C2 + o] = (uint8_t)bslot[q];
            }
        }
        have = haven; pt0 = pt0n; npts = nptsn;
#pragma unroll
        for (int u = 0; u < 3; ++u) jraw[u] = jnext[u];
        okm = okn; jbase = (pt0n / p.N) * p.N;
    }
    // per-workgroup BN2 partials: the two lane halves hold different rows of the same column
    ssum += __shfl_xor(ssum, 32, 64);
    ssq += __shfl_xor(ssq, 32, 64);
    if (h == 0) {
        p.part[((size_t)blockIdx.x * 2 + 0) * TN_C2 + o] = ssum;
        p.part[((size_t)blockIdx.x * 2 + 1) * TN_C2 + o] = ssq;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Forward, round 3 (launches of >= 1024 tiles, see launch_tnet_edge_fwd): the same tile walk, gather, register epilogue and outputs as
// tnet_edge_fwd2_kernel, with the 64 -> 128 contraction as
// fp32-ACCURATE products on the bf16 matrix cores (the split of gemm.hip's gemm_split_kernel: x = a + b + c exactly, six piece products,
// fp32 accumulation): 24 v_mfma_f32_32x32x16_bf16 per row block instead of 32 v_mfma_f32_32x32x2_f32 at twice the cycles each.  The
// activated H tile is split once, on its way into LDS (three bf16 images [160][64 + 8 pad]: 69 KB, still two workgroups per CU); W2's
// fragments are split once per workgroup and live in 48 registers.
typedef __bf16 tbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tbf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define TF3_PITCH 144                          // bytes per image row: 16 rows x 36 dwords cover the 64 banks exactly once (ds_read_b128)
#define TF3_PLANE (TF_ROWS * TF3_PITCH)        // 23,040 B
__device__ __forceinline__ uint32_t tn_cvt_pk(float lo, float hi) {
    const f32x2 v = {lo, hi};
    const tbf16x2 b = __builtin_convertvector(v, tbf16x2);
    return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ void tn_split2(float x0, float x1, uint32_t& a, uint32_t& b, uint32_t& c) {
    a = tn_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(a << 16), r1 = x1 - __uint_as_float(a & 0xffff0000u);
    b = tn_cvt_pk(r0, r1);
    c = tn_cvt_pk(r0 - __uint_as_float(b << 16), r1 - __uint_as_float(b & 0xffff0000u));
}
// value of the same lane of the OTHER 32-lane half (v_permlane32_swap: no LDS crossbar, no address registers)
__device__ __forceinline__ unsigned tn_xhalf(unsigned v, int h) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return h ? r[0] : r[1];
}
__device__ __forceinline__ double tn_xhalf_d(double v, int h) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = tn_xhalf((unsigned)u, h), hi = tn_xhalf((unsigned)(u >> 32), h);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ tbf16x8 tn_pack8(const uint32_t (&p)[4]) {
    const u32x4 v = {p[0], p[1], p[2], p[3]};
    return __builtin_bit_cast(tbf16x8, v);
}
// ONEP (`precision` = MLSP_PREC_BF16: operands ROUNDED to bf16): only the leading pieces are staged and multiplied -- one MFMA per k16 step
// instead of six; everything else (walk, gather, epilogue, sums) is the same code.
template <int K, bool ONEP = false>
__global__ __launch_bounds__(256, 2) void tnet_edge_fwd3_kernel(TnetFwdArgs p) {
    constexpr int PT = TF_ROWS / K;                  // points per tile
    constexpr int GP = K / 4;                        // aligned 4-row groups per point
    __shared__ __attribute__((aligned(16))) char Hs3[3 * TF3_PLANE];          // activated H tile as three bf16 images [160][64 + 8 pad]
    __shared__ float S1[2 * TN_C1];
    __shared__ __attribute__((aligned(16))) float Vs[PT * TN_C1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int o = 32 * wave + l31;
    if (tid < 2 * TN_C1) S1[tid] = p.bn1[tid];
    tbf16x8 w2p[4][3];                                // B fragments of the four k16 steps, three pieces each: W2[o][16 s + 8 h .. + 7]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const f32x4 lo = *(const f32x4*)(p.W2 + (size_t)o * TN_C1 + 16 * s + 8 * h), hi = *(const f32x4*)(p.W2 + (size_t)o * TN_C1 + 16 * s + 8 * h + 4);
        uint32_t pa[4], pb[4], pc[4];
        tn_split2(lo[0], lo[1], pa[0], pb[0], pc[0]); tn_split2(lo[2], lo[3], pa[1], pb[1], pc[1]);
        tn_split2(hi[0], hi[1], pa[2], pb[2], pc[2]); tn_split2(hi[2], hi[3], pa[3], pb[3], pc[3]);
        w2p[s][0] = tn_pack8(pa); w2p[s][1] = tn_pack8(pb); w2p[s][2] = tn_pack8(pc);
    }
    const bool use_max = p.gamma2[o] >= 0.f;
    double ssum = 0.0, ssq = 0.0;
    const int Bc = p.P / p.N;
    int pt0 = 0, npts = 0, pt0n = 0, nptsn = 0;
    __syncthreads();
    // neighbour indices of the rows this thread gathers in a tile (-1: padding); tile m+1's are fetched under tile m's MFMA phase so
    // that only ONE dependent global round trip (the rows themselves) sits on a tile's critical path
    // The index loads are UNCONDITIONAL and their values stay raw until the next tile's gather uses them (okm: which of the three are real
    // rows, jbase: the cloud's first point): `ok ? base + idx[..] : -1` makes the compiler wait for the load -- and for every row load issued
    // before it -- right there, three memory round trips in a row per tile.
    auto tile_rows = [&](int t0, int np, int (&jraw)[3], int& okm_) {
        okm_ = 0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int item = tid + 256 * u, row = item >> 2;
            const int pt = row / K, s = row - pt * K;
            const bool ok = item < TF_ROWS * 4 && pt < np;
            okm_ |= ok ? 1 << u : 0;
#ifdef TF_PROBE_NOGATHER
            jraw[u] = t0 + pt - (t0 / p.N) * p.N;
#else
            jraw[u] = p.idx[ok ? (size_t)(t0 + pt) * K + s : (size_t)0];
#endif
        }
    };
#ifdef TF3_STAMPS
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define TF3_STAMP(i_) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsum[i_] += t_ - tprev; tprev = t_; } while (0)
#else
#define TF3_STAMP(i_)
#endif
    bool have = tn_tile(0, Bc, p.N, PT, pt0, npts);
    int jraw[3] = {0, 0, 0}, okm = 0, jbase = 0;
    if (have) { tile_rows(pt0, npts, jraw, okm); jbase = (pt0 / p.N) * p.N; }
    for (int m = 0; have; ++m) {
        int jrow[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) jrow[u] = (okm >> u) & 1 ? jbase + jraw[u] : -1;
        // ---- gather: 160 rows x 4 quarter rows (16 channels) of u_j over 256 threads, everything in flight before the first use;
        //      the centre term v_i is the same for the K rows of a point: PT rows staged through LDS (Vs) instead of 160
        f32x4 ur[3][4];
        f32x4 vstage = {0.f, 0.f, 0.f, 0.f};
        if (tid < PT * 16 && (tid >> 4) < npts) vstage = *(const f32x4*)(p.uv + (size_t)(pt0 + (tid >> 4)) * 2 * TN_C1 + TN_C1 + 4 * (tid & 15));
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int qt = (tid + 256 * u) & 3;
            if (jrow[u] >= 0) {
                const f32x4* up = (const f32x4*)(p.uv + (size_t)jrow[u] * 2 * TN_C1 + 16 * qt);
#pragma unroll
                for (int e = 0; e < 4; ++e) ur[u][e] = up[e];
            }
        }
        const bool haven = tn_tile(m + 1, Bc, p.N, PT, pt0n, nptsn);
        int jnext[3], okn;
        tile_rows(haven ? pt0n : pt0, haven ? nptsn : 0, jnext, okn);
        TF3_STAMP(0);
        __syncthreads();                              // every wave is done reading the previous tile's H (and Vs)
        TF3_STAMP(1);
        if (tid < PT * 16) *(f32x4*)(Vs + 4 * tid) = vstage;
        __syncthreads();
        TF3_STAMP(2);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int item = tid + 256 * u, row = item >> 2, qt = item & 3;
            if (item < TF_ROWS * 4) {
                char* dst = Hs3 + row * TF3_PITCH + 32 * qt;          // 16 channels = 32 bytes of each image
                const float* vs = Vs + (row / K) * TN_C1 + 16 * qt;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 vv = *(const f32x4*)(vs + 4 * e);
                    float hv[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int ch = 16 * qt + 4 * e + c;
                        hv[c] = jrow[u] >= 0 ? lrelu(fmaf(ur[u][e][c] + vv[c], S1[ch], S1[TN_C1 + ch]), p.slope) : 0.f;
                    }
                    uint32_t pa[2], pb[2], pc[2];
                    tn_split2(hv[0], hv[1], pa[0], pb[0], pc[0]);
                    tn_split2(hv[2], hv[3], pa[1], pb[1], pc[1]);
                    *(uint2*)(dst + 8 * e) = make_uint2(pa[0], pa[1]);                      // four channels = 8 bytes of each image
                    if (!ONEP) {
                        *(uint2*)(dst + TF3_PLANE + 8 * e) = make_uint2(pb[0], pb[1]);
                        *(uint2*)(dst + 2 * TF3_PLANE + 8 * e) = make_uint2(pc[0], pc[1]);
                    }
                }
            }
        }
        TF3_STAMP(3);
        __syncthreads();
        TF3_STAMP(4);
        // ---- Z = H W2^T for this wave's 32 columns, one row block after the other; the register epilogue of block b - 1 (per point
        //      max / min over its K rows with the first arg-max, BN2 sums: ~6 vector instructions per element) is issued BETWEEN the
        //      MFMAs of block b.  A vector instruction that has to squeeze in between another wave's back-to-back MFMAs waits about one
        //      MFMA slot (gemm.hip, gemm_out_fast); inside the issuing wave's own MFMA stream it is free: the matrix pipe is busy for 64
        //      clocks per MFMA and the wave's next instructions issue meanwhile.  Only the last block's epilogue stays exposed.
        //      (Accumulation order per block and the order of the sums are those of the all-blocks-at-once form: bitwise the same.)
        //      Measured: forward op 228 -> 219 us.  Also moving BN1 + activation from the staging pass to the fragment reads (6 more
        //      vector instructions and three more LDS reads per MFMA pair) loses that again: 233 us.
        f32x16 acc[5];
        float best[PT];
        int bslot[PT];
#pragma unroll
        for (int q = 0; q < PT; ++q) { best[q] = use_max ? -INFINITY : INFINITY; bslot[q] = 0; }
        float s1 = 0.f, s2 = 0.f;
        auto epi = [&](int b, int r) {                                 // (b, r) are compile-time after unrolling
            const int q = r >> 2, e = r & 3;
            const int G0 = 8 * b + 2 * q;                              // 4-row group index of lane half 0 (half 1: G0 + 1)
            const int p0 = G0 / GP, r0 = G0 % GP;
            const bool cross = (r0 == GP - 1);                         // half 1's group belongs to the next point
            const float v = acc[b][r];
            s1 += v; s2 = fmaf(v, v, s2);
            if (!cross) {
                const int slot = 4 * (r0 + h) + e;
                const bool take = use_max ? (v > best[p0]) : (v < best[p0]);
                best[p0] = take ? v : best[p0]; bslot[p0] = take ? slot : bslot[p0];
            } else {
                {
                    const bool take = h == 0 && (use_max ? (v > best[p0]) : (v < best[p0]));
                    best[p0] = take ? v : best[p0]; bslot[p0] = take ? 4 * (GP - 1) + e : bslot[p0];
                }
                if (p0 + 1 < PT) {
                    const bool take = h == 1 && (use_max ? (v > best[p0 + 1]) : (v < best[p0 + 1]));
                    best[p0 + 1] = take ? v : best[p0 + 1]; bslot[p0 + 1] = take ? e : bslot[p0 + 1];
                }
            }
        };
#pragma unroll
        for (int b = 0; b < 5; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
            const char* hrow = Hs3 + (32 * b + l31) * TF3_PITCH + 16 * h;      // this lane's row of the block, k = 8 h .. of every k16 step
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                tbf16x8 a_cur[3];
                if constexpr (ONEP) {
                    a_cur[0] = *(const tbf16x8*)(hrow + 32 * s);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0], w2p[s][0], acc[b], 0, 0, 0);
                    if (b > 0) { epi(b - 1, 4 * s); epi(b - 1, 4 * s + 1); epi(b - 1, 4 * s + 2); epi(b - 1, 4 * s + 3); }
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) a_cur[q] = *(const tbf16x8*)(hrow + q * TF3_PLANE + 32 * s);
                // six piece products, smallest first; the register epilogue of block b - 1 rides between them (four entries per k16 step)
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1], w2p[s][1], acc[b], 0, 0, 0);
                if (b > 0) epi(b - 1, 4 * s);
                __builtin_amdgcn_sched_barrier(0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0], w2p[s][2], acc[b], 0, 0, 0);
                if (b > 0) epi(b - 1, 4 * s + 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[2], w2p[s][0], acc[b], 0, 0, 0);
                if (b > 0) epi(b - 1, 4 * s + 2);
                __builtin_amdgcn_sched_barrier(0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0], w2p[s][1], acc[b], 0, 0, 0);
                if (b > 0) epi(b - 1, 4 * s + 3);
                __builtin_amdgcn_sched_barrier(0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1], w2p[s][0], acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0], w2p[s][0], acc[b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        TF3_STAMP(5);
#pragma unroll
        for (int r = 0; r < 16; ++r) epi(4, r);
        ssum += s1; ssq += s2;
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            const float ob = __uint_as_float(tn_xhalf(__float_as_uint(best[q]), h));
            const int os = (int)tn_xhalf((unsigned)bslot[q], h);
            const bool better = use_max ? (ob > best[q]) : (ob < best[q]);
            if (better || (ob == best[q] && os < bslot[q])) { best[q] = ob; bslot[q] = os; }
            if (h == 0 && q < npts) {
                p.zsel[(size_t)(pt0 + q) * TN_C2 + o] = best[q];
                p.argsel[(size_t)(pt0 + q) * TN_C2 + o] = (uint8_t)bslot[q];
            }
        }
        TF3_STAMP(6);
        have = haven; pt0 = pt0n; npts = nptsn;
#pragma unroll
        for (int u = 0; u < 3; ++u) jraw[u] = jnext[u];
        okm = okn; jbase = (pt0n / p.N) * p.N;
    }
    // per-workgroup BN2 partials: the two lane halves hold different rows of the same column
    ssum += tn_xhalf_d(ssum, h);
    ssq += tn_xhalf_d(ssq, h);
    if (h == 0) {
        p.part[((size_t)blockIdx.x * 2 + 0) * TN_C2 + o] = ssum;
        p.part[((size_t)blockIdx.x * 2 + 1) * TN_C2 + o] = ssq;
    }
#ifdef TF3_STAMPS
    if (tid == 0) {                                            // diagnostic build only: overwrites this workgroup's BN2 partial
        unsigned long long* o2 = (unsigned long long*)(p.part + (size_t)blockIdx.x * 2 * TN_C2);
        o2[0] = 0x5446335354414d50ull;
        for (int q = 0; q < 8; ++q) o2[1 + q] = tsum[q];
    }
#endif
}
#undef TF3_STAMP


// t = act(scale2 * zsel + shift2)        [P][128]
__global__ __launch_bounds__(256) void tnet_out_kernel(const float* __restrict__ zsel, const float* __restrict__ bn2, size_t total,
                                                       float slope, float* __restrict__ out) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(t & (TN_C2 - 1));
        out[t] = lrelu(fmaf(zsel[t], bn2[c], bn2[TN_C2 + c]), slope);
    }
}

// backward pre-pass over the points: partial sums of dz and dz*zhat_sel (BN2), dz = dt * act'(t)
__global__ __launch_bounds__(256) void tnet_bwd_reduce_kernel(const float* __restrict__ dT, const float* __restrict__ T,
                                                              const float* __restrict__ zsel, const float* __restrict__ bn2,
                                                              int P, float slope, double* __restrict__ part) {
    __shared__ double sh[2][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * 512, r1 = min(P, r0 + 512);
    const float mu = bn2[2 * TN_C2 + c], is = bn2[3 * TN_C2 + c];
    double s = 0.0, q = 0.0;
    for (int r = r0 + w; r < r1; r += 4) {
        size_t t = (size_t)r * TN_C2 + c;
        float d = dT[t];
        if (!(T[t] > 0.f)) d *= slope;
        s += d; q += (double)d * ((zsel[t] - mu) * is);
    }
    sh[0][w][lane] = s; sh[1][w][lane] = q;
    __syncthreads();
    if (w == 0) {
        part[((size_t)blockIdx.y * 2 + 0) * TN_C2 + c] = sh[0][0][lane] + sh[0][1][lane] + sh[0][2][lane] + sh[0][3][lane];
        part[((size_t)blockIdx.y * 2 + 1) * TN_C2 + c] = sh[1][0][lane] + sh[1][1][lane] + sh[1][2][lane] + sh[1][3][lane];
    }
}

// g = scale2 * dz  [P][128];  coef[0..127] = A2 = scale2*mean_dz, coef[128..255] = B2 = scale2*invstd2*mean_dzy
__global__ __launch_bounds__(256) void tnet_bwd_g_kernel(const float* __restrict__ dT, const float* __restrict__ T,
                                                         const float* __restrict__ bn2, const float* __restrict__ mean_dz,
                                                         const float* __restrict__ mean_dzy, size_t total, float slope,
                                                         float* __restrict__ g, float* __restrict__ coef) {
    size_t t0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t0 < TN_C2) {
        float sc = bn2[t0];
        coef[t0] = mean_dz ? sc * mean_dz[t0] : 0.f;
        coef[TN_C2 + t0] = mean_dz ? sc * bn2[3 * TN_C2 + t0] * mean_dzy[t0] : 0.f;
    }
    for (size_t t = t0; t < total; t += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(t & (TN_C2 - 1));
        float d = dT[t];
        if (!(T[t] > 0.f)) d *= slope;
        g[t] = bn2[c] * d;
    }
}

struct TnetBwdArgs {
    const float* uv; const int* idx; const float* bn1; const float* W2; const float* bn2;
    const float* g; const uint8_t* argsel; const float* coef;    // coef: A2[128], B2[128]
    float* dhp;            // [E][64]  dh' = dH * act'(a)
    float* dW2part;        // [gridDim.x][128*64]
    double* part1;         // [gridDim.x][2][64]   sums of dh' and dh'*hhat (BN1 backward)
    int P, N, k, TP, ntiles; float slope;
};

__global__ __launch_bounds__(512) void tnet_edge_bwd_kernel(TnetBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Hs = sm;
    float* Ws = Hs + TN_C1 * TN_SH;
    float* Zs = Ws + TN_C2 * TN_SW;
    float* S1 = Zs + TN_ROWS * TN_SZ;               // scale1, shift1, mean1, invstd1   [4][64]
    float* C2 = S1 + 4 * TN_C1;                     // A2, B2, mean2, (unused)          [4][128]
    int* rowpt = (int*)(C2 + 4 * TN_C2);            // [128]  packed (pt << 8 | s), -1 = invalid
    float* gs = (float*)(rowpt + TN_ROWS);          // [TP<=8][128] g tile
    uint8_t* as = (uint8_t*)(gs + TN_MAXTP * TN_C2);   // [TP<=8][128] argsel tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int rt = wave & 3, ct = wave >> 2;        // dH: row tile rt, channel tile ct;  dW2: o tile rt, channel tile ct
    for (int e = tid; e < TN_C2 * TN_C1; e += 512) Ws[(e >> 6) * TN_SW + (e & 63)] = p.W2[e];
    if (tid < 4 * TN_C1) S1[tid] = p.bn1[tid];
    if (tid < 2 * TN_C2) C2[tid] = p.coef[tid];
    if (tid < TN_C2) C2[2 * TN_C2 + tid] = p.bn2[2 * TN_C2 + tid];

    f32x16 accW;                                      // dW2 rows o = 32*rt + map(r,h), cols c = 32*ct + l31
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[r] = 0.f;
    double sd = 0.0, sdh = 0.0;

    int pt0 = 0, npts = 0, pt0n = 0, nptsn = 0, pt0nn = 0, nptsnn = 0;
    const int Bc = p.P / p.N;
    bool have = tn_tile(0, Bc, p.N, p.TP, pt0, npts);
    TnRows rows;
    if (have) tn_load_rows(rows, p.uv, tn_row_index(p.idx, pt0, npts, p.k, p.N, tid), pt0, p.k, tid);
    bool haven = have && tn_tile(1, Bc, p.N, p.TP, pt0n, nptsn);
    int jn = haven ? tn_row_index(p.idx, pt0n, nptsn, p.k, p.N, tid) : -1;
    for (int m = 0; have; ++m) {
        __syncthreads();
        const int nvalid = npts * p.k;
        tn_store_h(Hs, rows, tid);
        // tile m+1's rows and tile m+2's neighbour index travel under the three MFMA stages of tile m
        if (haven) tn_load_rows(rows, p.uv, jn, pt0n, p.k, tid);
        const bool havenn = haven && tn_tile(m + 2, Bc, p.N, p.TP, pt0nn, nptsnn);
        const int jnn = havenn ? tn_row_index(p.idx, pt0nn, nptsnn, p.k, p.N, tid) : -1;
        if (tid < TN_ROWS) {
            int pt = tid / p.k, s = tid - pt * p.k;
            rowpt[tid] = tid < nvalid ? ((pt << 8) | s) : -1;
        }
        for (int e = tid; e < npts * TN_C2; e += 512) {
            size_t gi = (size_t)pt0 * TN_C2 + e;
            gs[e] = p.g[gi];
            as[e] = p.argsel[gi];
        }
        __syncthreads();
        tn_compute_z(Hs, Ws, S1, Zs, nvalid, p.slope, wm, wn, l31, h);
        __syncthreads();
        // dZ in place:  g*[s == argsel] - A2 - B2*(Z - mean2)   on valid rows, 0 elsewhere
        {
            const int o = tid & 127;
            const float A2 = C2[o], B2 = C2[TN_C2 + o], m2 = C2[2 * TN_C2 + o];
#pragma unroll 8
            for (int row = tid >> 7; row < TN_ROWS; row += 4) {
                const int rp = rowpt[row];
                float dz = 0.f;
                if (rp >= 0) {
                    const int pt = rp >> 8, s = rp & 255;
                    const float z = Zs[row * TN_SZ + o];
                    dz = ((int)as[pt * TN_C2 + o] == s ? gs[pt * TN_C2 + o] : 0.f) - A2 - B2 * (z - m2);
                }
                Zs[row * TN_SZ + o] = dz;
            }
        }
        __syncthreads();
        // dH[row][c] = sum_o dZ[row][o] W2[o][c]        wave: rows 32*rt.., channels 32*ct..
        f32x16 accH;
#pragma unroll
        for (int r = 0; r < 16; ++r) accH[r] = 0.f;
#pragma unroll 8
        for (int t = 0; t < TN_C2 / 2; ++t) {
            const int o = 2 * t + h;
            float a = Zs[(32 * rt + l31) * TN_SZ + o];
            float b0 = Ws[o * TN_SW + 32 * ct + l31];
            accH = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, accH, 0, 0, 0);
        }
        // dW2[o][c] += sum_row dZ[row][o] H[row][c]      wave: o tile rt, channel tile ct
        {
            const int c = 32 * ct + l31;
            const float sc0 = S1[c], sh0 = S1[TN_C1 + c];
#pragma unroll 8
            for (int t = 0; t < TN_ROWS / 2; ++t) {
                const int row = 2 * t + h;
                float a = Zs[row * TN_SZ + 32 * rt + l31];
                float b0 = row < nvalid ? lrelu(fmaf(Hs[c * TN_SH + row], sc0, sh0), p.slope) : 0.f;
                accW = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, accW, 0, 0, 0);
            }
        }
        // dh' = dH * act'(a), write valid rows; BN1-backward sums
        {
            const int c = 32 * ct + l31;
            const float sc = S1[c], sh = S1[TN_C1 + c], mu = S1[2 * TN_C1 + c], is = S1[3 * TN_C1 + c];
            float lsd = 0.f, lsdh = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < nvalid) {
                    const float hp = Hs[c * TN_SH + row];
                    const float a = fmaf(hp, sc, sh);
                    const float d = accH[r] * (a > 0.f ? 1.f : p.slope);
                    p.dhp[((size_t)pt0 * p.k + row) * TN_C1 + c] = d;
                    lsd += d; lsdh = fmaf(d, (hp - mu) * is, lsdh);
                }
            }
            sd += lsd; sdh += lsdh;
        }
        have = haven; pt0 = pt0n; npts = nptsn;
        haven = havenn; pt0n = pt0nn; nptsn = nptsnn; jn = jnn;
    }
    // per-block partials
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * h;
        p.dW2part[(size_t)blockIdx.x * TN_C2 * TN_C1 + o * TN_C1 + 32 * ct + l31] = accW[r];
    }
    __syncthreads();
    double* red = (double*)Zs;                       // [8 waves][2 kinds][32 lanes]
    {
        double a = sd + __shfl_xor(sd, 32, 64);
        double b = sdh + __shfl_xor(sdh, 32, 64);
        if (h == 0) { red[(wave * 2 + 0) * 32 + l31] = a; red[(wave * 2 + 1) * 32 + l31] = b; }
    }
    __syncthreads();
    if (tid < 2 * TN_C1) {
        const int kind = tid >> 6, c = tid & 63;
        const int cti = c >> 5, cl = c & 31;
        double s = 0.0;
        for (int r4 = 0; r4 < 4; ++r4) s += red[((cti * 4 + r4) * 2 + kind) * 32 + cl];
        p.part1[((size_t)blockIdx.x * 2 + kind) * TN_C1 + c] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Backward, round 2: the per-edge tile dZ is never formed.  With BN2's backward in closed form the tile is AFFINE in Z = H' W2^T,
//     dZ[row][o] = gsel[row][o] - B2[o] Z[row][o] - K2[o],     gsel[row][o] = g[pt][o] if row is the arg-max slot of (pt, o) else 0,
// so both products that consume it split into a SPARSE part (one non-zero per point and channel) and a part through 64 x 64 matrices:
//     dH  = dZ W2    = gsel W2  -  H' M  -  cv          M = W2^T diag(B2) W2 (64 x 64),  cv = K2^T W2       (tnet_bwd_prep_kernel)
//     dW2 = dZ^T H'  = gsel^T H'  -  diag(B2) W2 (H'^T H')  -  K2 (1^T H')                                   (tnet_bwd_finish_kernel)
// Per 128-row tile that is 2 x 32 MFMAs per wave (P = H' M, and the tile's share of the Gram matrix H'^T H') instead of three
// 128 x 128 x 64 products (1/3 of the round-1 MFMA work, 1/6 of its flops), plus 2 x 128 fused multiply-adds per point and channel lane:
//     G1[o][c] += g[pt][o] H'[row(pt, arg)][c]        wave = 16 output channels (registers), lane = c
//     S[row(pt, arg)][c] += g[pt][o] W2[o][c]         wave = point, lane = c, the point's rows in registers indexed by the uniform slot
// g / arg-max slots are wave-uniform and come through the scalar cache.  LDS: H' double-buffered (the next tile is gathered while
// this one is computed), S, M, W2 = 147 KB, one workgroup of 8 waves per CU.  The epilogue (dh' = dH act'(a), BN1-backward sums)
// runs in the MFMA layout on P still in registers; a is recovered from the activated value (LeakyReLU is a bijection for slope != 0).
#define TG_HP 66
#define TG_SLAB (TN_C2 * TN_C1 + TN_C1 * TN_C1 + TN_C1)          // per-workgroup partials: G1 [128][64], Gram [64][64], colsum(H') [64]
struct TnetBwdGLds {
    float Hs[2][TN_ROWS * TG_HP];
    float Ss[TN_ROWS * TN_C1];
    float Ms[TN_C1 * TN_C1];
    float Ws[TN_C2 * TN_C1];
    float S1[4 * TN_C1];                 // scale1, shift1, mean1, invstd1
    float cv[TN_C1];
    float gT[2][TN_MAXTP * TN_C2];       // g of the tile's points, staged one tile ahead
    uint32_t aT[2][TN_MAXTP * TN_C2 / 4];   // their arg-max slots, 4 per word
};

__global__ __launch_bounds__(256) void tnet_bwd_prep_kernel(const float* __restrict__ W2, const float* __restrict__ coef,
                                                            const float* __restrict__ bn2, float* __restrict__ Mc) {
    __shared__ __attribute__((aligned(16))) float Ws[TN_C2 * TN_C1];
    __shared__ float B2s[TN_C2], K2s[TN_C2];
    const int tid = threadIdx.x;
    for (int e = tid; e < TN_C2 * TN_C1 / 4; e += 256) ((f32x4*)Ws)[e] = ((const f32x4*)W2)[e];
    if (tid < TN_C2) { B2s[tid] = coef[TN_C2 + tid]; K2s[tid] = coef[tid] - coef[TN_C2 + tid] * bn2[2 * TN_C2 + tid]; }
    __syncthreads();
    const int c1 = blockIdx.x * 4 + (tid >> 6), c2 = tid & 63;     // 17 workgroups: rows 0..63 of M, row 64 = cv
    float acc = 0.f;
    if (c1 < TN_C1) {
        for (int o = 0; o < TN_C2; ++o) acc = fmaf(Ws[o * TN_C1 + c1] * B2s[o], Ws[o * TN_C1 + c2], acc);
        Mc[c1 * TN_C1 + c2] = acc;
    } else if (c1 == TN_C1) {
        for (int o = 0; o < TN_C2; ++o) acc = fmaf(K2s[o], Ws[o * TN_C1 + c2], acc);
        Mc[TN_C1 * TN_C1 + c2] = acc;
    }
}

// R = the workgroup partials summed in a fixed order: dW2 = G1 - diag(B2) W2 Gram - K2 colsum
__global__ void tnet_bwd_finish_kernel(const float* __restrict__ R, const float* __restrict__ W2, const float* __restrict__ coef,
                                       const float* __restrict__ bn2, float* __restrict__ dW2) {
    const int o = blockIdx.x, c = threadIdx.x;
    const float B2 = coef[TN_C2 + o], K2 = coef[o] - B2 * bn2[2 * TN_C2 + o];
    float zh = 0.f;
    for (int c1 = 0; c1 < TN_C1; ++c1) zh = fmaf(W2[o * TN_C1 + c1], R[TN_C2 * TN_C1 + c1 * TN_C1 + c], zh);
    dW2[o * TN_C1 + c] = (R[o * TN_C1 + c] - B2 * zh) - K2 * R[TN_C2 * TN_C1 + TN_C1 * TN_C1 + c];
}

// Register-indexed FMAs (VGPR index mode, M0 = a wave-uniform index): two SALU + one VALU instruction per sparse entry.  The indexed
// register block is pinned because the instruction text has to name its first register: v[224:255] (32 rows), or v[216:255] as an
// 8- and a 32-register operand (40 rows); the two uses (S rows / H' rows) never overlap.
//   tg_fma_dst8: acc[slot_i - s0] += g_i * w_i        (destination and addend indexed)
//   tg_fma_src8: c_i += g_i * rows[slot_i - s0]       (second factor indexed)
// for the 8 entries whose slots are the bytes of (a0, a1); g_i are float bit patterns in SGPRs.
// MODE 0: k <= 32, slot masked to [0, 32).  MODE 1: k <= 40 (configs[4]), 40-row block, slot clamped to [0, 40).  MODE 2 (k > 40): 32-slot ranges,
// entries whose slot lies outside [s0, s0 + 32) add an exact zero.
template <int MODE> __device__ __forceinline__ void tg_fma_dst8(f32x8& lo, f32x32& hi, uint32_t a0, uint32_t a1, const uint32_t (&g)[8], int s0,
                                                              const float* w) {
    uint32_t t, u;
    if (MODE == 2) {
        asm volatile(
            "s_and_b32 %[t], %[a0], 0xff\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g0], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_on %[t], gpr_idx(SRC2,DST)\n\ts_nop 0\n\tv_fma_f32 v224, %[u], %[w0], v224\n\t"
            "s_bfe_u32 %[t], %[a0], 0x80008\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g1], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[u], %[w1], v224\n\t"
            "s_bfe_u32 %[t], %[a0], 0x80010\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g2], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[u], %[w2], v224\n\t"
            "s_bfe_u32 %[t], %[a0], 0x80018\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g3], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[u], %[w3], v224\n\t"
            "s_and_b32 %[t], %[a1], 0xff\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g4], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[u], %[w4], v224\n\t"
            "s_bfe_u32 %[t], %[a1], 0x80008\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g5], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[u], %[w5], v224\n\t"
            "s_bfe_u32 %[t], %[a1], 0x80010\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g6], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[u], %[w6], v224\n\t"
            "s_bfe_u32 %[t], %[a1], 0x80018\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g7], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[u], %[w7], v224\n\t"
            "s_set_gpr_idx_off"
            : "+{v[224:255]}"(hi), [t] "=&s"(t), [u] "=&s"(u)
            : [a0] "s"(a0), [a1] "s"(a1), [s0] "s"(s0), [g0] "s"(g[0]), [g1] "s"(g[1]), [g2] "s"(g[2]), [g3] "s"(g[3]), [g4] "s"(g[4]), [g5] "s"(g[5]), [g6] "s"(g[6]), [g7] "s"(g[7]),
              [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]), [w6] "v"(w[6]), [w7] "v"(w[7])
            : "scc");
    } else if (MODE == 1) {
        asm volatile(
            "s_and_b32 %[t], %[a0], 63\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_on %[t], gpr_idx(SRC2,DST)\n\ts_nop 0\n\tv_fma_f32 v216, %[g0], %[w0], v216\n\t"
            "s_bfe_u32 %[t], %[a0], 0x60008\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v216, %[g1], %[w1], v216\n\t"
            "s_bfe_u32 %[t], %[a0], 0x60010\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v216, %[g2], %[w2], v216\n\t"
            "s_bfe_u32 %[t], %[a0], 0x60018\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v216, %[g3], %[w3], v216\n\t"
            "s_and_b32 %[t], %[a1], 63\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v216, %[g4], %[w4], v216\n\t"
            "s_bfe_u32 %[t], %[a1], 0x60008\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v216, %[g5], %[w5], v216\n\t"
            "s_bfe_u32 %[t], %[a1], 0x60010\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v216, %[g6], %[w6], v216\n\t"
            "s_bfe_u32 %[t], %[a1], 0x60018\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v216, %[g7], %[w7], v216\n\t"
            "s_set_gpr_idx_off"
            : "+{v[216:223]}"(lo), "+{v[224:255]}"(hi), [t] "=&s"(t)
            : [a0] "s"(a0), [a1] "s"(a1), [g0] "s"(g[0]), [g1] "s"(g[1]), [g2] "s"(g[2]), [g3] "s"(g[3]), [g4] "s"(g[4]), [g5] "s"(g[5]), [g6] "s"(g[6]), [g7] "s"(g[7]),
              [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]), [w6] "v"(w[6]), [w7] "v"(w[7])
            : "scc");
    } else {
        asm volatile(
            "s_and_b32 %[t], %[a0], 31\n\ts_set_gpr_idx_on %[t], gpr_idx(SRC2,DST)\n\ts_nop 0\n\tv_fma_f32 v224, %[g0], %[w0], v224\n\t"
            "s_bfe_u32 %[t], %[a0], 0x50008\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[g1], %[w1], v224\n\t"
            "s_bfe_u32 %[t], %[a0], 0x50010\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[g2], %[w2], v224\n\t"
            "s_bfe_u32 %[t], %[a0], 0x50018\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[g3], %[w3], v224\n\t"
            "s_and_b32 %[t], %[a1], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[g4], %[w4], v224\n\t"
            "s_bfe_u32 %[t], %[a1], 0x50008\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[g5], %[w5], v224\n\t"
            "s_bfe_u32 %[t], %[a1], 0x50010\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[g6], %[w6], v224\n\t"
            "s_bfe_u32 %[t], %[a1], 0x50018\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 v224, %[g7], %[w7], v224\n\t"
            "s_set_gpr_idx_off"
            : "+{v[224:255]}"(hi), [t] "=&s"(t)
            : [a0] "s"(a0), [a1] "s"(a1), [g0] "s"(g[0]), [g1] "s"(g[1]), [g2] "s"(g[2]), [g3] "s"(g[3]), [g4] "s"(g[4]), [g5] "s"(g[5]), [g6] "s"(g[6]), [g7] "s"(g[7]),
              [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]), [w6] "v"(w[6]), [w7] "v"(w[7])
            : "scc");
    }
}
template <int MODE> __device__ __forceinline__ void tg_fma_src8(float* c, const f32x8& lo, const f32x32& hi, uint32_t a0, uint32_t a1,
                                                              const uint32_t (&g)[8], int s0) {
    uint32_t t, u;
    if (MODE == 2) {
        asm volatile(
            "s_and_b32 %[t], %[a0], 0xff\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g0], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_on %[t], gpr_idx(SRC1)\n\ts_nop 0\n\tv_fma_f32 %[c0], %[u], v224, %[c0]\n\t"
            "s_bfe_u32 %[t], %[a0], 0x80008\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g1], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c1], %[u], v224, %[c1]\n\t"
            "s_bfe_u32 %[t], %[a0], 0x80010\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g2], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c2], %[u], v224, %[c2]\n\t"
            "s_bfe_u32 %[t], %[a0], 0x80018\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g3], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c3], %[u], v224, %[c3]\n\t"
            "s_and_b32 %[t], %[a1], 0xff\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g4], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c4], %[u], v224, %[c4]\n\t"
            "s_bfe_u32 %[t], %[a1], 0x80008\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g5], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c5], %[u], v224, %[c5]\n\t"
            "s_bfe_u32 %[t], %[a1], 0x80010\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g6], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c6], %[u], v224, %[c6]\n\t"
            "s_bfe_u32 %[t], %[a1], 0x80018\n\ts_sub_u32 %[t], %[t], %[s0]\n\ts_cmp_lt_u32 %[t], 32\n\ts_cselect_b32 %[u], %[g7], 0\n\ts_and_b32 %[t], %[t], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c7], %[u], v224, %[c7]\n\t"
            "s_set_gpr_idx_off"
            : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), [t] "=&s"(t), [u] "=&s"(u)
            : "{v[224:255]}"(hi), [a0] "s"(a0), [a1] "s"(a1), [s0] "s"(s0), [g0] "s"(g[0]), [g1] "s"(g[1]), [g2] "s"(g[2]), [g3] "s"(g[3]), [g4] "s"(g[4]), [g5] "s"(g[5]), [g6] "s"(g[6]), [g7] "s"(g[7])
            : "scc");
    } else if (MODE == 1) {
        asm volatile(
            "s_and_b32 %[t], %[a0], 63\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_on %[t], gpr_idx(SRC1)\n\ts_nop 0\n\tv_fma_f32 %[c0], %[g0], v216, %[c0]\n\t"
            "s_bfe_u32 %[t], %[a0], 0x60008\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c1], %[g1], v216, %[c1]\n\t"
            "s_bfe_u32 %[t], %[a0], 0x60010\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c2], %[g2], v216, %[c2]\n\t"
            "s_bfe_u32 %[t], %[a0], 0x60018\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c3], %[g3], v216, %[c3]\n\t"
            "s_and_b32 %[t], %[a1], 63\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c4], %[g4], v216, %[c4]\n\t"
            "s_bfe_u32 %[t], %[a1], 0x60008\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c5], %[g5], v216, %[c5]\n\t"
            "s_bfe_u32 %[t], %[a1], 0x60010\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c6], %[g6], v216, %[c6]\n\t"
            "s_bfe_u32 %[t], %[a1], 0x60018\n\ts_min_u32 %[t], %[t], 39\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c7], %[g7], v216, %[c7]\n\t"
            "s_set_gpr_idx_off"
            : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), [t] "=&s"(t)
            : "{v[216:223]}"(lo), "{v[224:255]}"(hi), [a0] "s"(a0), [a1] "s"(a1), [g0] "s"(g[0]), [g1] "s"(g[1]), [g2] "s"(g[2]), [g3] "s"(g[3]), [g4] "s"(g[4]), [g5] "s"(g[5]), [g6] "s"(g[6]), [g7] "s"(g[7])
            : "scc");
    } else {
        asm volatile(
            "s_and_b32 %[t], %[a0], 31\n\ts_set_gpr_idx_on %[t], gpr_idx(SRC1)\n\ts_nop 0\n\tv_fma_f32 %[c0], %[g0], v224, %[c0]\n\t"
            "s_bfe_u32 %[t], %[a0], 0x50008\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c1], %[g1], v224, %[c1]\n\t"
            "s_bfe_u32 %[t], %[a0], 0x50010\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c2], %[g2], v224, %[c2]\n\t"
            "s_bfe_u32 %[t], %[a0], 0x50018\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c3], %[g3], v224, %[c3]\n\t"
            "s_and_b32 %[t], %[a1], 31\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c4], %[g4], v224, %[c4]\n\t"
            "s_bfe_u32 %[t], %[a1], 0x50008\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c5], %[g5], v224, %[c5]\n\t"
            "s_bfe_u32 %[t], %[a1], 0x50010\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c6], %[g6], v224, %[c6]\n\t"
            "s_bfe_u32 %[t], %[a1], 0x50018\n\ts_set_gpr_idx_idx %[t]\n\ts_nop 0\n\tv_fma_f32 %[c7], %[g7], v224, %[c7]\n\t"
            "s_set_gpr_idx_off"
            : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7]), [t] "=&s"(t)
            : "{v[224:255]}"(hi), [a0] "s"(a0), [a1] "s"(a1), [g0] "s"(g[0]), [g1] "s"(g[1]), [g2] "s"(g[2]), [g3] "s"(g[3]), [g4] "s"(g[4]), [g5] "s"(g[5]), [g6] "s"(g[6]), [g7] "s"(g[7])
            : "scc");
    }
}

template <int MODE, int KR> __global__ __launch_bounds__(512) void tnet_edge_bwdg_kernel(
    const float* __restrict__ uv, const int* __restrict__ idx, const float* __restrict__ bn1, const float* __restrict__ W2,
    const float* __restrict__ Mc, const float* __restrict__ g, const uint8_t* __restrict__ argsel, float* __restrict__ dhp,
    float* __restrict__ slabs, double* __restrict__ part1, int P, int N, int k, int TP, float slope) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    TnetBwdGLds& L = *reinterpret_cast<TnetBwdGLds*>(smraw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    for (int e = tid; e < (int)(sizeof(TnetBwdGLds) / 16); e += 512) ((f32x4*)smraw)[e] = f32x4{0.f, 0.f, 0.f, 0.f};   // no NaN patterns anywhere
    __syncthreads();
    for (int e = tid; e < TN_C2 * TN_C1 / 4; e += 512) ((f32x4*)L.Ws)[e] = ((const f32x4*)W2)[e];
    for (int e = tid; e < TN_C1 * TN_C1 / 4; e += 512) ((f32x4*)L.Ms)[e] = ((const f32x4*)Mc)[e];
    if (tid < 4 * TN_C1) L.S1[tid] = bn1[tid];
    if (tid < TN_C1) L.cv[tid] = Mc[TN_C1 * TN_C1 + tid];
    // gather role: 4 threads per tile row, 16 channels each
    const int grow = tid >> 2, gq = tid & 3;
    const int gpt = grow / k, gslot = grow - gpt * k;
    // MFMA roles: P tile (row block rb, channel tile ct); Gram tile (gi, gj) over the row half gh
    const int rb = wave & 3, ct = wave >> 2;
    const int gi = wave & 1, gj = (wave >> 1) & 1, gh = wave >> 2;
    f32x16 accG;
    float accO[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { accG[r] = 0.f; accO[r] = 0.f; }
    double sd = 0.0, sdh = 0.0, shs = 0.0;
    const int Bc = P / N;
    constexpr int NB = MODE == 1 ? 40 : 32;                    // rows of the indexed register block
    const int nsw = MODE == 2 ? (k + 31) >> 5 : 1;             // 32-slot ranges per point (MODE 2 only: k > 40)
    const float rslope = 1.0f / slope;

    auto row_index = [&](int pt0_, int npts_) -> int {        // uv row of this thread's neighbour, -1 = padding row
        return gpt < npts_ ? (pt0_ / N) * N + idx[(size_t)(pt0_ + gpt) * k + gslot] : -1;
    };
    auto load_rows = [&](int j, int pt0_, f32x4 (&u)[4], f32x4 (&v)[4]) {
        if (j >= 0) {
            const f32x4* up = (const f32x4*)(uv + (size_t)j * 2 * TN_C1 + 16 * gq);
            const f32x4* vp = (const f32x4*)(uv + (size_t)(pt0_ + gpt) * 2 * TN_C1 + TN_C1 + 16 * gq);
#pragma unroll
            for (int e = 0; e < 4; ++e) { u[e] = up[e]; v[e] = vp[e]; }
        }
    };
    // g / arg-max slots of a tile's points: loaded with the rows, stored to the LDS tables with them
    auto load_scal = [&](int pt0_, int npts_, float (&gq)[2], uint32_t& aq) {
#pragma unroll
        for (int e = 0; e < 2; ++e) gq[e] = tid + 512 * e < npts_ * TN_C2 ? g[(size_t)pt0_ * TN_C2 + tid + 512 * e] : 0.f;
        aq = tid < npts_ * (TN_C2 / 4) ? ((const uint32_t*)argsel)[(size_t)pt0_ * (TN_C2 / 4) + tid] : 0u;
    };
    auto store_scal = [&](int buf, const float (&gq)[2], uint32_t aq) {
        L.gT[buf][tid] = gq[0]; L.gT[buf][tid + 512] = gq[1];
        if (tid < TN_MAXTP * TN_C2 / 4) L.aT[buf][tid] = aq;
    };
    auto store_rows = [&](float* Hd, int j, const f32x4 (&u)[4], const f32x4 (&v)[4]) {
        float* dst = Hd + grow * TG_HP + 16 * gq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float hv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ch = 16 * gq + 4 * e + c;
                hv[c] = j >= 0 ? lrelu(fmaf(u[e][c] + v[e][c], L.S1[ch], L.S1[TN_C1 + ch]), slope) : 0.f;
            }
            *(float2*)(dst + 4 * e) = make_float2(hv[0], hv[1]);
            *(float2*)(dst + 4 * e + 2) = make_float2(hv[2], hv[3]);
        }
    };

    int pt0 = 0, npts = 0, pt0n = 0, nptsn = 0;
    bool have = tn_tile(0, Bc, N, TP, pt0, npts);
    bool haven = have && tn_tile(1, Bc, N, TP, pt0n, nptsn);
    __syncthreads();                                           // S1 is read by store_rows
    {
        f32x4 u[4], v[4];
        const int j0 = have ? row_index(pt0, npts) : -1;
        float gq[2]; uint32_t aq;
        load_rows(j0, pt0, u, v);
        load_scal(pt0, have ? npts : 0, gq, aq);
        store_rows(L.Hs[0], j0, u, v);
        store_scal(0, gq, aq);
    }
    int jn = haven ? row_index(pt0n, nptsn) : -1;
    __syncthreads();
    for (int m = 0; have; ++m) {
        const float* Hc = L.Hs[m & 1];
        const int nvalid = npts * k;
        f32x4 u[4], v[4];
        float gq[2]; uint32_t aq;
        load_rows(jn, pt0n, u, v);                             // next tile's rows and scalars in flight during this tile's work
        load_scal(pt0n, haven ? nptsn : 0, gq, aq);
        int pt0nn = 0, nptsnn = 0;
        const bool havenn = haven && tn_tile(m + 2, Bc, N, TP, pt0nn, nptsnn);
        const int jnn = havenn ? row_index(pt0nn, nptsnn) : -1;
        // ---- two independent halves per tile and wave: the 64 MFMAs (P = H' M for rows 32 rb.., channels 32 ct..; Gram tile (gi, gj)
        // += H'^T H' over rows 64 gh..) and the sparse work (S rows, G1).  Waves 0-3 run the MFMA half first, waves 4-7 the sparse
        // half first: each SIMD holds one wave of either kind, so its matrix pipe and its scalar/vector issue are busy at the same time.
        f32x16 accP;
#pragma unroll
        for (int r = 0; r < 16; ++r) accP[r] = 0.f;
        const float* gt = L.gT[m & 1];
        const uint32_t* at = L.aT[m & 1];
        for (int half = 0; half < 2; ++half) {
            if ((half == 0) == (wave < 4)) {
#ifndef TG_PROBE_NOMFMA
                // the two accumulation chains alternate: a wave alone keeps the matrix pipe full (a chain by itself waits for each result)
#pragma unroll 4
                for (int kk = 0; kk < 16; ++kk) {
                    const float2 a = *(const float2*)(Hc + (32 * rb + l31) * TG_HP + 4 * kk + 2 * h);
                    const float b0 = L.Ms[(4 * kk + 2 * h) * TN_C1 + 32 * ct + l31], b1 = L.Ms[(4 * kk + 2 * h + 1) * TN_C1 + 32 * ct + l31];
                    const float* hr = Hc + (64 * gh + 4 * kk + h) * TG_HP + l31;
                    const float ga0 = hr[32 * gi], gb0 = hr[32 * gj], ga1 = hr[2 * TG_HP + 32 * gi], gb1 = hr[2 * TG_HP + 32 * gj];
                    accP = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0, accP, 0, 0, 0);
                    accG = __builtin_amdgcn_mfma_f32_32x32x2f32(ga0, gb0, accG, 0, 0, 0);
                    accP = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1, accP, 0, 0, 0);
                    accG = __builtin_amdgcn_mfma_f32_32x32x2f32(ga1, gb1, accG, 0, 0, 0);
                }
#endif
            } else {
                // ---- S[row(pt, arg)][lane] = sum_o g[pt][o] W2[o][lane] over the channels whose arg-max is that row: one wave per
                // (point, 32-slot range) keeps its rows in the indexed register block, adds in ascending o, then stores the rows
#ifndef TG_PROBE_NOSCATTER
                if (wave < npts * nsw) {
                    const int spt = wave / nsw, s0 = (wave - spt * nsw) * 32;
                    const int gv0 = __float_as_int(gt[spt * TN_C2 + lane]), gv1 = __float_as_int(gt[spt * TN_C2 + 64 + lane]);
                    const uint32_t av = at[spt * (TN_C2 / 4) + l31];
                    const float* wb = L.Ws + lane;
                    f32x32 acc;
                    f32x8 acc2;                                       // rows 0..7 of the 40-row block (MODE 1); acc: the last 32
#pragma unroll
                    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc2[i] = 0.f;
                    float w[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) w[i] = wb[i * TN_C1];
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) {
                        const int chn = ch < 7 ? ch + 1 : ch;
                        float wn[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) wn[i] = wb[(16 * chn + i) * TN_C1];     // the next chunk's W2 values are requested first
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int o = 16 * ch + 8 * q;
                            const int gsrc = o < 64 ? gv0 : gv1;                          // g values travel as bit patterns in SGPRs
                            uint32_t ge[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) ge[e] = __builtin_amdgcn_readlane(gsrc, (o + e) & 63);
                            tg_fma_dst8<MODE>(acc2, acc, __builtin_amdgcn_readlane(av, o >> 2), __builtin_amdgcn_readlane(av, (o >> 2) + 1), ge,
                                              s0, w + 8 * q);
                        }
#pragma unroll
                        for (int i = 0; i < 16; ++i) w[i] = wn[i];
                    }
                    float* sb = L.Ss + (spt * k + s0) * TN_C1 + lane;
#pragma unroll
                    for (int i = 0; i < NB; ++i)
                        if (s0 + i < k) sb[i * TN_C1] = MODE == 1 ? (i < 8 ? acc2[i & 7] : acc[(i - 8) & 31]) : acc[i & 31];
                }
#endif
                // ---- G1[16 wave + i][lane] += g[pt][o] H'[row(pt, arg)][lane]: KR (>= k, or 32 when RANGED) of the point's H' rows are
                // loaded into the indexed register block, the 16 entries then cost one indexed FMA each.  Rows past k are the next
                // points' rows (or, past the tile, whatever follows in LDS, zeroed at kernel start): finite, and only ever multiplied
                // by an exact zero.
#ifndef TG_PROBE_NOG1
                for (int pt = 0; pt < npts; ++pt) {
                    const int gvi = __float_as_int(gt[pt * TN_C2 + 16 * wave + (lane & 15)]);
                    const uint32_t av = at[pt * (TN_C2 / 4) + 4 * wave + (lane & 3)];
                    uint32_t ge[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) ge[e] = __builtin_amdgcn_readlane(gvi, e);
                    uint32_t ae[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) ae[e] = __builtin_amdgcn_readlane(av, e);
                    for (int rg = 0; rg < nsw; ++rg) {
                        const int s0 = 32 * rg;
                        const float* hr = Hc + (pt * k + s0) * TG_HP + lane;
                        f32x32 rows;
                        f32x8 rows2;                                  // rows 0..7 of the 40-row block (MODE 1); rows: the last 32
#pragma unroll
                        for (int sl = 0; sl < 8; ++sl) rows2[sl] = MODE == 1 ? hr[sl * TG_HP] : 0.f;
#pragma unroll
                        for (int sl = 0; sl < 32; ++sl) {
                            const int r = MODE == 1 ? sl + 8 : sl;
                            rows[sl] = r < KR ? hr[r * TG_HP] : 0.f;
                        }
                        tg_fma_src8<MODE>(accO, rows2, rows, ae[0], ae[1], (const uint32_t(&)[8])ge[0], s0);
                        tg_fma_src8<MODE>(accO + 8, rows2, rows, ae[2], ae[3], (const uint32_t(&)[8])ge[8], s0);
                    }
                }
#endif
            }
        }
        if (haven) { store_rows(L.Hs[(m & 1) ^ 1], jn, u, v); store_scal((m & 1) ^ 1, gq, aq); }
        __syncthreads();
        // ---- dH = S - P - cv in the MFMA layout: channel c = 32 ct + l31, rows 32 rb + map(r, h)
        {
            const int c = 32 * ct + l31;
            const float sc = L.S1[c], sh = L.S1[TN_C1 + c], mu = L.S1[2 * TN_C1 + c], is = L.S1[3 * TN_C1 + c], cvc = L.cv[c];
            const float rsc = sc != 0.f ? 1.0f / sc : 0.f;
            float lsd = 0.f, lsdh = 0.f, lhs = 0.f;
            const int row0 = 32 * rb + 4 * h;                  // one base per array + compile-time offsets (LDS / global immediates)
            const float* sp = L.Ss + row0 * TN_C1 + c;
            const float* hp = Hc + row0 * TG_HP + c;
            float* dp = dhp + ((size_t)pt0 * k + row0) * TN_C1 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                const float sv = sp[dr * TN_C1];
                const float hv = hp[dr * TG_HP];
#ifdef TG_PROBE_NOEPI
                if (row0 + dr < nvalid && sv == 123.456f) {
#else
                if (row0 + dr < nvalid) {
#endif
                    const float dH = (sv - accP[r]) - cvc;
                    const float a = hv > 0.f ? hv : hv * rslope;
                    const float d = dH * (hv > 0.f ? 1.f : slope);
                    dp[dr * TN_C1] = d;
                    lsd += d; lsdh = fmaf(d, ((a - sh) * rsc - mu) * is, lsdh); lhs += hv;
                }
            }
            sd += lsd; sdh += lsdh; shs += lhs;
        }
        __syncthreads();
        pt0 = pt0n; npts = nptsn; have = haven;
        pt0n = pt0nn; nptsn = nptsnn; haven = havenn; jn = jnn;
    }
    // ---- per-workgroup partials
    float* slab = slabs + (size_t)blockIdx.x * TG_SLAB;
#pragma unroll
    for (int i = 0; i < 16; ++i) slab[(16 * wave + i) * TN_C1 + lane] = accO[i];
    float* gred = L.Ss;                                        // 4 Gram tiles of the upper row half [4][16][64]
    if (gh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gred[((wave & 3) * 16 + r) * 64 + lane] = accG[r];
    }
    double* red = reinterpret_cast<double*>(L.Hs[0]);          // [3 kinds][4 row blocks][64 channels]
    {
        const double a = sd + __shfl_xor(sd, 32, 64), b = sdh + __shfl_xor(sdh, 32, 64), c3 = shs + __shfl_xor(shs, 32, 64);
        if (h == 0) {
            const int c = 32 * ct + l31;
            red[(0 * 4 + rb) * 64 + c] = a; red[(1 * 4 + rb) * 64 + c] = b; red[(2 * 4 + rb) * 64 + c] = c3;
        }
    }
    __syncthreads();
    if (gh == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            slab[TN_C2 * TN_C1 + (32 * gi + (r & 3) + 8 * (r >> 2) + 4 * h) * TN_C1 + 32 * gj + l31] =
                accG[r] + gred[((wave & 3) * 16 + r) * 64 + lane];
    }
    if (tid < 3 * TN_C1) {
        const int kind = tid >> 6, c = tid & 63;
        const double t = (red[(kind * 4 + 0) * 64 + c] + red[(kind * 4 + 1) * 64 + c]) + (red[(kind * 4 + 2) * 64 + c] + red[(kind * 4 + 3) * 64 + c]);
        if (kind < 2) part1[((size_t)blockIdx.x * 2 + kind) * TN_C1 + c] = t;
        else slab[TN_C2 * TN_C1 + TN_C1 * TN_C1 + c] = (float)t;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Backward, round 4 (product modes 1 / 2, even k in [8, 64]): the same Gram form with the SPARSE half turned into dense products on the
// bf16 matrix cores.  gsel (one non-zero per point and channel) is never stored: its MFMA fragments are built in registers from the
// tile's g / arg-max tables -- a compare per element instead of a register-indexed FMA per non-zero and lane (~70 clocks each, which
// bounded tnet_edge_bwdg_kernel).  Every fp32 value is split into three bf16 pieces as in gemm_split_kernel (six piece products, fp32
// accumulation).  Per 128-row tile:
//     waves 0-3 (row block rb = wave): dH = [gsel | H'] [W2 ; -M]   (K = 128 + 64) for both 32-channel tiles: 144 MFMAs; the epilogue
//                (dh' = dH act'(a), BN1-backward sums) runs on the accumulators after the barrier, while the next tile is staged;
//     waves 4-7 (o tile mt = wave - 4):  G1 += gsel^T H' (K = 128 rows, both channel tiles) and one tile of the Gram matrix H'^T H'
//                from the same B fragments: 144 MFMAs.
// LDS images are [row][piece][64 channels] bf16 with a 448-byte row pitch (= 192 mod 256: the four k-rows of a transposed read fall in
// four different 64-byte bank windows); H' is read both ways (row-major A fragments for H' M, transposed for the products over rows), its
// 16-byte slots are XOR-swizzled inside each 64-byte block by (row >> 2) & 3 so that the ds_read_b128 pattern is conflict-free as well.
// H' (56 KB) + W2 (56 KB) + -M (28 KB) + tables = sizeof(TnetBwdSLds), about 158 KB of the CU's 160: one workgroup of 8 waves per CU.
typedef __bf16 tbf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int tu32x2 __attribute__((ext_vector_type(2)));
#define TB_PITCH 448
#define TB_LDS __attribute__((address_space(3)))
struct TnetBwdSLds {
    char Hp[TN_ROWS * TB_PITCH];                  // H' [row][piece][64 c], swizzled
    char Wp[TN_C2 * TB_PITCH];                    // W2 [o][piece][64 c]
    char Mp[TN_C1 * TB_PITCH];                    // -M [c1][piece][64 c]
    unsigned short gp[TN_MAXTP * 3 * TN_C2];      // g of the tile's points, three bf16 pieces [pt][piece][o]
    uint8_t ap[TN_MAXTP * TN_C2];                 // their arg-max slots [pt][o]
    uint2 gq[TN_MAXTP * TN_C2];                   // the same packed per (pt, o): {piece 0 | piece 1 << 16, piece 2 | slot << 16}
    float S1[4 * TN_C1];                          // scale1, shift1, mean1, invstd1
    float cv[TN_C1];
    float Vs[TN_MAXTP * TN_C1];                   // centre rows v_i of the tile being staged
};
typedef TB_LDS char* tb_lds_ptr;                  // 32-bit LDS addresses: one register each, 16-bit instruction offsets
__device__ __forceinline__ tbf16x8 tb_tr(tb_lds_ptr lo, tb_lds_ptr hi) {
    const tbf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((TB_LDS tbf16x4*)lo);
    const tbf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((TB_LDS tbf16x4*)hi);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
// acc += a b with the six piece products, smallest first
__device__ __forceinline__ void tb_mac6(f32x16& acc, const tbf16x8 (&a)[3], const tbf16x8 (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}
// two / three independent accumulation chains, interleaved: a chain by itself waits for each result
__device__ __forceinline__ void tb_mac6x2(f32x16& c0, f32x16& c1, const tbf16x8 (&a0)[3], const tbf16x8 (&b0)[3], const tbf16x8 (&a1)[3],
                                          const tbf16x8 (&b1)[3]) {
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[PA[i]], b0[PB[i]], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[PA[i]], b1[PB[i]], c1, 0, 0, 0);
    }
}
__device__ __forceinline__ void tb_mac6x3(f32x16& c0, f32x16& c1, f32x16& c2, const tbf16x8 (&a0)[3], const tbf16x8 (&b0)[3],
                                          const tbf16x8 (&a1)[3], const tbf16x8 (&b1)[3], const tbf16x8 (&a2)[3], const tbf16x8 (&b2)[3]) {
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[PA[i]], b0[PB[i]], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[PA[i]], b1[PB[i]], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[PA[i]], b2[PB[i]], c2, 0, 0, 0);
    }
}
__device__ __forceinline__ void tb_split1(float x, unsigned short& a, unsigned short& b, unsigned short& c) {
    const uint32_t pa = tn_cvt_pk(x, 0.f);
    const float r = x - __uint_as_float(pa << 16);
    const uint32_t pb = tn_cvt_pk(r, 0.f);
    const uint32_t pc = tn_cvt_pk(r - __uint_as_float(pb << 16), 0.f);
    a = (unsigned short)pa; b = (unsigned short)pb; c = (unsigned short)pc;
}

// the workgroup's tile walk without a division per tile (tn_tile's order: XCD-aware when the cloud count and the grid allow it); three
// positions (tiles m, m + 1, m + 2) share the parameters
struct TbWalkParams { int tpc, per, cstep, B; };
struct TbWalk {
    int cloud, tic;
    __device__ __forceinline__ void init(TbWalkParams& w, int B_, int N, int TP) {
        w.B = B_; w.tpc = (N + TP - 1) / TP;
        int L;
        if ((B_ & 7) == 0 && (gridDim.x & 7) == 0) { L = blockIdx.x >> 3; w.per = gridDim.x >> 3; w.cstep = 8; cloud = (blockIdx.x & 7) + 8 * (L / w.tpc); }
        else { L = blockIdx.x; w.per = gridDim.x; w.cstep = 1; cloud = L / w.tpc; }
        tic = L % w.tpc;
    }
    __device__ __forceinline__ bool ok(const TbWalkParams& w) const { return cloud < w.B; }
    __device__ __forceinline__ void next(const TbWalkParams& w) {
        tic += w.per;
        while (tic >= w.tpc) { tic -= w.tpc; cloud += w.cstep; }
    }
};

// ONEP (`precision` = MLSP_PREC_BF16): the contractions multiply the leading pieces only (one MFMA where the fp32-accurate form has six);
// H' for the epilogue is still recovered from all three (its sign and value enter the BatchNorm-backward sums).
template <bool ONEP>
__global__ __launch_bounds__(512) void tnet_edge_bwds_kernel(
    const float* __restrict__ uv, const int* __restrict__ idx, const float* __restrict__ bn1, const float* __restrict__ W2,
    const float* __restrict__ Mc, const float* __restrict__ g, const uint8_t* __restrict__ argsel, float* __restrict__ dhp,
    float* __restrict__ slabs, double* __restrict__ part1, int P, int N, int k, int TP, float slope) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    static_assert(sizeof(TnetBwdSLds) <= 160 * 1024, "tnet_edge_bwds_kernel: the LDS image must fit one CU");
    TnetBwdSLds& L = *reinterpret_cast<TnetBwdSLds*>(smraw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    for (int e = tid; e < (int)(sizeof(TnetBwdSLds) / 16); e += 512) ((f32x4*)smraw)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    {   // W2 pieces: thread -> row o, 16 channels
        const int o = tid >> 2, q4 = tid & 3;
        const f32x4* src = (const f32x4*)(W2 + (size_t)o * TN_C1 + 16 * q4);
        char* dst = L.Wp + o * TB_PITCH + 32 * q4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 x = src[e];
            uint32_t pa[2], pb[2], pc[2];
            tn_split2(x[0], x[1], pa[0], pb[0], pc[0]);
            tn_split2(x[2], x[3], pa[1], pb[1], pc[1]);
            *(uint2*)(dst + 8 * e) = make_uint2(pa[0], pa[1]);
            *(uint2*)(dst + 128 + 8 * e) = make_uint2(pb[0], pb[1]);
            *(uint2*)(dst + 256 + 8 * e) = make_uint2(pc[0], pc[1]);
        }
    }
    {   // -M pieces: thread -> row c1, 8 channels
        const int c1 = tid >> 3, q8 = tid & 7;
        const f32x4* src = (const f32x4*)(Mc + (size_t)c1 * TN_C1 + 8 * q8);
        char* dst = L.Mp + c1 * TB_PITCH + 16 * q8;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const f32x4 x = src[e];
            uint32_t pa[2], pb[2], pc[2];
            tn_split2(-x[0], -x[1], pa[0], pb[0], pc[0]);
            tn_split2(-x[2], -x[3], pa[1], pb[1], pc[1]);
            *(uint2*)(dst + 8 * e) = make_uint2(pa[0], pa[1]);
            *(uint2*)(dst + 128 + 8 * e) = make_uint2(pb[0], pb[1]);
            *(uint2*)(dst + 256 + 8 * e) = make_uint2(pc[0], pc[1]);
        }
    }
    if (tid < 4 * TN_C1) L.S1[tid] = bn1[tid];
    if (tid < TN_C1) L.cv[tid] = Mc[TN_C1 * TN_C1 + tid];
    // gather role: 4 threads per tile row, 16 channels each
    const int grow = tid >> 2, gq = tid & 3;
    const int gpt = grow / k, gslot = grow - gpt * k;
    const int kinv = 65536 / k + 1;                            // (r * kinv) >> 16 == r / k for r < 128, 8 <= k <= 64

    // Every global load of the walk is UNCONDITIONAL (clamped addresses, masks applied where the value is consumed): a load under a
    // branch whose other side writes the same register makes the compiler drain the whole memory queue right there (measured: 4,000 clocks
    // per tile at the top of the loop).
    TbWalkParams wp;
    TbWalk wc, wn, wnn;                                        // tiles m, m + 1, m + 2 of this workgroup
    struct RowIdx { int raw, base; bool valid; };
    auto row_index = [&](const TbWalk& w) -> RowIdx {         // uv row of this thread's neighbour in tile w (raw: still in flight)
        const int pt0_ = w.cloud * N + w.tic * TP, npts_ = min(TP, N - w.tic * TP);
        const bool valid = w.ok(wp) && gpt < npts_;
        RowIdx r;
        r.raw = idx[valid ? (size_t)(pt0_ + gpt) * k + gslot : (size_t)0]; r.base = w.cloud * N; r.valid = valid;
        return r;
    };
    auto row_of = [&](const RowIdx& r) -> int { return r.valid ? r.base + r.raw : -1; };     // -1 = padding row
    // the centre term v_i is the same for the k rows of a point: TP rows staged through LDS (Vs), one float per thread
    auto load_rows = [&](int j, int pt0_, f32x4 (&u)[4], float& vst) {
        const f32x4* up = (const f32x4*)(uv + (size_t)max(j, 0) * 2 * TN_C1 + 16 * gq);
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = up[e];
        vst = uv[(size_t)min(pt0_ + (tid >> 6), P - 1) * 2 * TN_C1 + TN_C1 + (tid & 63)];
    };
    // g / arg-max slots of a tile's points: g values tid and tid + 512 (point = value >> 7, channel = value & 127) with their own slot bytes
    // (the packed table of waves 4-7), and the same bytes in the byte table of waves 0-3; npts_ == 0: nothing is stored
    auto load_scal = [&](int pt0_, int npts_, float (&gv)[2], uint8_t (&ab)[2]) {
        const int last = max(npts_ * TN_C2 - 1, 0);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const size_t at = (size_t)pt0_ * TN_C2 + min(tid + 512 * e, last);
            gv[e] = g[at]; ab[e] = argsel[at];
        }
    };
    auto store_scal = [&](int npts_, const float (&gv)[2], const uint8_t (&ab)[2]) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = tid + 512 * e, pt = i >> 7, o = i & 127;
            unsigned short a, b, c;
            tb_split1(i < npts_ * TN_C2 ? gv[e] : 0.f, a, b, c);
            L.gp[(pt * 3 + 0) * TN_C2 + o] = a; L.gp[(pt * 3 + 1) * TN_C2 + o] = b; L.gp[(pt * 3 + 2) * TN_C2 + o] = c;
            L.gq[i] = make_uint2((uint32_t)a | ((uint32_t)b << 16), (uint32_t)c | ((uint32_t)ab[e] << 16));
            L.ap[i] = ab[e];                                   // (points past npts_ have g = 0: their slots do not matter)
        }
    };
    auto store_rows = [&](int j, const f32x4 (&u)[4]) {       // Vs holds the tile's centre rows
        char* dst = L.Hp + grow * TB_PITCH;
        const int x = ((grow >> 2) & 3) << 4;
        const float* vs = L.Vs + min(gpt, TN_MAXTP - 1) * TN_C1 + 16 * gq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 vv = *(const f32x4*)(vs + 4 * e);
            float hv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ch = 16 * gq + 4 * e + c;
                hv[c] = j >= 0 ? lrelu(fmaf(u[e][c] + vv[c], L.S1[ch], L.S1[TN_C1 + ch]), slope) : 0.f;
            }
            uint32_t pa[2], pb[2], pc[2];
            tn_split2(hv[0], hv[1], pa[0], pb[0], pc[0]);
            tn_split2(hv[2], hv[3], pa[1], pb[1], pc[1]);
            const int off = (32 * gq + 8 * e) ^ x;
            *(uint2*)(dst + off) = make_uint2(pa[0], pa[1]);
            *(uint2*)(dst + 128 + off) = make_uint2(pb[0], pb[1]);
            *(uint2*)(dst + 256 + off) = make_uint2(pc[0], pc[1]);
        }
    };

    // fragment addressing.  Transposed reads (ds_read_b64_tr_b16): lane 4q + p of a 16-lane group supplies k-row q, columns 4p .. 4p+3
    const int ti = lane & 15, tg16 = (lane >> 4) & 1;
    const int tcol = 32 * tg16 + 8 * (ti & 3);                                  // byte column inside a 64-byte (32-channel) block
    const int trow_lo = 8 * h + (ti >> 2), trow_hi = trow_lo + 4;
    const tb_lds_ptr lds = (tb_lds_ptr)smraw;
    constexpr int OFF_HP = 0, OFF_WP = TN_ROWS * TB_PITCH, OFF_MP = OFF_WP + TN_C2 * TB_PITCH, OFF_GP = OFF_MP + TN_C1 * TB_PITCH,
                  OFF_AP = OFF_GP + TN_MAXTP * 3 * TN_C2 * 2, OFF_GQ = OFF_AP + TN_MAXTP * TN_C2;
    static_assert(OFF_AP == offsetof(TnetBwdSLds, ap) && OFF_GP == offsetof(TnetBwdSLds, gp) && OFF_GQ == offsetof(TnetBwdSLds, gq), "LDS layout");
    const tb_lds_ptr wlo = lds + OFF_WP + trow_lo * TB_PITCH + tcol;             // + 16 s * TB_PITCH + 64 ct + 128 piece
    const tb_lds_ptr mlo = lds + OFF_MP + trow_lo * TB_PITCH + tcol;
    const tb_lds_ptr hlo = lds + OFF_HP + trow_lo * TB_PITCH + (tcol ^ (((2 * h) & 3) << 4));
    const tb_lds_ptr hhi = lds + OFF_HP + trow_hi * TB_PITCH + (tcol ^ (((2 * h + 1) & 3) << 4));
    const int mt = wave & 3, gi = mt >> 1, gj = mt & 1;
    // identity B fragment (waves 0-3): H' M's A fragments times the identity put H' itself -- exactly, the pieces sum to the fp32 value --
    // into the accumulator layout of dH, where the epilogue needs it.  Rebuilt at its four uses per tile (12 instructions; eight registers)
    auto ident = [&](int sp) -> tbf16x8 {
        uint32_t d[4];
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            const int k0_ = 16 * sp + 8 * h + 2 * i2;
            d[i2] = (k0_ == l31 ? 0x3f80u : 0u) | (k0_ + 1 == l31 ? 0x3f800000u : 0u);
        }
        return tn_pack8(d);
    };
    // accumulators: waves 0-3 use acc[0..1] for the dH tile and acch[0..1] for H' in the same layout (cleared per tile), waves 4-7 keep G1
    // (acc[0..1]) and the Gram tile (acch[0]) across the whole walk -- the same registers, the role of a wave never changes
    f32x16 acc[2], acch[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; acch[0][r] = 0.f; acch[1][r] = 0.f; }
    double sd[2] = {0.0, 0.0}, sdh[2] = {0.0, 0.0}, shs[2] = {0.0, 0.0};

#ifdef TB_STAMPS
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define TB_STAMP(i_) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsum[i_] += t_ - tprev; tprev = t_; } while (0)
#else
#define TB_STAMP(i_)
#endif
    bool pend = false; int ppt0 = 0, pnvalid = 0;              // waves 0-3: the epilogue of (channel tile 1, previous tile) is still due
    wc.init(wp, P / N, N, TP);
    wn = wc; wn.next(wp);
    wnn = wn; wnn.next(wp);
    __syncthreads();                                           // S1 is read by store_rows
    if (wc.ok(wp)) {
        f32x4 u[4]; float vst;
        const int j0 = row_of(row_index(wc));
        const int pt0_ = wc.cloud * N + wc.tic * TP, npts_ = min(TP, N - wc.tic * TP);
        float gv[2]; uint8_t ab[2];
        load_rows(j0, pt0_, u, vst);
        load_scal(pt0_, npts_, gv, ab);
        L.Vs[tid] = vst;
        __syncthreads();
        store_rows(j0, u);
        store_scal(npts_, gv, ab);
    }
    RowIdx jn = row_index(wn);
    __syncthreads();
#ifdef TB_STAMPS
    tprev = __builtin_amdgcn_s_memtime();
#endif
    while (wc.ok(wp)) {
        const int pt0 = wc.cloud * N + wc.tic * TP, npts = min(TP, N - wc.tic * TP);
        const int pt0n = wn.ok(wp) ? wn.cloud * N + wn.tic * TP : pt0, nptsn = wn.ok(wp) ? min(TP, N - wn.tic * TP) : 0;
        const int nvalid = npts * k;
        f32x4 u[4]; float vst;
        float gv[2]; uint8_t ab[2];
        const int jrow = row_of(jn);
        load_rows(jrow, pt0n, u, vst);                         // next tile's rows and scalars in flight during this tile's products
        load_scal(pt0n, nptsn, gv, ab);
        const RowIdx jnn = row_index(wnn);
        TB_STAMP(0);
        int opq;                                               // an opaque zero: keeps the role-specific addressing below out of the loop-invariant set
        asm volatile("s_mov_b32 %0, 0" : "=s"(opq));
        if (wave < 4) {
            // this lane's tile row as the M index of dH
            const int arow = 32 * wave + l31 + opq;
            int apt = (arow * kinv) >> 16;
            uint32_t slotrep = (uint32_t)(arow - apt * k) * 0x01010101u;
            if (apt >= TP) { apt = 0; slotrep = 0x7f7f7f7fu; }                   // rows past the tile's points: no slot matches
            const tb_lds_ptr gpb = lds + OFF_GP + (apt * 3 * TN_C2 + 8 * h) * 2; // + 32 s (+ 256 per piece)
            const tb_lds_ptr apb = lds + OFF_AP + apt * TN_C2 + 8 * h;           // + 16 s
            const int ax = ((arow >> 2) & 3) << 4;
            const tb_lds_ptr hrow = lds + OFF_HP + arow * TB_PITCH + ((16 * h) ^ (ax & 16));     // + (32 s) ^ (ax & 32) + 128 piece
            // Two passes over the 12 k16 steps (8 of gsel W2: A fragment = this row's 8 channels o, non-zero where the arg-max slot is this
            // row's slot; then 4 of H' (-M)), one per 32-channel tile ct.  The register epilogue of the OTHER tile rides between the MFMAs:
            // pass 0 carries the epilogue of (ct = 1, previous tile), pass 1 that of (ct = 0, this tile) -- a vector instruction issued
            // while another wave streams MFMAs waits about one MFMA slot, inside the issuing wave's own stream it is free (see
            // tnet_edge_fwd3_kernel).  The A fragments are rebuilt in the second pass (30 vector instructions per step, hidden the same way).
            auto ld_g = [&](int s_, u32x4 (&gp_)[3], tu32x2& aw_) {
                gp_[0] = *(TB_LDS const u32x4*)(gpb + 32 * s_); gp_[1] = *(TB_LDS const u32x4*)(gpb + 2 * TN_C2 + 32 * s_);
                gp_[2] = *(TB_LDS const u32x4*)(gpb + 4 * TN_C2 + 32 * s_);
                aw_ = *(TB_LDS const tu32x2*)(apb + 16 * s_);
            };
            auto ld_h = [&](int s_, tbf16x8 (&a_)[3]) {
                const tb_lds_ptr ar = hrow + ((32 * s_) ^ (ax & 32));
#pragma unroll
                for (int q = 0; q < 3; ++q) a_[q] = *(TB_LDS const tbf16x8*)(ar + 128 * q);
            };
            // A wave issues in order and stalls at an MFMA whose accumulator is still in the pipe: everything else of a step (the LDS requests
            // of step s + 1, the two epilogue entries, the masks and A fragment of step s + 1) is placed BETWEEN the six dependent MFMAs of
            // step s, one scheduling barrier per slot.
            auto pass = [&](auto ct_tag, bool epi_on, int ept0, int envalid) {
                constexpr int ct = decltype(ct_tag)::value, ec = 1 - ct;       // products of tile ct, epilogue of tile ec
                constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};      // piece products, smallest first
                auto ld_b = [&](tb_lds_ptr base, int s_, tbf16x8 (&b_)[3]) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const tb_lds_ptr w = base + 16 * s_ * TB_PITCH + 64 * ct + 128 * q;
                        b_[q] = tb_tr(w, w + 4 * TB_PITCH);
                    }
                };
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[ct][r] = 0.f; acch[ct][r] = 0.f; }
                const int c = 32 * ec + l31;
                const float cvc = L.cv[c];
                float lsd = 0.f, lsh = 0.f, lhs = 0.f;
                const int lim = epi_on ? envalid - (32 * wave + 4 * h) : 0;      // valid rows of this lane: map(r) < lim
                float* dp = dhp + ((size_t)ept0 * k + 32 * wave + 4 * h) * TN_C1 + c;
                auto epi = [&](int r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    const float hv = acch[ec][r];
                    const float dH = acc[ec][r] - cvc;
                    const float dd = dH * (hv > 0.f ? 1.f : slope);
                    if (dr < lim) { dp[dr * TN_C1] = dd; lsd += dd; }
                    lsh = fmaf(dH, hv, lsh); lhs += hv;       // rows past the tile's points have h' = 0
                };
                uint32_t mk[4];
                auto masks = [&](const tu32x2& aw_) {
                    const uint32_t t0 = aw_[0] ^ slotrep, t1 = aw_[1] ^ slotrep;         // bytes < 0x80: zero where the slot matches
                    const uint32_t f0 = ((((t0 + 0x7f7f7f7fu) >> 7) & 0x01010101u) ^ 0x01010101u) * 0xffu;
                    const uint32_t f1 = ((((t1 + 0x7f7f7f7fu) >> 7) & 0x01010101u) ^ 0x01010101u) * 0xffu;
                    mk[0] = __builtin_amdgcn_perm(0u, f0, 0x01010000u); mk[1] = __builtin_amdgcn_perm(0u, f0, 0x03030202u);
                    mk[2] = __builtin_amdgcn_perm(0u, f1, 0x01010000u); mk[3] = __builtin_amdgcn_perm(0u, f1, 0x03030202u);
                };
                auto frag = [&](const u32x4& gp_) -> tbf16x8 {
                    const uint32_t d[4] = {gp_[0] & mk[0], gp_[1] & mk[1], gp_[2] & mk[2], gp_[3] & mk[3]};
                    return tn_pack8(d);
                };
                tbf16x8 a[3], bc[3];
                {
                    u32x4 gp0[3]; tu32x2 aw0;
                    ld_g(0, gp0, aw0); ld_b(wlo, 0, bc);
                    masks(aw0);
#pragma unroll
                    for (int q = 0; q < 3; ++q) a[q] = frag(gp0[q]);
                }
#pragma unroll
                for (int s = 0; s < 12; ++s) {
                    u32x4 gpn[3]; tu32x2 awn; tbf16x8 bn_[3], an[3];
#pragma unroll
                    for (int i2 = 0; i2 < 6; ++i2) {
                        if (!ONEP || i2 == 5) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[i2]], bc[PB[i2]], acc[ct], 0, 0, 0);
                        if (i2 == 0) {                         // requests of step s + 1
                            if (s + 1 < 8) { ld_g(s + 1, gpn, awn); ld_b(wlo, s + 1, bn_); }
                            else if (s + 1 < 12) { ld_h(s + 1 - 8, an); ld_b(mlo, s + 1 - 8, bn_); }
                        }
                        if (s < 8 && i2 == 1) epi(2 * s);
                        if (s < 8 && i2 == 2) epi(2 * s + 1);
                        if (s + 1 < 8) {                       // A fragment of step s + 1
                            if (i2 == 3) masks(awn);
                            if (i2 == 4) { an[0] = frag(gpn[0]); an[1] = frag(gpn[1]); }
                            if (i2 == 5) an[2] = frag(gpn[2]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (s >= 8 && ((s - 8) >> 1) == ct) {      // H' itself for the epilogue: channels 16 (s - 8) .. + 15 belong to tile (s - 8) >> 1
                        const tbf16x8 idf = ident((s - 8) & 1);
#pragma unroll
                        for (int q = 0; q < 3; ++q) acch[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], idf, acch[ct], 0, 0, 0);
                    }
                    if (s + 1 < 12) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) { bc[q] = bn_[q]; a[q] = an[q]; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (epi_on) {
                    // BN1-backward sum of dh' hhat with hhat = ((a - shift) / scale - mean) invstd = alpha a - beta, taken as alpha sum(dH h') -
                    // beta sum(dh'): dh' a = dH h' on both sides of the activation (slope * (h' / slope) = h' to the last bit or one)
                    const float sc = L.S1[c], sh = L.S1[TN_C1 + c], mu = L.S1[2 * TN_C1 + c], is = L.S1[3 * TN_C1 + c];
                    const float rsc = sc != 0.f ? 1.0f / sc : 0.f;
                    sd[ec] += lsd; sdh[ec] += (rsc * is) * lsh - ((sh * rsc + mu) * is) * lsd; shs[ec] += lhs;
                }
            };
            pass(std::integral_constant<int, 0>{}, pend, ppt0, pnvalid);
            TB_STAMP(1);
            pass(std::integral_constant<int, 1>{}, true, pt0, nvalid);
            pend = true; ppt0 = pt0; pnvalid = nvalid;
            TB_STAMP(3);
        } else {
            // ---- G1 += gsel^T H' and the Gram tile (gi, gj): K = the tile's rows; A fragment = channel `go_` of the 8 rows of the k16 step
            const int go_ = 32 * mt + l31 + opq;               // this lane's output channel o as the M index of G1
            const tb_lds_ptr gqb = lds + OFF_GQ + go_ * 8;     // + pt * 1024
            tbf16x8 bhc[2][3];
            tu32x2 ec[2];
            auto ld_bh = [&](int s_, tbf16x8 (&b_)[2][3]) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        b_[nt][q] = tb_tr(hlo + 16 * s_ * TB_PITCH + 64 * nt + 128 * q, hhi + 16 * s_ * TB_PITCH + 64 * nt + 128 * q);
            };
            auto ld_e = [&](int s_, tu32x2 (&e_)[2]) {          // packed (pieces, slot) of channel go_ for the two points the 8-row window covers
                const int pA = ((16 * s_ + 8 * h) * kinv) >> 16;
#pragma unroll
                for (int w = 0; w < 2; ++w) e_[w] = *(TB_LDS const tu32x2*)(gqb + min(pA + w, TN_MAXTP - 1) * (TN_C2 * 8));
            };
            // the 18 MFMAs of a step run as three independent chains (G1 tile 0, G1 tile 1, Gram); the operands of step s + 1 are requested
            // before them.  (Hand-placing the fragment construction between the MFMAs as in waves 0-3 was measured slower: 15.4 vs 10.4
            // thousand clocks per tile -- three chains already keep the wave issuing.)
            ld_bh(0, bhc); ld_e(0, ec);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                tbf16x8 bhn[2][3];
                tu32x2 en[2];
                if (s + 1 < 8) { ld_bh(s + 1, bhn); ld_e(s + 1, en); }
                const int r0 = 16 * s + 8 * h;
                const int pA = (r0 * kinv) >> 16;
                uint32_t d[3][4];
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) d[q][i2] = 0u;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int pX = pA + w;
                    const int e = pX * k + (int)(ec[w][1] >> 16) - r0;
                    const int di = ((unsigned)e < 8u && pX < TP) ? (e >> 1) : 4;
                    const int sh = (e & 1) << 4;
                    const uint32_t t[3] = {(ec[w][0] & 0xffffu) << sh, (ec[w][0] >> 16) << sh, (ec[w][1] & 0xffffu) << sh};
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) d[q][i2] = di == i2 ? t[q] : d[q][i2];
                }
                tbf16x8 a[3];
                a[0] = tn_pack8(d[0]); a[1] = tn_pack8(d[1]); a[2] = tn_pack8(d[2]);
                if constexpr (ONEP) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bhc[0][0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bhc[1][0], acc[1], 0, 0, 0);
                    acch[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((gi ? bhc[1] : bhc[0])[0], (gj ? bhc[1] : bhc[0])[0], acch[0], 0, 0, 0);
                } else tb_mac6x3(acc[0], acc[1], acch[0], a, bhc[0], a, bhc[1], gi ? bhc[1] : bhc[0], gj ? bhc[1] : bhc[0]);
                if (s + 1 < 8) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int q = 0; q < 3; ++q) bhc[nt][q] = bhn[nt][q];
                    ec[0] = en[0]; ec[1] = en[1];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            TB_STAMP(1);
        }
        L.Vs[tid] = vst;                                       // (Vs is only read between the two barriers)
        __syncthreads();                                       // every wave has read its fragments: the images may be overwritten
        TB_STAMP(4);
        if (wn.ok(wp)) { store_rows(jrow, u); store_scal(nptsn, gv, ab); }
        TB_STAMP(5);
        __syncthreads();
        TB_STAMP(6);
        wc = wn; wn = wnn; wnn.next(wp); jn = jnn;
    }
    if (wave < 4 && pend) {                                    // the last tile's second channel tile
        const int c = 32 + l31;
        const float sc = L.S1[c], sh = L.S1[TN_C1 + c], mu = L.S1[2 * TN_C1 + c], is = L.S1[3 * TN_C1 + c], cvc = L.cv[c];
        const float rsc = sc != 0.f ? 1.0f / sc : 0.f;
        float lsd = 0.f, lsh = 0.f, lhs = 0.f;
        const int lim = pnvalid - (32 * wave + 4 * h);
        float* dp = dhp + ((size_t)ppt0 * k + 32 * wave + 4 * h) * TN_C1 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const float hv = acch[1][r];
            const float dH = acc[1][r] - cvc;
            const float dd = dH * (hv > 0.f ? 1.f : slope);
            if (dr < lim) { dp[dr * TN_C1] = dd; lsd += dd; }
            lsh = fmaf(dH, hv, lsh); lhs += hv;
        }
        sd[1] += lsd; sdh[1] += (rsc * is) * lsh - ((sh * rsc + mu) * is) * lsd; shs[1] += lhs;
    }
    // ---- per-workgroup partials: G1 [128][64], Gram [64][64], colsum(H') [64]; BN1-backward sums
    float* slab = slabs + (size_t)blockIdx.x * TG_SLAB;
    double* red = reinterpret_cast<double*>(L.Hp);            // [3 kinds][4 row blocks][64 channels]
    if (wave >= 4) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h) * TN_C1 + 32 * nt + l31] = acc[nt][r];
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[TN_C2 * TN_C1 + (32 * gi + (r & 3) + 8 * (r >> 2) + 4 * h) * TN_C1 + 32 * gj + l31] = acch[0][r];
    } else {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const double a = sd[ct] + __shfl_xor(sd[ct], 32, 64), b = sdh[ct] + __shfl_xor(sdh[ct], 32, 64), c3 = shs[ct] + __shfl_xor(shs[ct], 32, 64);
            if (h == 0) {
                const int c = 32 * ct + l31;
                red[(0 * 4 + wave) * 64 + c] = a; red[(1 * 4 + wave) * 64 + c] = b; red[(2 * 4 + wave) * 64 + c] = c3;
            }
        }
    }
    __syncthreads();
    if (tid < 3 * TN_C1) {
        const int kind = tid >> 6, c = tid & 63;
        const double t = (red[(kind * 4 + 0) * 64 + c] + red[(kind * 4 + 1) * 64 + c]) + (red[(kind * 4 + 2) * 64 + c] + red[(kind * 4 + 3) * 64 + c]);
        if (kind < 2) part1[((size_t)blockIdx.x * 2 + kind) * TN_C1 + c] = t;
        else slab[TN_C2 * TN_C1 + TN_C1 * TN_C1 + c] = (float)t;
    }
#ifdef TB_STAMPS
    __syncthreads();
    if (lane == 0 && (wave == 0 || wave == 4)) {              // diagnostic build only: overwrites this workgroup's G1 partial
        unsigned long long* o = (unsigned long long*)slab + (wave ? 16 : 0);
        o[0] = 0x5354414d50533031ull + (wave ? 1 : 0);
        for (int q = 0; q < 8; ++q) o[1 + q] = tsum[q];
    }
#endif
}
#undef TB_STAMP

// Fold dh' onto the points (BN1 backward in closed form), wave per point, lane = channel (64):
//   g_e = scale1*(dh'_e - m1 - hhat_e*m2);  dv_i = sum_s g_(i,s);  du_j = sum_{e in rev(j)} g_e
__global__ __launch_bounds__(256) void tnet_edge_bwd2_kernel(const float* __restrict__ dhp, const float* __restrict__ uv,
                                                             const float* __restrict__ s1, const float* __restrict__ bn1,
                                                             const float* __restrict__ m1v, const float* __restrict__ m2v,
                                                             const int* __restrict__ rev_off, const int* __restrict__ rev_ent,
                                                             int P, int N, int k, float* __restrict__ duv) {
    const int c = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int j = blockIdx.x * 4 + w;
    if (N % 4 == 0) {
        int cloud, chunk;
        xcd_cloud_map(blockIdx.x, N / 4, P / N, cloud, chunk);
        j = cloud * N + chunk * 4 + w;
    }
    if (j >= P) return;
    const int base = (j / N) * N;
    const float sc = bn1[c], mu = bn1[2 * TN_C1 + c], is = bn1[3 * TN_C1 + c];
    const float m1 = m1v ? m1v[c] : 0.f, m2 = m1v ? m2v[c] : 0.f;
    const float u = uv[(size_t)j * 2 * TN_C1 + c], v = uv[(size_t)j * 2 * TN_C1 + TN_C1 + c];
    float S = 0.f;
    for (int s = 0; s < k; ++s) S += dhp[((size_t)j * k + s) * TN_C1 + c];
    const float fk = (float)k;
    float dv = sc * (S - fk * m1 - m2 * is * (s1[(size_t)j * TN_C1 + c] + fk * (v - mu)));
    const int e0 = rev_off[j], e1 = rev_off[j + 1];
    float R = 0.f, Vs = 0.f;
    for (int e = e0; e < e1; ++e) {
        const int ent = rev_ent[e];
        const int i = base + (ent >> 8);
        R += dhp[((size_t)i * k + (ent & 255)) * TN_C1 + c];
        Vs += uv[(size_t)i * 2 * TN_C1 + TN_C1 + c];
    }
    const float deg = (float)(e1 - e0);
    float du = sc * (R - deg * m1 - m2 * is * (deg * (u - mu) + Vs));
    duv[(size_t)j * 2 * TN_C1 + c] = du;
    duv[(size_t)j * 2 * TN_C1 + TN_C1 + c] = dv;
}

// ---------------------------------------------------------------------------------------------
int gemm_precision_mode();        // gemm.hip
int tnet_grid(int ntiles) { return ntiles < 512 ? ((ntiles + 7) / 8) * 8 : 512; }
int tnet_points_per_tile(int k) { return k > 0 && k <= TN_ROWS ? (TN_ROWS / k > 8 ? 8 : TN_ROWS / k) : 0; }

static int tnet_set_lds(const void* fn) {
    size_t lds = (size_t)TN_LDS_FLOATS * sizeof(float);
    hipError_t e = mlsp_lds_limit(fn, lds);
    return e == hipSuccess ? MLSP_OK : (int)e;
}

// number of BN2 partial rows launch_tnet_edge_fwd writes (= its grid size)
int tnet_fwd_parts(int B, int N, int k) {
    if (k == 20 || k == 40) {
        const int pt = TF_ROWS / k, ntiles = B * ((N + pt - 1) / pt);
        return ntiles < 512 ? ((ntiles + 7) / 8) * 8 : 512;          // two 4-wave workgroups per CU (237 VGPRs), a multiple of 8 (XCD-aware walk)
    }
    const int tp = tnet_points_per_tile(k);
    return tp > 0 ? tnet_grid(B * ((N + tp - 1) / tp)) : 0;
}

int launch_tnet_edge_fwd(hipStream_t st, const float* uv, const int* idx, const float* bn1, const float* W2, const float* gamma2,
                         int P, int N, int k, float slope, float* zsel, uint8_t* argsel, double* part) {
    TnetFwdArgs a;
    a.uv = uv; a.idx = idx; a.bn1 = bn1; a.W2 = W2; a.gamma2 = gamma2; a.zsel = zsel; a.argsel = argsel; a.part = part;
    a.P = P; a.N = N; a.k = k; a.TP = tnet_points_per_tile(k); a.slope = slope;
    if (a.TP <= 0) return MLSP_ERR_UNSUPPORTED;
    if (k == 20 || k == 40) {
        const int grid = tnet_fwd_parts(P / N, N, k);
        a.TP = TF_ROWS / k;
        a.ntiles = (P / N) * ((N + a.TP - 1) / a.TP);
        // tnet_edge_fwd3_kernel (split products on the bf16 cores; closer to float64 than the f32-MFMA kernel: 1.7e-7 vs 2.0e-7 rel-L2,
        // tools/tnet_acc.py) whenever the call's `precision` asks for products on the bf16 cores (modes 1 and 2), tnet_edge_fwd2_kernel
        // (f32 MFMA) in mode 0 -- the product mode decides, at every size: B = 32, N = 1024 forward op 222 -> 173 us, B = 8 81 -> 70 us; small
        // launches are enqueue-bound and the two kernels take the same time (tools/time_tnet.py: 88 vs 84 us at B = 4, N = 128).
        // Read-once A/B switches: MLSP_TNET_FWD_SPLIT=1 / MLSP_TNET_FWD_F32=1.
        static const bool split_env = getenv("MLSP_TNET_FWD_SPLIT") != nullptr, f32_env = getenv("MLSP_TNET_FWD_F32") != nullptr;
        const bool split_products = !f32_env && (split_env || gemm_precision_mode() != 0);
        if (!split_products) {
            if (k == 20) hipLaunchKernelGGL((tnet_edge_fwd2_kernel<20>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((tnet_edge_fwd2_kernel<40>), dim3(grid), dim3(256), 0, st, a);
        } else {
            static const bool six_env = getenv("MLSP_TNET_BF16_SIX") != nullptr;      // A/B: mode 1 on the six-product kernels (rounds 3-5)
            const bool onep = gemm_precision_mode() == 1 && !split_env && !six_env;
            if (k == 20 && !onep) hipLaunchKernelGGL((tnet_edge_fwd3_kernel<20>), dim3(grid), dim3(256), 0, st, a);
            else if (k == 20) hipLaunchKernelGGL((tnet_edge_fwd3_kernel<20, true>), dim3(grid), dim3(256), 0, st, a);
            else if (!onep) hipLaunchKernelGGL((tnet_edge_fwd3_kernel<40>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((tnet_edge_fwd3_kernel<40, true>), dim3(grid), dim3(256), 0, st, a);
        }
        return mlsp_launch_status();
    }
    a.ntiles = (P / N) * ((N + a.TP - 1) / a.TP);
    int rc = tnet_set_lds((const void*)tnet_edge_fwd_kernel);
    if (rc) return rc;
    hipLaunchKernelGGL(tnet_edge_fwd_kernel, dim3(tnet_grid(a.ntiles)), dim3(512), (size_t)TN_LDS_FLOATS * sizeof(float), st, a);
    return mlsp_launch_status();
}

int launch_tnet_out(hipStream_t st, const float* zsel, const float* bn2, int P, float slope, float* out) {
    size_t total = (size_t)P * TN_C2;
    size_t b = (total + 255) / 256;
    hipLaunchKernelGGL(tnet_out_kernel, dim3((unsigned)(b < 4096 ? b : 4096)), dim3(256), 0, st, zsel, bn2, total, slope, out);
    return mlsp_launch_status();
}

int launch_tnet_bwd_reduce(hipStream_t st, const float* dT, const float* T, const float* zsel, const float* bn2, int P,
                           float slope, double* part) {
    hipLaunchKernelGGL(tnet_bwd_reduce_kernel, dim3(TN_C2 / 64, (P + 511) / 512), dim3(256), 0, st, dT, T, zsel, bn2, P, slope, part);
    return mlsp_launch_status();
}

int launch_tnet_bwd_g(hipStream_t st, const float* dT, const float* T, const float* bn2, const float* mean_dz,
                      const float* mean_dzy, int P, float slope, float* g, float* coef) {
    size_t total = (size_t)P * TN_C2;
    size_t b = (total + 255) / 256;
    hipLaunchKernelGGL(tnet_bwd_g_kernel, dim3((unsigned)(b < 4096 ? b : 4096)), dim3(256), 0, st, dT, T, bn2, mean_dz, mean_dzy,
                       total, slope, g, coef);
    return mlsp_launch_status();
}

int launch_slab_reduce(hipStream_t st, const float* slab, float* C, int M, int N, int ldc, int nsplit);
int tnet_bwd_grid(int ntiles) { return ntiles < 256 ? ((ntiles + 7) / 8) * 8 : 256; }
// floats of scratch the backward needs: workgroup partial slabs + M/cv + the reduced slab (Gram path), or the round-1 kernel's dW2 slabs
size_t tnet_bwd_scratch_floats(int ntiles) {
    const size_t a = (size_t)tnet_bwd_grid(ntiles) * TG_SLAB + TN_C1 * TN_C1 + TN_C1 + TG_SLAB;
    const size_t b = (size_t)tnet_grid(ntiles) * TN_C2 * TN_C1;
    return a > b ? a : b;
}
// Per-edge backward: dh' [E][64], dW2, and the BN1-backward partial sums part1 [*nparts][2][64].
int launch_tnet_edge_bwd(hipStream_t st, const float* uv, const int* idx, const float* bn1, const float* W2, const float* bn2,
                         const float* g, const uint8_t* argsel, const float* coef, int P, int N, int k, float slope, float* dhp,
                         float* scratch, double* part1, float* dW2, int* nparts) {
    const int TP = tnet_points_per_tile(k);
    if (TP <= 0) return MLSP_ERR_UNSUPPORTED;
    const int ntiles = (P / N) * ((N + TP - 1) / TP);
    static const bool use_old = getenv("MLSP_TNET_BWD_OLD") != nullptr;       // A/B switch (tools/time_tnet.py): the round-1 kernel
    if (!use_old && slope > 0.f) {                             // the pre-activation is recovered from the activated value: needs a bijection
        const int nb = tnet_bwd_grid(ntiles);
        float* slabs = scratch;
        float* Mc = slabs + (size_t)nb * TG_SLAB;
        float* R = Mc + TN_C1 * TN_C1 + TN_C1;
        hipLaunchKernelGGL(tnet_bwd_prep_kernel, dim3(TN_C1 / 4 + 1), dim3(256), 0, st, W2, coef, bn2, Mc);
        // products on the bf16 cores (modes 1 and 2): the dense split form (tnet_edge_bwds_kernel); mode 0 keeps exact fp32 products with
        // the register-indexed sparse half.  Read-once A/B switch: MLSP_TNET_BWD_F32=1.
        static const bool f32_env = getenv("MLSP_TNET_BWD_F32") != nullptr;
        if (!f32_env && gemm_precision_mode() != 0 && k % 2 == 0 && k >= 8 && k <= 64) {
            const size_t lds = sizeof(TnetBwdSLds);
            static const bool six_env = getenv("MLSP_TNET_BF16_SIX") != nullptr;      // A/B: mode 1 on the six-product kernel (rounds 4-5)
            auto kern = (gemm_precision_mode() == 1 && !six_env) ? tnet_edge_bwds_kernel<true> : tnet_edge_bwds_kernel<false>;
            hipError_t e = mlsp_lds_limit((const void*)kern, lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, st, uv, idx, bn1, W2, Mc, g, argsel, dhp, slabs, part1, P, N, k, TP, slope);
        } else {
            const size_t lds = sizeof(TnetBwdGLds);
            auto kern = k <= 20 ? tnet_edge_bwdg_kernel<0, 20> : k <= 24 ? tnet_edge_bwdg_kernel<0, 24> : k <= 32 ? tnet_edge_bwdg_kernel<0, 32>
                      : k <= 40 ? tnet_edge_bwdg_kernel<1, 40> : tnet_edge_bwdg_kernel<2, 32>;
            hipError_t e = mlsp_lds_limit((const void*)kern, lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, st, uv, idx, bn1, W2, Mc, g, argsel, dhp, slabs, part1, P, N, k, TP, slope);
        }
        int rc = mlsp_launch_status();
        if (rc) return rc;
        rc = launch_slab_reduce(st, slabs, R, TG_SLAB / TN_C1, TN_C1, TN_C1, nb);
        if (rc) return rc;
        hipLaunchKernelGGL(tnet_bwd_finish_kernel, dim3(TN_C2), dim3(TN_C1), 0, st, R, W2, coef, bn2, dW2);
        *nparts = nb;
        return mlsp_launch_status();
    }
    TnetBwdArgs a;
    a.uv = uv; a.idx = idx; a.bn1 = bn1; a.W2 = W2; a.bn2 = bn2; a.g = g; a.argsel = argsel; a.coef = coef;
    a.dhp = dhp; a.dW2part = scratch; a.part1 = part1;
    a.P = P; a.N = N; a.k = k; a.TP = TP; a.slope = slope;
    a.ntiles = ntiles;
    int rc = tnet_set_lds((const void*)tnet_edge_bwd_kernel);
    if (rc) return rc;
    const int nb = tnet_grid(ntiles);
    hipLaunchKernelGGL(tnet_edge_bwd_kernel, dim3(nb), dim3(512), (size_t)TN_LDS_FLOATS * sizeof(float), st, a);
    rc = mlsp_launch_status();
    if (rc) return rc;
    *nparts = nb;
    return launch_slab_reduce(st, scratch, dW2, TN_C2, TN_C1, TN_C1, nb);
}

int launch_tnet_edge_bwd2(hipStream_t st, const float* dhp, const float* uv, const float* s1, const float* bn1, const float* m1,
                          const float* m2, const int* rev_off, const int* rev_ent, int P, int N, int k, float* duv) {
    hipLaunchKernelGGL(tnet_edge_bwd2_kernel, dim3((P + 3) / 4), dim3(256), 0, st, dhp, uv, s1, bn1, m1, m2, rev_off, rev_ent, P, N,
                       k, duv);
    return mlsp_launch_status();
}

// ---- first T-Net conv, weight gradient WITHOUT folding dh' onto the points (the input cloud needs no gradient) ---------------------
// With hpre_e = Wa x_j + Wv x_i (Wa = W1[:, :C], Wv = W1[:, C:] - Wa; edge e = (i, s), j its neighbour) and the closed-form BN1
// backward g_e = scale (dh'_e - m1 - hhat_e m2), the gradients dA = sum_e g_e x_j^T and dD = sum_e g_e x_i^T need from the E-sized data
// only  T1 = sum_e dh'_e x_j^T  and  T2 = sum_e dh'_e x_i^T  (64 x C each): ONE sequential pass over dh' instead of the reverse-index
// gather of tnet_edge_bwd2_kernel (random 256-byte rows) + the [P,128]^T [P,C] GEMM.  Everything else is a function of the
// coordinate moments  A = sum_e x_j x_j^T = sum_j deg_j x_j x_j^T,  Bm = sum_e x_i x_j^T = sum_i x_i (sum_s x_j(i,s))^T,
// Cm = k sum_i x_i x_i^T,  sj = sum_j deg_j x_j,  si = k sum_i x_i  (a P-sized pass):
//   dA[c] = sc_c ( T1[c] - m1_c sj - m2_c is_c ( Wa[c] A + Wv[c] Bm - mu_c sj ) )
//   dD[c] = sc_c ( T2[c] - m1_c si - m2_c is_c ( Wa[c] Bm^T + Wv[c] Cm - mu_c si ) )        dW1 = [dA - dD | dD]
// (eval mode: m1 = m2 = 0).  C <= 4 coordinates.
#define TW_ROWS 512          // edge rows per workgroup of the T pass (two chunks of 256)
#define TW_XM 56             // moments per block: A[16] Bm[16] Cm[16] sj[4] si[4]
// T pass: a workgroup owns TW_ROWS consecutive edge rows.  Per chunk of 256 rows every thread first fetches ONE row's coordinates
// (idx -> x_j, and x_i: the only dependent loads, 256 of them in parallel) into LDS; then the four waves stream the dh' rows
// (lane = channel, 256-byte coalesced loads, eight rows in flight) against the staged coordinates.
__global__ __launch_bounds__(256) void tnet_bwd_tmom_kernel(const float* __restrict__ dhp, const int* __restrict__ idx, const float* __restrict__ x,
                                                            int ldx, long E, int N, int k, int C, double* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float xs[256][8];            // [row of the chunk][x_j (4) | x_i (4)]
    __shared__ float red[4][TN_C1][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long e0 = (long)blockIdx.x * TW_ROWS, e1 = e0 + TW_ROWS < E ? e0 + TW_ROWS : E;
    float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
    for (long r0 = e0; r0 < e1; r0 += 256) {
        const long e = r0 + tid;
        f32x4 vj = {0.f, 0.f, 0.f, 0.f}, vi = {0.f, 0.f, 0.f, 0.f};
        if (e < e1) {
            const long i = e / k;
            const long j = (i / N) * N + idx[e];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < C) { vj[c] = x[j * ldx + c]; vi[c] = x[i * ldx + c]; }
        }
        __syncthreads();                                                 // the previous chunk's readers are done
        *(f32x4*)&xs[tid][0] = vj;
        *(f32x4*)&xs[tid][4] = vi;
        __syncthreads();
        const int nrow = (int)(e1 - r0 < 256 ? e1 - r0 : 256);
        for (int rb = wave; rb < nrow; rb += 32) {                       // rows rb, rb + 4, ... rb + 28 of this wave in flight
            float d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + 4 * u;
                d[u] = r < nrow ? dhp[(size_t)(r0 + r) * TN_C1 + lane] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + 4 * u;
                const int rr = r < nrow ? r : 0;
                const f32x4 cj = *(const f32x4*)&xs[rr][0], ci = *(const f32x4*)&xs[rr][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { t1[c] = fmaf(d[u], cj[c], t1[c]); t2[c] = fmaf(d[u], ci[c], t2[c]); }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { red[wave][lane][c] = t1[c]; red[wave][lane][4 + c] = t2[c]; }
    __syncthreads();
    for (int t = tid; t < TN_C1 * 8; t += 256) {
        const int ch = t >> 3, q = t & 7;
        part[((size_t)ch * gridDim.x + blockIdx.x) * 8 + q] = ((double)red[0][ch][q] + (double)red[1][ch][q]) + ((double)red[2][ch][q] + (double)red[3][ch][q]);
    }
}

// coordinate moments: one wave per 64 points (many small workgroups: the pass is a chain of two dependent gathers, not bandwidth)
__global__ __launch_bounds__(64) void tnet_bwd_xmom_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ idx,
                                                           const int* __restrict__ rev_off, int P, int N, int k, int C, double* __restrict__ part) {
    const int lane = threadIdx.x;
    const int i = blockIdx.x * 64 + lane;
    float m[TW_XM];
#pragma unroll
    for (int q = 0; q < TW_XM; ++q) m[q] = 0.f;
    if (i < P) {
        float xi[4] = {0.f, 0.f, 0.f, 0.f}, xb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) xi[c] = x[(size_t)i * ldx + c];
        const int base = (i / N) * N;
        float aj[4][4];                                                   // sum over this point's neighbours of x_j x_j^T
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) aj[a][b] = 0.f;
        for (int s0 = 0; s0 < k; s0 += 16) {                              // sixteen neighbours' coordinates in flight
            // unconditional loads (clamped slot / coordinate, masked afterwards): a load under a branch whose other side writes the
            // register makes the compiler wait for it on the spot -- sixteen round trips in a row instead of one
            int j[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) j[u] = idx[(size_t)i * k + min(s0 + u, k - 1)];
            float v[16][4];
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) v[u][c] = x[(size_t)(base + j[u]) * ldx + min(c, C - 1)];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool on = s0 + u < k;
                const float vv[4] = {on ? v[u][0] : 0.f, (on && C > 1) ? v[u][1] : 0.f, (on && C > 2) ? v[u][2] : 0.f, (on && C > 3) ? v[u][3] : 0.f};
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    xb[a] += vv[a];
#pragma unroll
                    for (int b = 0; b < 4; ++b) aj[a][b] = fmaf(vv[a], vv[b], aj[a][b]);
                }
            }
        }
        const float fk = (float)k;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                m[4 * a + b] = aj[a][b];                      // A  = sum_e x_j x_j^T (per edge: no in-degrees, no reverse index)
                m[16 + 4 * a + b] = xi[a] * xb[b];            // Bm[a][b] = sum_e x_i[a] x_j[b]
                m[32 + 4 * a + b] = fk * xi[a] * xi[b];       // Cm
            }
            m[48 + a] = xb[a];                                // sj = sum_e x_j
            m[52 + a] = fk * xi[a];                           // si
        }
    }
#pragma unroll
    for (int q = 0; q < TW_XM; ++q) {
        float v = m[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) part[(size_t)blockIdx.x * TW_XM + q] = (double)v;
    }
}

// one workgroup per conv1 output channel c: sums the partials (parallel over the partial blocks, fixed order), evaluates the closed
// form, writes row c of dW1 [C1][2C]
__global__ __launch_bounds__(256) void tnet_bwd_w1_finish_kernel(const double* __restrict__ tpart, int ntb, const double* __restrict__ xpart, int nxb,
                                                                 const float* __restrict__ W1, const float* __restrict__ bn1,
                                                                 const float* __restrict__ m1v, const float* __restrict__ m2v, int C,
                                                                 float* __restrict__ dW1) {
    __shared__ double sh[4][8];
    __shared__ double shx[4][TW_XM];
    __shared__ double xm[TW_XM];
    __shared__ double T[8];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // T1 | T2 of channel c: thread (group g = tid >> 3, value q = tid & 7) sums the partial blocks g, g + 32, ...
        const int q = tid & 7, g = tid >> 3;
        double a = 0.0;
        const double* tp = tpart + (size_t)c * ntb * 8 + q;               // [c][block][8]: this channel's partials are contiguous
        for (int b0 = g; b0 < ntb; b0 += 32 * 8) {                        // eight loads in flight, added in block order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = b0 + 32 * u < ntb ? tp[(size_t)(b0 + 32 * u) * 8] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
        if (lane < 8) sh[wave][lane] = a;
    }
    {   // coordinate moments: thread (group g = tid / 56 < 4, value q = tid % 56)
        const int q = tid % TW_XM, g = tid / TW_XM;
        double a = 0.0;
        if (g < 4)
            for (int b0 = g; b0 < nxb; b0 += 4 * 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = b0 + 4 * u < nxb ? xpart[(size_t)(b0 + 4 * u) * TW_XM + q] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) a += v[u];
            }
        if (g < 4) shx[g][q] = a;
    }
    __syncthreads();
    if (tid < 8) T[tid] = (sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid]);
    if (tid < TW_XM) xm[tid] = (shx[0][tid] + shx[1][tid]) + (shx[2][tid] + shx[3][tid]);
    __syncthreads();
    if (tid < C) {
        const int d = tid;
        const double sc = bn1[c], mu = bn1[2 * TN_C1 + c], is = bn1[3 * TN_C1 + c];
        const double m1 = m1v ? (double)m1v[c] : 0.0, m2 = m1v ? (double)m2v[c] : 0.0;
        double hxj = 0.0, hxi = 0.0;                          // sum_e hpre_e[c] x_j[d], sum_e hpre_e[c] x_i[d]
        for (int q = 0; q < C; ++q) {
            const double wa = W1[(size_t)c * 2 * C + q], wv = (double)W1[(size_t)c * 2 * C + C + q] - wa;
            hxj += wa * xm[4 * q + d] + wv * xm[16 + 4 * q + d];
            hxi += wa * xm[16 + 4 * d + q] + wv * xm[32 + 4 * q + d];
        }
        const double sj = xm[48 + d], si = xm[52 + d];
        const double dA = sc * (T[d] - m1 * sj - m2 * is * (hxj - mu * sj));
        const double dD = sc * (T[4 + d] - m1 * si - m2 * is * (hxi - mu * si));
        dW1[(size_t)c * 2 * C + d] = (float)(dA - dD);
        dW1[(size_t)c * 2 * C + C + d] = (float)dD;
    }
}

static int tw_blocks(long E) { return (int)((E + TW_ROWS - 1) / TW_ROWS); }
size_t tnet_w1_moment_doubles(int P, int k) { return (size_t)tw_blocks((long)P * k) * TN_C1 * 8 + (size_t)((P + 63) / 64) * TW_XM; }
int launch_tnet_bwd_w1_moments(hipStream_t st, const float* dhp, const int* idx, const int* rev_off, const float* x, int ldx, const float* W1,
                               const float* bn1, const float* m1, const float* m2, int P, int N, int k, int C, double* scratch, float* dW1) {
    if (C < 1 || C > 4) return MLSP_ERR_UNSUPPORTED;
    const long E = (long)P * k;
    const int nxb = (P + 63) / 64;
    const int ntb = tw_blocks(E);
    double* tpart = scratch;
    double* xpart = scratch + (size_t)ntb * TN_C1 * 8;
    hipLaunchKernelGGL(tnet_bwd_tmom_kernel, dim3(ntb), dim3(256), 0, st, dhp, idx, x, ldx, E, N, k, C, tpart);
    hipLaunchKernelGGL(tnet_bwd_xmom_kernel, dim3(nxb), dim3(64), 0, st, x, ldx, idx, rev_off, P, N, k, C, xpart);
    hipLaunchKernelGGL(tnet_bwd_w1_finish_kernel, dim3(TN_C1), dim3(256), 0, st, (const double*)tpart, ntb, (const double*)xpart, nxb, W1, bn1, m1,
                       m2, C, dW1);
    return mlsp_launch_status();
}
