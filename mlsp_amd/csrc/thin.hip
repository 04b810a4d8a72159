// "Thin" GEMMs: one dimension of the contraction is tiny (3 coordinates, 3 / 16 outputs) while the row count is the point count.
// On the 128x128 MFMA tiles these launches ran at 1-10 TFLOP/s (a 3-deep K or a 3-wide N wastes 97 % of every tile) and cost
// 0.31 ms of the 7 ms step for 0.4 % of its FLOPs (round-1 verdict).  They are pure streaming: every byte of the big operand is
// read or written exactly once, so they are priced against HBM bandwidth, not against the MFMA roof:
//
//   thin_smallk   C[M,N] = A[M,K] * op(B) (+bias),  K <= 16   first layer of the graph stages (x [P,3] -> [u|v] [P,128]), dgrad of
//                                                              the heads' 128->3 / 256->16 output layers
//   thin_smalln   C[M,N] = A[M,K] * op(B) (+bias),  N <= 16   the heads' output layers (128->3, 256->16), dgrad of the first layer
//   thin_tn       C[M,N] = A[K,M]^T * B[K,N], min(M,N) <= 16  their weight gradients (K = point count): column reduction of a
//                                                              [P, <=16] x [P, wide] outer product, partial slabs + fixed-order reduce
//
// fp32 VALU FMAs: the arithmetic is negligible (<= 16 FMAs per loaded or stored float).
#include "common.h"

// ---------------------------------------------------------------------------------------------- small K
// thread = (row, column quad); lanes of a wave cover consecutive quads of a row (16-byte coalesced stores), the K values of the row
// are wave-broadcast loads.  B is staged once per workgroup as Bs[k][n].
// BS (the dgrad of a head's 3- / 16-channel output layer: C is the gradient w.r.t. the previous layer's ACTIVATED output): the result is
// multiplied by that layer's activation derivative and dropout mask (BsDev: its pre-BN output and parameters at C's column 0) before it
// is stored, and the block -- 128 rows -- leaves its column sums of d' and d' * yhat in part [block][2][stat_ld]: see gemm_out_bs.
struct BsDev { const float* y; int ldy; const float* bn; int bnld; float slope, inv_keep; uint32_t thresh, xH; int ld4, col; double* part; int stat_ld; float* amax; };
template <bool TB, bool BS = false, int NT = 256, int KM = 16>
__global__ __launch_bounds__(NT) void thin_smallk_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                          float* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N,
                                                          int K, int rows_per_block, BsDev bs) {
    extern __shared__ __attribute__((aligned(16))) float Bs[];          // [K][N]  (BS: + fp64 reduction scratch behind it)
    const int tid = threadIdx.x;
    for (int i = tid; i < K * N; i += NT) {
        const int k = i / N, n = i - k * N;
        Bs[i] = TB ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n];
    }
    __syncthreads();
    const int nq = N >> 2;
    const int row0 = blockIdx.x * rows_per_block;
    // BS: a thread keeps ONE column quad (NT % nq == 0): its scale / shift / mean / invstd and its running sums live in registers
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sf = sc, mu = sc, is = sc;
    double ps[4] = {0, 0, 0, 0}, pq[4] = {0, 0, 0, 0};
    float pm[4] = {0.f, 0.f, 0.f, 0.f};                              // BS: column maxima of |d'|
    if (BS) {
        const int c = 4 * (tid % nq);
        sc = *(const f32x4*)(bs.bn + c); sf = *(const f32x4*)(bs.bn + bs.bnld + c);
        mu = *(const f32x4*)(bs.bn + 2 * bs.bnld + c); is = *(const f32x4*)(bs.bn + 3 * bs.bnld + c);
    }
    if constexpr (BS && KM > 0) {
        // U rows per thread in flight: their A values (K <= KM broadcast loads; KM = 16: 16-byte loads, K % 4 == 0) and y quads are all
        // issued before the first store (the stores to C may alias bs.y as far as the compiler knows: row by row, every load waited for
        // the previous row's store).  KM = 0: the row-by-row loop below (K > 4 with an unaligned A).
        const int q = tid % nq, rstep = NT / nq;
        constexpr int U = KM == 16 ? 2 : 4;                  // rows in flight (KM = 16: 32 A values per row pair; 128 registers at 1024 threads)
        for (int rb = tid / nq; rb < rows_per_block; rb += U * rstep) {
            f32x4 yv[U];
            float av[U][KM > 0 ? KM : 1];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rl = rb + u * rstep, r = row0 + rl;
                ok[u] = rl < rows_per_block && r < M;
                const int rr = ok[u] ? r : row0;
                yv[u] = *(const f32x4*)(bs.y + (size_t)rr * bs.ldy + 4 * q);
                if constexpr (KM == 16) {
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        f32x4 t = {0.f, 0.f, 0.f, 0.f};
                        if (4 * k4 < K) t = *(const f32x4*)(A + (size_t)rr * lda + 4 * k4);
                        av[u][4 * k4] = t[0]; av[u][4 * k4 + 1] = t[1]; av[u][4 * k4 + 2] = t[2]; av[u][4 * k4 + 3] = t[3];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < KM; ++k) av[u][k] = k < K ? A[(size_t)rr * lda + k] : 0.f;
                }
            }
            f32x4 acc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                if (k < K) {
                    const f32x4 bv = *(const f32x4*)(Bs + k * N + 4 * q);
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[u][e] = fmaf(av[u][k], bv[e], acc[u][e]);
                }
            }
            f32x4 bb = {0.f, 0.f, 0.f, 0.f};
            if (bias) bb = *(const f32x4*)(bias + 4 * q);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                const int r = row0 + rb + u * rstep;
                const uint32_t hq = bs.thresh ? mix32(((uint32_t)r * (uint32_t)bs.ld4 + ((uint32_t)(bs.col + 4 * q) >> 2)) ^ bs.xH) : 0u;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = acc[u][e] + bb[e];
                    if (bs.thresh) d = ((hq >> (8 * e)) & 255u) >= bs.thresh ? d * bs.inv_keep : 0.f;
                    const float a1 = fmaf(yv[u][e], sc[e], sf[e]);
                    if (!(a1 > 0.f)) d *= bs.slope;
                    ps[e] += d; pq[e] += (double)d * ((yv[u][e] - mu[e]) * is[e]);
                    pm[e] = fmaxf(pm[e], fabsf(d));
                    o[e] = d;
                }
                *(f32x4*)(C + (size_t)r * ldc + 4 * q) = o;
            }
        }
    } else
    for (int i = tid; i < rows_per_block * nq; i += NT) {
        const int r = row0 + i / nq, q = i % nq;
        if (r >= M) break;
        const float* a = A + (size_t)r * lda;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < K; ++k) {
            const float av = a[k];
            const f32x4 bv = *(const f32x4*)(Bs + k * N + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(av, bv[e], acc[e]);
        }
        if (bias) {
            const f32x4 bb = *(const f32x4*)(bias + 4 * q);
            acc = acc + bb;
        }
        if (BS) {
            const f32x4 y = *(const f32x4*)(bs.y + (size_t)r * bs.ldy + 4 * q);
            const uint32_t hq = bs.thresh ? mix32(((uint32_t)r * (uint32_t)bs.ld4 + ((uint32_t)(bs.col + 4 * q) >> 2)) ^ bs.xH) : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d = acc[e];
                if (bs.thresh) d = ((hq >> (8 * e)) & 255u) >= bs.thresh ? d * bs.inv_keep : 0.f;
                const float av = fmaf(y[e], sc[e], sf[e]);
                if (!(av > 0.f)) d *= bs.slope;
                ps[e] += d; pq[e] += (double)d * ((y[e] - mu[e]) * is[e]);
                pm[e] = fmaxf(pm[e], fabsf(d));
                acc[e] = d;
            }
        }
        *(f32x4*)(C + (size_t)r * ldc + 4 * q) = acc;
    }
    if (BS) {       // the block's column sums: the NT / nq threads of a quad, in thread order, through LDS
        __syncthreads();
        double* red = (double*)(Bs + ((K * N + 1) & ~1));               // [NT][8]
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = ps[e]; red[tid * 8 + 4 + e] = pq[e]; }
        __syncthreads();
        if (tid < nq) {
            double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int g2 = 0; g2 < NT / nq; ++g2)
#pragma unroll
                for (int e = 0; e < 8; ++e) s8[e] += red[(g2 * nq + tid) * 8 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bs.part[((size_t)blockIdx.x * 2 + 0) * bs.stat_ld + 4 * tid + e] = s8[e];
                bs.part[((size_t)blockIdx.x * 2 + 1) * bs.stat_ld + 4 * tid + e] = s8[4 + e];
            }
        }
        if (bs.amax) {      // the block's column maxima of |d'| (a by-product for the producer's two-piece f16 products: gemm.hip GemmArgs bs_amax)
            __syncthreads();
            float* redf = (float*)red;
#pragma unroll
            for (int e = 0; e < 4; ++e) redf[tid * 4 + e] = pm[e];
            __syncthreads();
            if (tid < nq) {
                float m4[4] = {0.f, 0.f, 0.f, 0.f};
                for (int g2 = 0; g2 < NT / nq; ++g2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) m4[e] = fmaxf(m4[e], redf[(g2 * nq + tid) * 4 + e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) bs.amax[(size_t)blockIdx.x * bs.stat_ld + 4 * tid + e] = m4[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- small N
// 16 lanes per row: lane g takes the k-quads g, g+16, ... of the row (16-byte coalesced loads), keeps N partial dot products, and the
// 16 partials of a row are summed with xor-shuffles in a fixed order.  B is staged as Bs[n][k].
// XF: A holds the PRE-BatchNorm output of the previous layer (GemmXf / XfDev): every loaded quad becomes that layer's activated output
// first (scale / shift of the K channels staged behind Bs; one dropout hash per quad).
template <bool TB, int NMAX, bool XF = false>
__global__ __launch_bounds__(256) void thin_smalln_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                          float* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N,
                                                          int K, XfDev xf) {
    extern __shared__ __attribute__((aligned(16))) float Bs[];          // [N][K] (+ XF: scale [K], shift [K])
    const int tid = threadIdx.x;
    for (int i = tid; i < K * N; i += 256) {
        const int n = i / K, k = i - n * K;
        Bs[i] = TB ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n];
    }
    float* xs = Bs + K * N;
    if (XF) for (int i = tid; i < K; i += 256) { xs[i] = xf.scale[i]; xs[K + i] = xf.shift[i]; }
    __syncthreads();
    const int g = tid & 15, sub = tid >> 4;                              // 16 row slots per workgroup pass
    const int kq = K >> 2;
    const float bv_out = (bias && g < N) ? bias[g] : 0.f;
    for (int r0 = blockIdx.x * 32 + sub; r0 < M; r0 += gridDim.x * 32) {     // two rows (r0, r0 + 16) per slot and iteration
        const int r1 = r0 + 16;
        const float* a0 = A + (size_t)r0 * lda;
        const float* a1 = A + (size_t)(r1 < M ? r1 : r0) * lda;
        float acc0[NMAX], acc1[NMAX];
#pragma unroll
        for (int n = 0; n < NMAX; ++n) { acc0[n] = 0.f; acc1[n] = 0.f; }
        for (int q = g; q < kq; q += 16) {
            f32x4 x0 = *(const f32x4*)(a0 + 4 * q), x1 = *(const f32x4*)(a1 + 4 * q);
            if (XF) {
                const f32x4 sc = *(const f32x4*)(xs + 4 * q), sh = *(const f32x4*)(xs + K + 4 * q);
                const uint32_t q0 = (uint32_t)(((uint64_t)r0 * xf.ld + xf.col + 4 * q) >> 2), q1 = (uint32_t)(((uint64_t)(r1 < M ? r1 : r0) * xf.ld + xf.col + 4 * q) >> 2);
                x0 = xf_apply_quad(x0, sc, sh, xf.slope, xf.thresh, xf.inv_keep, xf.thresh ? mix32(q0 ^ xf.xH) : 0u);
                x1 = xf_apply_quad(x1, sc, sh, xf.slope, xf.thresh, xf.inv_keep, xf.thresh ? mix32(q1 ^ xf.xH) : 0u);
            }
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                if (n < N) {
                    const f32x4 bv = *(const f32x4*)(Bs + n * K + 4 * q);
                    acc0[n] = fmaf(x0[3], bv[3], fmaf(x0[2], bv[2], fmaf(x0[1], bv[1], fmaf(x0[0], bv[0], acc0[n]))));
                    acc1[n] = fmaf(x1[3], bv[3], fmaf(x1[2], bv[2], fmaf(x1[1], bv[1], fmaf(x1[0], bv[0], acc1[n]))));
                }
            }
        }
        float out0 = 0.f, out1 = 0.f;
#pragma unroll
        for (int n = 0; n < NMAX; ++n) {
            if (n < N) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { acc0[n] += __shfl_xor(acc0[n], o, 64); acc1[n] += __shfl_xor(acc1[n], o, 64); }
            }
            out0 = (g == n) ? acc0[n] : out0;
            out1 = (g == n) ? acc1[n] : out1;
        }
        // lane g writes column g (N <= 16): a 4N-byte contiguous run per row
        if (g < N) {
            C[(size_t)r0 * ldc + g] = out0 + bv_out;
            if (r1 < M) C[(size_t)r1 * ldc + g] = out1 + bv_out;
        }
    }
}

// ---------------------------------------------------------------------------------------------- A^T B with one small side
// D[s][l] = sum_k S[k][s] * L[k][l]  (S: [K, ns <= 16], L: [K, nl], nl % 4 == 0).  A workgroup owns a K chunk: thread = (column quad
// of L, row group); it accumulates ns x 4 partials over its rows, the row groups are summed through LDS in a fixed order and the
// chunk's partial goes to slab[chunk] in the layout of C (ds = stride of s, dl = stride of l), reduced afterwards in slab order.
// XF: the LARGE side L holds the pre-BatchNorm output of the previous layer (the weight gradient of a layer whose input activation was
// never materialised): this thread's channel quad keeps its scale / shift in registers, every loaded quad is transformed first.
template <int NS, bool XF = false>
__global__ __launch_bounds__(256) void thin_tn_kernel(const float* __restrict__ S, int lds_, const float* __restrict__ L, int ldl,
                                                      float* __restrict__ slab, size_t slab_stride, int ds, int dl, int K, int ns, int nl,
                                                      int rows_per_chunk, XfDev xf) {
    extern __shared__ __attribute__((aligned(16))) float red[];         // [row groups][NS][nl]
    const int tid = threadIdx.x;
    const int nq = nl >> 2, ngr = 256 / nq;                              // nq in {32, 64, 128}: 8 / 4 / 2 row groups
    const int q = tid % nq, gr = tid / nq;
    const int k0 = blockIdx.x * rows_per_chunk, k1 = min(K, k0 + rows_per_chunk);
    f32x4 acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (gr < ngr) {
        f32x4 xsc = {0.f, 0.f, 0.f, 0.f}, xsh = {0.f, 0.f, 0.f, 0.f};
        if (XF) { xsc = *(const f32x4*)(xf.scale + 4 * q); xsh = *(const f32x4*)(xf.shift + 4 * q); }
        for (int kb = k0 + gr; kb < k1; kb += 4 * ngr) {                 // four rows per thread in flight
            f32x4 lv[4];
            float sv[4][NS];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kb + u * ngr;
                const bool ok = k < k1;
                lv[u] = ok ? *(const f32x4*)(L + (size_t)k * ldl + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
                if (XF && ok) {
                    const uint32_t qi = (uint32_t)(((uint64_t)k * xf.ld + xf.col + 4 * q) >> 2);
                    lv[u] = xf_apply_quad(lv[u], xsc, xsh, xf.slope, xf.thresh, xf.inv_keep, xf.thresh ? mix32(qi ^ xf.xH) : 0u);
                }
                const float* sp = S + (size_t)(ok ? k : k0) * lds_;
#pragma unroll
                for (int s = 0; s < NS; ++s) sv[u][s] = (s < ns) ? sp[s] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[s][e] = fmaf(sv[u][s], lv[u][e], acc[s][e]);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (s < ns) *(f32x4*)(red + ((size_t)gr * NS + s) * nl + 4 * q) = acc[s];
    }
    __syncthreads();
    float* out = slab + (size_t)blockIdx.x * slab_stride;
    for (int i = tid; i < ns * nl; i += 256) {
        const int s = i / nl, l = i - s * nl;
        float v = 0.f;
        for (int g2 = 0; g2 < ngr; ++g2) v += red[((size_t)g2 * NS + s) * nl + l];
        out[(size_t)s * ds + (size_t)l * dl] = v;
    }
}

int launch_slab_reduce(hipStream_t st, const float* slab, float* C, int M, int N, int ldc, int nsplit);
#ifndef THIN_SN_BLOCKS
#define THIN_SN_BLOCKS 1024  // workgroups of the small-N kernel at most (512: 24.9 / 8.1 us, 1024: 21.2 / 7.2 us for the 16- / 3-channel layers)
#endif
#ifndef THIN_TN_ROWS
#define THIN_TN_ROWS 64      // rows of the K dimension per workgroup (one partial slab each)
#endif

// rows of the K dimension per workgroup: 64 with 16 accumulator quads per thread, 32 with 4 (measured: 6.8 -> 6.2 us for the 3-channel
// layers, 21.4 -> 27.7 us for the 16-channel one at 32)
static inline int thin_tn_rows(int ns) { return ns <= 4 ? THIN_TN_ROWS / 2 : THIN_TN_ROWS; }
// slab floats the thin TN path needs for (M, N, K) (0: shape not handled here)
size_t thin_tn_slab_floats(int M, int N, int K) {
    const bool small_m = M <= 16 && N % 4 == 0 && N >= 64 && N <= 512 && (N & (N - 1)) == 0;
    const bool small_n = N <= 16 && M % 4 == 0 && M >= 64 && M <= 512 && (M & (M - 1)) == 0;
    if (!(small_m || small_n) || K < 2048) return 0;
    const int rows = thin_tn_rows(small_m ? M : N);
    const int chunks = (K + rows - 1) / rows;
    return (size_t)chunks * M * N;
}

// Returns MLSP_ERR_UNSUPPORTED when the shape is not thin (the caller continues with the MFMA kernels).
// xf (nullable): operand transform (common.h GemmXf); which == 1: A of the small-N kernel, which == 2: the wide B of the A^T B kernel.
// Any other combination with a transform: MLSP_ERR_UNSUPPORTED.
// row panels (128 rows each) the small-K kernel writes partial sums for under a GemmBs; 0: not this kernel's shape
int thin_bs_parts(int M, int N, int K) {
    return (K <= 16 && N % 4 == 0 && N >= 64 && N <= 512 && M >= 1024 && 256 % (N / 4) == 0 && M % 128 == 0) ? M / 128 : 0;
}
int launch_thin_gemm(hipStream_t st, bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                     int ldc, const float* bias, float* slab, size_t slab_floats, const GemmXf* xf, const GemmBs* bs) {
    auto al16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    XfDev xd = {nullptr, nullptr, 1.f, 1.f, 0u, 0u, 0, 0};
    if (xf) {
        if ((xf->ld & 3) || (xf->col & 3) || !al16(xf->scale) || !al16(xf->shift) || (double)(xf->which == 1 ? M : K) * xf->ld >= 17179869184.0) return MLSP_ERR_UNSUPPORTED;
        xd = xf_dev(*xf);
    }
    if (!xf && !ta && K <= 16 && N % 4 == 0 && N >= 64 && N <= 512 && M >= 1024 && ldc % 4 == 0 && al16(C) && al16(bias)) {
        const int nq = N / 4;
        if (256 % nq) return MLSP_ERR_UNSUPPORTED;
        BsDev bd = {nullptr, 0, nullptr, 0, 1.f, 1.f, 0u, 0u, 0, 0, nullptr, 0, nullptr};
        if (bs) {
            if (thin_bs_parts(M, N, K) == 0 || !bs->y || !bs->bn || !bs->part || (bs->ld & 3) || (bs->col & 3) || (bs->ldy & 3) || (bs->bnld & 3) ||
                !al16(bs->y) || !al16(bs->bn) || (double)M * bs->ld >= 17179869184.0) return MLSP_ERR_UNSUPPORTED;
            bd.y = bs->y; bd.ldy = bs->ldy; bd.bn = bs->bn; bd.bnld = bs->bnld; bd.slope = bs->act == 0 ? 1.f : bs->act == 1 ? 0.f : bs->slope;
            bd.inv_keep = bs->inv_keep; bd.thresh = bs->thresh; bd.xH = mix32_host((uint32_t)bs->seed) ^ (uint32_t)(bs->seed >> 32) * 0x9e3779b9U;
            bd.ld4 = bs->ld / 4; bd.col = bs->col; bd.part = bs->part; bd.stat_ld = bs->stat_ld; bd.amax = bs->amax;
        }
        const int rpb = bs ? 128 : 4 * 256 / nq;                          // four passes of the workgroup per block (fused statistics: one 128-row panel)
        // (fused statistics: one block per 128-row panel -- 1024 threads, so that M / 128 blocks still fill the chip)
        const size_t lds = (size_t)((K * N + 1) & ~1) * sizeof(float) + (bs ? 1024 * 8 * sizeof(double) : 0);
        const dim3 grid((M + rpb - 1) / rpb);
        if (bs) {
            // (K = 16, N = 256: two rows in flight with 16-byte A loads 25.6 us, the row-by-row loop 32.5 us; K = 3: 14.5 against 15.5)
            const bool av16 = K % 4 == 0 && lda % 4 == 0 && al16(A);
            auto kern = K <= 4 ? (tb ? thin_smallk_kernel<true, true, 1024, 4> : thin_smallk_kernel<false, true, 1024, 4>)
                      : av16 ? (tb ? thin_smallk_kernel<true, true, 1024, 16> : thin_smallk_kernel<false, true, 1024, 16>)
                             : (tb ? thin_smallk_kernel<true, true, 1024, 0> : thin_smallk_kernel<false, true, 1024, 0>);
            if (lds > 64 * 1024) {
                hipError_t e_ = mlsp_lds_limit((const void*)kern, lds);
                if (e_ != hipSuccess) return (int)e_;
            }
            hipLaunchKernelGGL(kern, grid, dim3(1024), lds, st, A, lda, B, ldb, C, ldc, bias, M, N, K, rpb, bd);
        } else if (tb) hipLaunchKernelGGL((thin_smallk_kernel<true, false>), grid, dim3(256), lds, st, A, lda, B, ldb, C, ldc, bias, M, N, K, rpb, bd);
        else hipLaunchKernelGGL((thin_smallk_kernel<false, false>), grid, dim3(256), lds, st, A, lda, B, ldb, C, ldc, bias, M, N, K, rpb, bd);
        return mlsp_launch_status();
    }
    if (bs) return MLSP_ERR_UNSUPPORTED;
    if ((!xf || xf->which == 1) && !ta && N <= 16 && K % 4 == 0 && K >= 64 && K <= 1024 && M >= 1024 && lda % 4 == 0 && al16(A)) {
        const size_t lds = (size_t)K * (N + (xf ? 2 : 0)) * sizeof(float);
        int blocks = (M + 31) / 32;
        if (blocks > THIN_SN_BLOCKS) blocks = THIN_SN_BLOCKS;                                      // B is staged once per workgroup: keep many rows per workgroup
#define THIN_SN(TBV, XFV) do { if (N <= 4) hipLaunchKernelGGL((thin_smalln_kernel<TBV, 4, XFV>), dim3(blocks), dim3(256), lds, st, A, lda, B, ldb, C, ldc, bias, M, N, K, xd); \
                               else hipLaunchKernelGGL((thin_smalln_kernel<TBV, 16, XFV>), dim3(blocks), dim3(256), lds, st, A, lda, B, ldb, C, ldc, bias, M, N, K, xd); } while (0)
        if (xf) { if (tb) THIN_SN(true, true); else THIN_SN(false, true); }
        else if (tb) THIN_SN(true, false); else THIN_SN(false, false);
#undef THIN_SN
        return mlsp_launch_status();
    }
    if (xf && xf->which != 2) return MLSP_ERR_UNSUPPORTED;
    if (ta && !tb && !bias) {
        const size_t need = thin_tn_slab_floats(M, N, K);
        if (!need || !slab || slab_floats < need) return MLSP_ERR_UNSUPPORTED;
        const bool small_m = M <= 16 && N % 4 == 0 && N >= 64;
        const int tn_rows = thin_tn_rows(small_m ? M : N);
        const int chunks = (K + tn_rows - 1) / tn_rows;
        if (small_m ? (ldb % 4 || !al16(B)) : (lda % 4 || !al16(A))) return MLSP_ERR_UNSUPPORTED;   // the wide operand is read 16 bytes per lane
        if (xf && !small_m) return MLSP_ERR_UNSUPPORTED;                                            // the transform is on B = the wide side
        // small side S, large side L; partial slabs are written in C's [M][N] layout
        const float* S = small_m ? A : B; const int lds_ = small_m ? lda : ldb; const int ns = small_m ? M : N;
        const float* L = small_m ? B : A; const int ldl = small_m ? ldb : lda; const int nl = small_m ? N : M;
        const int ds = small_m ? N : 1, dl = small_m ? 1 : N;
        const int ngr = 256 / (nl / 4);
#define THIN_TN(NSV, XFV) hipLaunchKernelGGL((thin_tn_kernel<NSV, XFV>), dim3(chunks), dim3(256), (size_t)ngr * NSV * nl * sizeof(float), st, S, lds_, L, ldl, \
                                             slab, (size_t)M * N, ds, dl, K, ns, nl, tn_rows, xd)
        if (xf) { if (ns <= 4) THIN_TN(4, true); else THIN_TN(16, true); }
        else if (ns <= 4) THIN_TN(4, false); else THIN_TN(16, false);
#undef THIN_TN
        int rc = mlsp_launch_status();
        if (rc != MLSP_OK) return rc;
        return launch_slab_reduce(st, slab, C, M, N, ldc, chunks);
    }
    return MLSP_ERR_UNSUPPORTED;
}

// does launch_thin_gemm take this contraction WITH an operand transform on `which` (1: A of the small-N kernel, 2: the wide B of A^T B)?
// (conditions of the two branches above; the slab is the caller's: gemm_slab_floats covers thin_tn_slab_floats)
bool thin_xf_supported(bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, int which) {
    auto al16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    if (which == 1) return !ta && N <= 16 && K % 4 == 0 && K >= 64 && K <= 1024 && M >= 1024 && lda % 4 == 0 && al16(A);
    if (which == 2) return ta && !tb && thin_tn_slab_floats(M, N, K) != 0 && M <= 16 && N % 4 == 0 && N >= 64 && ldb % 4 == 0 && al16(B);
    return false;
}
