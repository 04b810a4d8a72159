// kNN v6 (k <= 24, N % 128 == 0, N <= 4096, C <= 128): barrier-free approximate sweeps + exact resolution of the ambiguous survivors only.
// Replaces PointDA/model_utils.py:9-16 `knn` (twin PointSegDA/Models.py:8-15) for the wide graph stages of DGCNN (C = 64, 64, 128 at
// k = 20); same canonical arithmetic and total order as knn.hip / oracle/knn_canon.c, indices bit-exact.
//
// What v5 (knn.hip) spends its time on: one __syncthreads per 32-candidate tile (LDS staging shared by 8 waves), an exact f32-MFMA second
// sweep (64 matrix cycles per 2 channels), and a ballot / popcount / branch sequence per accumulator register in the survivor pass
// (~700 instructions per tile).  Here:
//   * knn6_prep_kernel splits every point ONCE into bf16 hi / lo pieces next to the canonical squared norms (it replaces sqnorm_kernel),
//     stored FRAGMENT-major: [32-point tile][16-channel block][hi | lo][half h][row][8 bf16] -- the 64 lanes of a wave fetch one MFMA
//     operand as ONE contiguous KiB (16 bytes per lane, every cache line used whole).  The sweeps load their fragments straight from
//     that image (L2-resident) into a register ring: no LDS staging, no conversion, NO barrier inside a sweep.
//   * TRANSPOSED tiles: A = 32 candidates, B = 32 queries, so a lane owns ONE query (column) and its 16 accumulator registers are 16
//     candidates of the tile.  Per-query state (threshold, survivor cursor) is per-lane: the survivor pass is v_cmpx / ds_write2 / add per
//     pair, no ballots, no scalar branches.  The accumulator starts at -xx_j / 2 (one LDS read per 4 rows), so a = dot' - xx_j / 2
//     orders like the distance and pd' = 2 a - xx_q.  A wave carries TWO query groups (64 queries) against half the candidates: every
//     fragment it loads feeds both (the vector-memory path, 64 B / clk / CU, is what bounds one group per wave).
//   * BOTH sweeps use the split-bf16 products hi hi + lo hi + hi lo.  Error budget (worst case, not typical): bf16 has 8 significant
//     bits, so |x - hi| <= 2^-8 |x|, |x - hi - lo| <= 2^-16 |x|; the dropped terms (lo lo, the two residual products) are at most
//     3 x 2^-16 |x_q,c x_j,c| per channel, i.e. |2 dot' - 2 dot| <= 3 x 2^-16 (xx_q + xx_j) = 2^-14.4 (xx_q + xx_j); the fp32 accumulation
//     inside and between the 3 C / 16 MFMAs adds at most 2^-16.1 (xx_q + xx_j) at C = 128; the canonical value's own two roundings
//     2^-22.  Sum < 2^-13.9 (xx_q + xx_j): E = 2^-13 (xx_q + X), X >= xx_j, bounds it with a factor 1.9 to spare.
//     Pass A: 64 running maxima per query (16 registers x 2 half-wave lanes x 2 candidate halves), tau = their k-th largest: at least k
//     candidates have pd' >= tau, so the true k-th best is >= tau - E and every true neighbour has pd' >= tau - 2 E (X = the cloud's
//     largest squared norm): pass B keeps exactly those (~1.4 k of N).
//   * Exactness: a survivor whose pd' is farther than 2 E from every other survivor's (X = the largest norm among the query's
//     survivors) has its rank decided by pd' alone: F1 counts, per survivor, the survivors ahead of it and those inside its 2 E window.
//     The others (~10-20 %) get their canonical distance from an fmaf chain over the fp32 rows (lane per pair, batched per wave), then
//     a recount under the full order (value desc, index asc) over lists that mix exact and approximate values -- safe, because an
//     unambiguous value differs from every other by more than 2 E and an exact one from its approximation by at most E.
//   * Any list overflow (massive ties: more than 24 survivors in a quarter of a query's candidates) or NaN / inf bound sends the whole
//     workgroup through an exact path: f32 MFMA tiles from the fp32 rows, per-lane sorted top-k lists in registers, the same final.
// Measured (B = 32, N = 1024, k = 20, one call incl. the prep kernel): C = 64 78 us (v5 113), C = 128 104 us (v5 176); C = 3 60 us (v5 49:
// the selection phases run one wave per SIMD here and dominate when the sweeps are trivial, so C <= 16 stays on v5).  Phase cycles per
// wave at C = 64 (tools/knn6_stamps.py): sweeps 20 k + 43 k, tau 5 k, list conversion 9 k, F1 43 k, F2 10 k.
#include "common.h"
#include <math.h>
#include <type_traits>

typedef __bf16 k6bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int k6u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int k6u32x4 __attribute__((ext_vector_type(4)));
typedef int k6i32x4 __attribute__((ext_vector_type(4)));

#define K6_KMAX 24          // largest k
#define K6_CAP 24           // survivors one (query, quarter) list keeps
#define K6_LENT 28          // entries of a list: the cursor is clamped every 4 appends
#define K6_LSTR 232         // bytes between lists (58 dwords: 16 consecutive lists start in 16 different even banks)
#define K6_XS 68            // floats per query of the tau exchange image (16-byte aligned rows, 4-bank skew)
#define K6_EPS 1.220703125e-04f      // 2^-13 (error budget in the header)
#define K6_WL 512           // items of a wave's work list of ambiguous survivors (4 bytes each)

// byte offset of the 16-byte piece (point row of its tile, half h) of block kb, plane p (0 hi, 1 lo) of tile T
__device__ __forceinline__ size_t k6_piece(size_t T, int nkb, int kb, int p, int h, int row) {
    return (((T * nkb + kb) * 2 + p) * 2 + h) * 512 + (size_t)row * 16;
}

// canonical squared norms xx (fmaf chain over the raw coordinates, c ascending: what the exact distances use) + the fragment-major bf16
// hi / lo image of every point RELATIVE TO THE FIRST POINT OF ITS CLOUD and the squared norms xc of those differences (P % 32 == 0).
// Distances do not change under a translation, the error of the split products scales with the NORMS of what is multiplied: features
// after BatchNorm + LeakyReLU + max sit far from the origin compared with their spread (squared norm ~14x the centred one at the first
// EdgeConv output), and bounds in terms of the raw norms would declare most survivors ambiguous.
template <int CT>
__global__ __launch_bounds__(256) void knn6_prep_kernel(const float* __restrict__ x, int ld, int P, int N, int C, float* __restrict__ xx,
                                                        float* __restrict__ xc, char* __restrict__ planes) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float* r = x + (size_t)i * ld;
    const float* r0 = x + (size_t)(i / N) * N * ld;           // first point of the cloud
    const bool vec = (ld & 3) == 0 && (((uintptr_t)x) & 15) == 0;
    const size_t T = (size_t)(i >> 5);
    const int row = i & 31;
    float acc = 0.f, accc = 0.f;
#pragma unroll
    for (int c0 = 0; c0 < CT; c0 += 8) {
        float v[8], o[8];
        if (vec && c0 + 8 <= C) {
            const f32x4 a = *(const f32x4*)(r + c0), b = *(const f32x4*)(r + c0 + 4);
            const f32x4 a0 = *(const f32x4*)(r0 + c0), b0 = *(const f32x4*)(r0 + c0 + 4);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
            o[0] = a0[0]; o[1] = a0[1]; o[2] = a0[2]; o[3] = a0[3]; o[4] = b0[0]; o[5] = b0[1]; o[6] = b0[2]; o[7] = b0[3];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = (c0 + e < C) ? r[c0 + e] : 0.f; o[e] = (c0 + e < C) ? r0[c0 + e] : 0.f; }
        }
        k6bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (c0 + e < C) acc = fmaf(v[e], v[e], acc);
            const float d = v[e] - o[e];
            accc = fmaf(d, d, accc);
            const __bf16 hv = (__bf16)d;
            hi[e] = hv;
            lo[e] = (__bf16)(d - (float)hv);
        }
        *(k6bf16x8*)(planes + k6_piece(T, CT / 16, c0 >> 4, 0, (c0 >> 3) & 1, row)) = hi;
        *(k6bf16x8*)(planes + k6_piece(T, CT / 16, c0 >> 4, 1, (c0 >> 3) & 1, row)) = lo;
    }
    xx[i] = acc;
    xc[i] = accc;
}

template <int I, int Nn, class F>
__device__ __forceinline__ void k6_static_for(F&& f) {
    if constexpr (I < Nn) { f(std::integral_constant<int, I>{}); k6_static_for<I + 1, Nn>(f); }
}

// candidate beats list entry (value desc, index asc); false for a NaN candidate
__device__ __forceinline__ bool k6_beats(float d, int j, float pv, int pi) { return d > pv || (d == pv && j < pi); }

// Workgroup = 4 waves = 128 queries x all N candidates of their cloud.  Wave w: dg = w & 1 picks 64 queries (two 32-query groups A / B
// that share every candidate fragment the wave loads: half the fragment traffic per MFMA, and group A's selection work runs under
// group B's MFMAs), ch = w >> 1 the half of the candidates it sweeps.  One wave per SIMD (the register file is the prefetch buffer:
// K6_PF tiles of fragments in flight per wave), no barrier inside a sweep.
template <int CT>
__global__ __launch_bounds__(256, CT <= 16 ? 2 : 1) void knn6_kernel(const float* __restrict__ x, int ld, const float* __restrict__ xx_all, const float* __restrict__ xc_all,
                                                   const char* __restrict__ planes, int N, int C, int k, int* __restrict__ idx, int B) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NKB = CT / 16;                              // 16-channel blocks = bf16 MFMA K steps per tile
    constexpr int PF = NKB >= 4 ? 2 : 4;        // tiles of fragments in flight (the ring holds PF * NKB blocks of 2 x 4 registers)
    constexpr int NR = PF * NKB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // (scalar: addresses below stay in SGPRs)
    const int l31 = lane & 31, h = lane >> 5, dg = wave & 1, ch = wave >> 1;
    int b, chunk;
    xcd_cloud_map(blockIdx.x, N / 128, B, b, chunk);
    const float* xxb = xx_all + (size_t)b * N;                // canonical squared norms (raw coordinates)
    const float* xcb = xc_all + (size_t)b * N;                // squared norms relative to the cloud's first point (the sweeps' coordinates)
    const size_t T0 = (size_t)b * (N / 32);                  // first 32-point tile of this cloud in the fragment-major image
    const float* xb = x + (size_t)b * N * ld;
    const int nt2 = N / 64;                                   // 32-candidate tiles of this wave's half

    char* lists = (char*)sm;                                  // [512 lists][K6_LSTR]; list = ((g*2 + ch)*2 + h)*32 + query of group g
    float* xch = sm;                                          // tau exchange image [128 queries][K6_XS], dead before pass B
    float* nxx = (float*)(lists + 512 * K6_LSTR);             // [N]  -xx_j / 2
    float* tauv = nxx + N;                                    // [128]
    int* cnts = (int*)(tauv + 128);                           // [128 queries][4 quarters]
    float* red = (float*)(cnts + 512);                        // [16]
    unsigned* wlbase = (unsigned*)(red + 16);                 // [4 waves][K6_WL] work lists of ambiguous survivors
    float* lmn = (float*)(wlbase + 4 * K6_WL);                          // [128 queries][4 quarters] min of -xx_j / 2 over the list's survivors
    const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)sm);       // LDS byte address of `lists`

#ifdef K6_STAMP
    long long stamp[10];
    int st_items = 0, st_flushes = 0;
    long long st_f2 = 0;
#define K6_T(i_) stamp[i_] = (long long)__builtin_amdgcn_s_memtime()
#else
#define K6_T(i_)
#endif
    K6_T(0);
    float xcmax, xxmax;
    {
        float m = 0.f, mr = 0.f;
        for (int j = tid; j < N; j += 256) { const float v = xcb[j]; nxx[j] = -0.5f * v; m = fmaxf(m, v); mr = fmaxf(mr, xxb[j]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); mr = fmaxf(mr, __shfl_xor(mr, o, 64)); }
        if (lane == 0) { red[wave] = m; red[4 + wave] = mr; }
        __syncthreads();
        xcmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        xxmax = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    }
    // |pd' - pd_canonical| <= K6_EPS (xc_q + xc_j)  [split products on the centred coordinates, header]
    //                        + K6_CANON (xx_q + xx_j)  [the canonical value's own distance from -|x_q - x_j|^2: an fmaf chain of C terms, the two
    //                          norms and two more roundings on the RAW coordinates: < (C + 4) 2^-23 (xx_q + xx_j)]
    const float K6_CANON = (float)(C + 4) * 1.1920929e-07f;
    const int gA = dg * 2, gB = dg * 2 + 1;                   // the wave's two query groups (of the workgroup's four)
    const int qA0 = chunk * 128 + gA * 32;                    // first query of group A, local to the cloud; group B = + 32
    const float xxqA = xxb[qA0 + l31], xxqB = xxb[qA0 + 32 + l31];          // raw (exact path)
    const float xcqA = xcb[qA0 + l31], xcqB = xcb[qA0 + 32 + l31];          // centred (sweeps)
    const float EqA = K6_EPS * (xcqA + xcmax) + K6_CANON * (xxqA + xxmax), EqB = K6_EPS * (xcqB + xcmax) + K6_CANON * (xxqB + xxmax);

    // ---------------------------------------------------------------------------------------------------------------- approximate sweeps
    k6bf16x8 qhA[NKB], qlA[NKB], qhB[NKB], qlB[NKB];          // B operands: query row l31 of each group, channels 16 kb + 8 h .. + 7
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        qhA[kb] = *(const k6bf16x8*)(planes + k6_piece(T0 + (qA0 >> 5), NKB, kb, 0, h, l31));
        qlA[kb] = *(const k6bf16x8*)(planes + k6_piece(T0 + (qA0 >> 5), NKB, kb, 1, h, l31));
        qhB[kb] = *(const k6bf16x8*)(planes + k6_piece(T0 + (qA0 >> 5) + 1, NKB, kb, 0, h, l31));
        qlB[kb] = *(const k6bf16x8*)(planes + k6_piece(T0 + (qA0 >> 5) + 1, NKB, kb, 1, h, l31));
    }
    const unsigned voff = (unsigned)(h * 512 + l31 * 16);     // this lane's 16 bytes inside a fragment KiB
    const char* cand0 = planes + (T0 + (size_t)ch * nt2) * (NKB * 2048);     // (uniform) first tile of this wave's half
    auto frag_load = [&](int tl, int kb, k6bf16x8& ah, k6bf16x8& al) {       // A operand: candidate row l31 of tile tl of this half
        const char* p = cand0 + (size_t)tl * (NKB * 2048) + kb * 2048;                       // uniform; one contiguous KiB per wave and piece
        ah = *(const k6bf16x8*)(p + voff);
        al = *(const k6bf16x8*)(p + 1024 + voff);
    };
    // one sweep over the half: sel(accA, accB, tl) sees the finished 32 x 32 tiles of both groups:
    // acc[r] = a(query l31 of the group, candidate (r & 3) + 8 (r >> 2) + 4 h of the tile)
    // Fragment ring: slot (tile % PF, block) is refilled right after its use with the tile PF ahead; a scheduling barrier after every
    // refill keeps the loads where they are written (the scheduler otherwise sinks them next to their use: prefetch distance zero).
    // The compiler drains the memory counter once per loop trip (its wait-count analysis merges pessimistically at the back edge), so a
    // trip covers UNR tiles: the drain exposes one L2 latency per UNR tiles of MFMA work.
    constexpr int UNR = NKB >= 8 ? 4 : 8;
    auto sweep = [&](auto&& sel) {
        k6bf16x8 fh[NR], fl[NR];
        auto tile = [&](auto SLOT, auto RING, int tl, int tn, auto&& sel_) {  // tn: the tile that refills this ring slot (always loaded: no branch)
            constexpr int s0 = decltype(SLOT)::value * NKB;
            constexpr bool ring = decltype(RING)::value;
            f32x16 accA, accB;
            const float* p = nxx + (ch * nt2 + tl) * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *(const f32x4*)(p + 8 * g);
                accA[4 * g] = v[0]; accA[4 * g + 1] = v[1]; accA[4 * g + 2] = v[2]; accA[4 * g + 3] = v[3];
            }
            accB = accA;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[s0 + kb], qhA[kb], accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[s0 + kb], qhB[kb], accB, 0, 0, 0);
                accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[s0 + kb], qhA[kb], accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[s0 + kb], qhB[kb], accB, 0, 0, 0);
                accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[s0 + kb], qlA[kb], accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[s0 + kb], qlB[kb], accB, 0, 0, 0);
                if (ring) {
                    frag_load(tn, kb, fh[s0 + kb], fl[s0 + kb]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            sel_(accA, accB, tl);
        };
        const int nmain = (nt2 / UNR) * UNR;
        if (nmain > 0) {
#pragma unroll
            for (int tu = 0; tu < PF; ++tu)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) frag_load(tu, kb, fh[tu * NKB + kb], fl[tu * NKB + kb]);
            for (int t0 = 0; t0 < nmain; t0 += UNR) {
                k6_static_for<0, UNR>([&](auto TU) {
                    constexpr int tu = decltype(TU)::value;
                    const int tl = t0 + tu;
                    const int tn = min(tl + PF, nt2 - 1);      // (past the end the last tile is fetched again: harmless, and no branch in the body)
                    tile(std::integral_constant<int, tu % PF>{}, std::true_type{}, tl, tn, sel);
                });
            }
        }
        for (int tl = nmain; tl < nt2; ++tl) {                // ragged tail (N / 64 not a multiple of UNR): one tile at a time through slot 0
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) frag_load(tl, kb, fh[kb], fl[kb]);
            tile(std::integral_constant<int, 0>{}, std::false_type{}, tl, tl, sel);
        }
    };

    K6_T(1);
    // ---- pass A: 16 running maxima per lane and group (acc domain: a = dot' - xx_j / 2 is monotone in the distance for a fixed query)
    {
        float cmA[16], cmB[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { cmA[r] = -INFINITY; cmB[r] = -INFINITY; }
        sweep([&](const f32x16& accA, const f32x16& accB, int) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { cmA[r] = fmaxf(cmA[r], accA[r]); cmB[r] = fmaxf(cmB[r], accB[r]); }
        });
        K6_T(2);
        float* dA = xch + (gA * 32 + l31) * K6_XS + (ch * 2 + h) * 16;
        float* dB = xch + (gB * 32 + l31) * K6_XS + (ch * 2 + h) * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 va = {cmA[4 * g], cmA[4 * g + 1], cmA[4 * g + 2], cmA[4 * g + 3]};
            const f32x4 vb = {cmB[4 * g], cmB[4 * g + 1], cmB[4 * g + 2], cmB[4 * g + 3]};
            *(f32x4*)(dA + 4 * g) = va;
            *(f32x4*)(dB + 4 * g) = vb;
        }
    }
    __syncthreads();
#if defined(K6_PROBE) && K6_PROBE == 1
    return;
#endif
    K6_T(3);
    // tau = k-th largest of a query's 64 maxima: one lane per query (wave w sorts group w with its first half-wave), bitonic network in registers
    if (lane < 32) {
        float v[64];
        const float* src = xch + (wave * 32 + l31) * K6_XS;
#pragma unroll
        for (int i4 = 0; i4 < 16; ++i4) {
            const f32x4 t = *(const f32x4*)(src + 4 * i4);
            v[4 * i4] = t[0]; v[4 * i4 + 1] = t[1]; v[4 * i4 + 2] = t[2]; v[4 * i4 + 3] = t[3];
        }
#pragma unroll
        for (int k2 = 2; k2 <= 64; k2 <<= 1)
#pragma unroll
            for (int j = k2 >> 1; j > 0; j >>= 1)
#pragma unroll
                for (int i = 0; i < 64; ++i) {
                    const int l = i ^ j;
                    if (l > i) {
                        const float lo = fminf(v[i], v[l]), hi = fmaxf(v[i], v[l]);
                        if ((i & k2) == 0) { v[i] = hi; v[l] = lo; }           // descending overall
                        else { v[i] = lo; v[l] = hi; }
                    }
                }
        float t = v[0];
#pragma unroll
        for (int i = 1; i < K6_KMAX; ++i) t = (i == k - 1) ? v[i] : t;
        tauv[wave * 32 + l31] = t;
    }
    __syncthreads();                                          // tau complete; the exchange image (aliases the lists) is dead from here on
#if defined(K6_PROBE) && K6_PROBE == 2
    return;
#endif
    K6_T(4);
    float thrA = tauv[gA * 32 + l31] - EqA, thrB = tauv[gB * 32 + l31] - EqB;  // acc domain: pd' >= tau_pd - 2 E  <=>  a >= a_tau - E
    thrA = thrA == thrA ? thrA : -INFINITY;                   // NaN bound: everything survives, the lists overflow, the exact path takes over
    thrB = thrB == thrB ? thrB : -INFINITY;

    // ---- pass B: survivors -> this lane's private lists (acc-domain value, candidate index); always stored at the cursor, kept when it moves on
    char* LBA = lists + (size_t)((((gA * 2 + ch) * 2 + h) * 32 + l31) * K6_LSTR);
    char* LBB = lists + (size_t)((((gB * 2 + ch) * 2 + h) * 32 + l31) * K6_LSTR);
    // per pair: v_cmpx (exec = survivors) / ds_write2_b32 {value, index} at the cursor / cursor += 8 under that mask / exec back to all lanes:
    // three vector instructions and no branch; the cursor is clamped every 4 appends (lists have 4 entries of slack).
    const unsigned baseA = lds0 + (unsigned)((((gA * 2 + ch) * 2 + h) * 32 + l31) * K6_LSTR);
    const unsigned baseB = lds0 + (unsigned)((((gB * 2 + ch) * 2 + h) * 32 + l31) * K6_LSTR);
    unsigned adA = baseA, adB = baseB, topA = baseA, topB = baseB;
    sweep([&](const f32x16& accA, const f32x16& accB, int tl) {
        const int jb = (ch * nt2 + tl) * 32 + 4 * h;
        int jv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) jv[r] = jb + (r & 3) + 8 * (r >> 2);
        // The compiler does not place the MFMA -> VALU / LDS read wait states for instructions INSIDE an asm statement: a visible VALU read of
        // each accumulator (gate) comes first -- the hazard recognizer pads in front of it -- and every append depends on the gate.
        const float gateA = fmaxf(accA[0], accA[15]), gateB = fmaxf(accB[0], accB[15]);
#define K6_APPEND(ad_, val_, thr_, j_, gate_) asm volatile("v_cmpx_ge_f32_e32 vcc, %1, %2\n\tds_write2_b32 %0, %1, %3 offset1:1\n\tv_add_u32_e32 %0, 8, %0\n\ts_mov_b64 exec, -1" \
                                                           : "+v"(ad_) : "v"(val_), "v"(thr_), "v"(j_), "v"(gate_) : "vcc", "memory")
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            K6_APPEND(adA, accA[r], thrA, jv[r], gateA);
            if ((r & 3) == 3) { topA = max(topA, adA); adA = min(adA, baseA + K6_CAP * 8); }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            K6_APPEND(adB, accB[r], thrB, jv[r], gateB);
            if ((r & 3) == 3) { topB = max(topB, adB); adB = min(adB, baseB + K6_CAP * 8); }
        }
#undef K6_APPEND
    });
    int cntA = (int)(adA - baseA) >> 3, cntB = (int)(adB - baseB) >> 3;
    const int ovf = (topA > baseA + K6_CAP * 8 || topB > baseB + K6_CAP * 8) ? K6_CAP + 1 : 0;
    K6_T(5);
    bool exact_lists = false;
    if (__syncthreads_or(ovf > K6_CAP ? 1 : 0)) {
        // ------------------------------------------------------------------------------------------------------------ exact path (rare)
        // f32 MFMA tiles straight from the fp32 rows (same transposed layout; the MFMA chain is the canonical fmaf chain, channels
        // ascending), every lane keeps the exact top-K6_KMAX of ITS quarter of the candidates in a sorted register list; group A, then B.
        const bool vec = (ld & 3) == 0 && (((uintptr_t)x) & 15) == 0 && (C & 3) == 0;
        // fragment of 32 channels [c0, c0 + 32) of a row for the f32 MFMA: f[s] = row[c0 + 2 s + h], zero beyond C (16 K-steps of 2 channels)
        auto load_frag = [&](const float* row, int c0, float (&f)[16]) {
            if (vec) {
#pragma unroll
                for (int m4 = 0; m4 < 8; ++m4) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (c0 + 4 * m4 < C) v = *(const f32x4*)(row + c0 + 4 * m4);
                    f[2 * m4] = h ? v[1] : v[0];
                    f[2 * m4 + 1] = h ? v[3] : v[2];
                }
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) f[s] = (c0 + 2 * s + h < C) ? row[c0 + 2 * s + h] : 0.f;
            }
        };
        for (int sg = 0; sg < 2; ++sg) {
            const float* qrowp = xb + (size_t)(qA0 + 32 * sg + l31) * ld;
            const float xxq = sg ? xxqB : xxqA;
            float tv[K6_KMAX];
            int ti[K6_KMAX];
#pragma unroll
            for (int s = 0; s < K6_KMAX; ++s) { tv[s] = -INFINITY; ti[s] = 0x7fffffff; }
            for (int tl = 0; tl < nt2; ++tl) {
                const int j0 = (ch * nt2 + tl) * 32;
                const float* crowp = xb + (size_t)(j0 + l31) * ld;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                for (int c0 = 0; c0 < C; c0 += 32) {           // channels ascending: the canonical chain (query fragments re-read: rare path)
                    float ca[16], qa[16];
                    load_frag(crowp, c0, ca);
                    load_frag(qrowp, c0, qa);
#pragma unroll
                    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[s], qa[s], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = j0 + 4 * h + (r & 3) + 8 * (r >> 2);
                    const float xxj = xxb[j];                                // canonical norm of the raw row
                    const float pd = fmaf(2.0f, acc[r], -xxj) - xxq;
                    if (k6_beats(pd, j, tv[K6_KMAX - 1], ti[K6_KMAX - 1])) {
                        bool bs[K6_KMAX];
#pragma unroll
                        for (int s = 0; s < K6_KMAX; ++s) bs[s] = k6_beats(pd, j, tv[s], ti[s]);
#pragma unroll
                        for (int s = K6_KMAX - 1; s > 0; --s) {
                            tv[s] = bs[s - 1] ? tv[s - 1] : (bs[s] ? pd : tv[s]);
                            ti[s] = bs[s - 1] ? ti[s - 1] : (bs[s] ? j : ti[s]);
                        }
                        tv[0] = bs[0] ? pd : tv[0];
                        ti[0] = bs[0] ? j : ti[0];
                    }
                }
            }
            char* LB = sg ? LBB : LBA;
            int c = 0;
#pragma unroll
            for (int s = 0; s < K6_KMAX; ++s) {
                if (s < k && ti[s] != 0x7fffffff) {
                    const k6u32x2 e = {(unsigned)__float_as_int(tv[s]), (unsigned)ti[s]};
                    *(k6u32x2*)(LB + s * 8) = e;
                    c = s + 1;
                }
            }
            if (sg) cntB = c; else cntA = c;
        }
        exact_lists = true;
    }
    K6_T(6);
    // ---- F0: every lane moves ITS lists to the pd domain (pd' = 2 a - xx_q; the exact path stored pd itself) and fills them to the end with
    //      {-inf, 0} (never ahead of, never close to a real entry: the counting loops below read whole blocks unmasked)
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        char* LB = sg ? LBB : LBA;
        const int cnt = sg ? cntB : cntA;
        const float xxq = sg ? xcqB : xcqA;                    // (centred norm: the sweeps' coordinates)
        float mn = 0.f;                                        // min of -xx_j / 2 = -(largest squared norm among this list's survivors) / 2
#pragma unroll
        for (int p0 = 0; p0 < K6_LENT; p0 += 4) {
            k6u32x2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const k6u32x2*)(LB + (p0 + u) * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a = __int_as_float((int)v[u][0]);
                const float p = exact_lists ? a : fmaf(2.0f, a, -xxq);
                const bool live = p0 + u < cnt;
                const k6u32x2 w = {(unsigned)__float_as_int(live ? p : -INFINITY), live ? v[u][1] : 0u};
                *(k6u32x2*)(LB + (p0 + u) * 8) = w;
                if (live) mn = fminf(mn, nxx[v[u][1] & 4095u]);
            }
        }
        cnts[((sg ? gB : gA) * 32 + l31) * 4 + ch * 2 + h] = cnt;
        lmn[((sg ? gB : gA) * 32 + l31) * 4 + ch * 2 + h] = mn;
    }
    __syncthreads();
#if defined(K6_PROBE) && K6_PROBE == 3
    return;
#endif

    // ---------------------------------------------------------------------------------------------------------------- final: exact ranks
    // wave w finishes the 32 queries of group w, two per trip (one per half-wave); lane l31 owns entries l31 + 32 s of the query's
    // concatenated quarter lists.  F1: rank by counting over the keys (blocks of 8 entries per quarter read at once), keys scattered by
    // rank, neighbours in rank compared: a survivor within 2 E of the next / previous one is AMBIGUOUS (E from the largest norm among the
    // query's survivors).  The unambiguous are written out at once, the ambiguous go to the wave's work list.  F2 (once per wave, all
    // lanes busy): canonical distance of every listed pair (fmaf chain over the fp32 rows), exact keys back into the lists, recount.
    K6_T(7);
    // wave w finishes the 32 queries of group w, two per trip (one per half-wave); lane l31 owns entries l31 + 32 s of the query's
    // concatenated quarter lists.  F1, per entry of the query: rank += (pd'_e > mine), near += (|pd'_e - mine| <= 2 E) -- five cheap vector
    // instructions, blocks of 12 entries per quarter read at once; E from the largest norm among the query's survivors.  near > 1 (I count
    // myself): AMBIGUOUS -> the wave's work list; everyone else is written out at once (an unambiguous rank is final: see the header).
    // F2 (once per wave, all lanes busy): canonical distance of every listed pair (fmaf chain over the fp32 rows) back into the lists,
    // then a recount under the full order (value desc, index asc).
    const bool xvec = (ld & 3) == 0 && (((uintptr_t)x) & 15) == 0 && (C & 3) == 0;
    const int g = wave;                                        // the group this wave finishes
    unsigned* wl = wlbase + wave * K6_WL;                      // items: entry offset / 8 | query of the workgroup << 16
    int wcnt = 0;                                              // wave-uniform
    const float xq_all = xxb[chunk * 128 + g * 32 + l31];     // squared norms of the wave's 32 queries, one per lane (both halves): raw ...
    const float xcq_all = xcb[chunk * 128 + g * 32 + l31];    // ... and centred
    auto list_at = [&](int qlc_, int t) -> char* { return lists + (size_t)((((g * 2 + (t >> 1)) * 2 + (t & 1)) * 32 + qlc_) * K6_LSTR); };
    auto flush = [&]() {
#ifdef K6_STAMP
        const long long f0_ = (long long)__builtin_amdgcn_s_memtime();
        st_items += wcnt; st_flushes++;
#endif
        // F2a: exact distances
        for (int i0 = 0; i0 < wcnt; i0 += 64) {
            const int i = i0 + lane;
            if (i < wcnt) {
                const unsigned item = wl[i];
                const int qq_ = (int)(item >> 16);
                char* ent = lists + (size_t)(item & 0xffffu) * 8;
                const int j = *(const int*)(ent + 4);
                const int qrow_ = chunk * 128 + qq_;
                const float* rq = xb + (size_t)qrow_ * ld;
                const float* rj = xb + (size_t)j * ld;
                float acc = 0.f;
                if (xvec) {
                    for (int c = 0; c < C; c += 32) {          // C % 4 == 0; up to 32 channels of both rows in flight
                        f32x4 a4[8], b4[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (c + 4 * u < C) { a4[u] = *(const f32x4*)(rq + c + 4 * u); b4[u] = *(const f32x4*)(rj + c + 4 * u); }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (c + 4 * u < C) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc = fmaf(a4[u][e], b4[u][e], acc);
                            }
                    }
                } else {
                    for (int c = 0; c < C; ++c) acc = fmaf(rq[c], rj[c], acc);
                }
                const float t2 = fmaf(2.0f, acc, -xxb[j]);
                *(float*)ent = t2 - xxb[qrow_];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // F2b: recount against the query's lists (exact values where it matters, approximate ones elsewhere: safe, see the header)
        for (int i0 = 0; i0 < wcnt; i0 += 64) {
            const int i = i0 + lane;
            const bool on = i < wcnt;
            const unsigned item = on ? wl[i] : 0u;
            const int qq_ = (int)(item >> 16), qlc_ = qq_ & 31;
            const k6u32x2 me = *(const k6u32x2*)(lists + (size_t)(item & 0xffffu) * 8);
            const float pm = __int_as_float((int)me[0]);
            const int jm = (int)me[1];
            int rank = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const char* L = list_at(qlc_, t);
#pragma unroll
                for (int p0 = 0; p0 < K6_CAP; p0 += 8) {
                    k6u32x2 ke[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) ke[u] = *(const k6u32x2*)(L + (p0 + u) * 8);
#pragma unroll
                    for (int u = 0; u < 8; ++u) rank += k6_beats(__int_as_float((int)ke[u][0]), (int)ke[u][1], pm, jm) ? 1 : 0;
                }
            }
            if (on && rank < k) idx[((size_t)b * N + chunk * 128 + qq_) * k + rank] = jm;
        }
        wcnt = 0;
#ifdef K6_STAMP
        st_f2 += (long long)__builtin_amdgcn_s_memtime() - f0_;
#endif
    };

    for (int it = 0; it < 16; ++it) {
        const int qlc = it * 2 + h;                            // query of the group
        const int qq = g * 32 + qlc;                           // query of the workgroup
        const int qrow = chunk * 128 + qq;                     // query of the cloud
        const float xq0 = __builtin_amdgcn_readlane(xq_all, it * 2), xq1 = __builtin_amdgcn_readlane(xq_all, it * 2 + 1);
        const float xq = h ? xq1 : xq0;
        const float xc0 = __builtin_amdgcn_readlane(xcq_all, it * 2), xc1 = __builtin_amdgcn_readlane(xcq_all, it * 2 + 1);
        const float xcq = h ? xc1 : xc0;
        const char* L0 = list_at(qlc, 0);
        const char* L1 = list_at(qlc, 1);
        const char* L2 = list_at(qlc, 2);
        const char* L3 = list_at(qlc, 3);
        float pe[48];                                          // first 12 values of every quarter: issued before anything depends on the counts
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            pe[u] = *(const float*)(L0 + u * 8);
            pe[12 + u] = *(const float*)(L1 + u * 8);
            pe[24 + u] = *(const float*)(L2 + u * 8);
            pe[36 + u] = *(const float*)(L3 + u * 8);
        }
        const k6i32x4 c4 = *(const k6i32x4*)(cnts + qq * 4);
        const f32x4 m4 = *(const f32x4*)(lmn + qq * 4);
        const float xm = -2.0f * fminf(fminf(m4[0], m4[1]), fminf(m4[2], m4[3]));       // largest centred squared norm among the query's survivors
        const float E2 = exact_lists ? 0.0f : 2.0f * (K6_EPS * (xcq + xm) + K6_CANON * (xq + xxmax));     // (exact lists: only exact ties go through the full-order recount)
        const int p1 = c4[0], p2 = p1 + c4[1], p3 = p2 + c4[2], n = p3 + c4[3];
        const int nmax = max(__builtin_amdgcn_readlane(n, 0), __builtin_amdgcn_readlane(n, 32));
        int cmx = max(max(c4[0], c4[1]), max(c4[2], c4[3]));
        cmx = max(__builtin_amdgcn_readlane(cmx, 0), __builtin_amdgcn_readlane(cmx, 32));
        for (int s0 = 0; s0 < nmax; s0 += 32) {                // one trip unless a query has more than 32 survivors
            const int e = s0 + l31;
            const bool valid = e < n;
            const int t = (e >= p1) + (e >= p2) + (e >= p3);
            const int pos = e - (t == 0 ? 0 : t == 1 ? p1 : t == 2 ? p2 : p3);
            const char* mine = list_at(qlc, valid ? t : 0) + (valid ? pos : 0) * 8;
            const k6u32x2 me = *(const k6u32x2*)mine;
            const float pm = __int_as_float((int)me[0]);
            const int j = (int)me[1];
            int rank = 0, near = 0;
#pragma unroll
            for (int u = 0; u < 48; ++u) {
                rank += pe[u] > pm ? 1 : 0;
                near += fabsf(pe[u] - pm) <= E2 ? 1 : 0;
            }
            for (int p0 = 12; p0 < cmx; p0 += 4) {             // a quarter with more than 12 survivors (rare)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float a0 = *(const float*)(L0 + (p0 + u) * 8), a1 = *(const float*)(L1 + (p0 + u) * 8);
                    const float a2 = *(const float*)(L2 + (p0 + u) * 8), a3 = *(const float*)(L3 + (p0 + u) * 8);
                    rank += (a0 > pm ? 1 : 0) + (a1 > pm ? 1 : 0) + (a2 > pm ? 1 : 0) + (a3 > pm ? 1 : 0);
                    near += (fabsf(a0 - pm) <= E2 ? 1 : 0) + (fabsf(a1 - pm) <= E2 ? 1 : 0) + (fabsf(a2 - pm) <= E2 ? 1 : 0) + (fabsf(a3 - pm) <= E2 ? 1 : 0);
                }
            }
            bool amb = valid && !(near <= 1);                  // another survivor inside my 2 E window (I count once myself)
#if defined(K6_PROBE) && K6_PROBE == 4
            amb = false;
#endif
            if (valid && !amb && rank < k) idx[((size_t)b * N + qrow) * k + rank] = j;
            const unsigned long long m = __ballot(amb);
            if (m) {
                const int before = __builtin_popcountll(m & ((1ull << lane) - 1ull));
                if (amb) wl[wcnt + before] = (unsigned)((mine - lists) >> 3) | ((unsigned)qq << 16);
                wcnt += __builtin_popcountll(m);
            }
        }
        if (wcnt > K6_WL - 192 || (it == 15 && wcnt > 0)) {    // (a trip adds at most 2 x 96 items)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            flush();
        }
    }
#ifdef K6_STAMP
    K6_T(8);
    __syncthreads();
    if (lane == 0 && blockIdx.x < 64) {           // diagnostic build only (tools/knn6_stamps.py): stamps go behind the fragment image in the caller's workspace
        int* o = (int*)(const_cast<char*>(planes) + (size_t)B * N * CT * 4 + (size_t)B * N * 4) + (blockIdx.x * 4 + wave) * 16;
        for (int i = 1; i <= 8; ++i) o[i - 1] = (int)(stamp[i] - stamp[i - 1]);
        o[8] = (int)st_f2; o[9] = st_items; o[10] = st_flushes; o[11] = (int)(stamp[8] - stamp[0]);
    }
#endif
}

size_t knn6_lds_bytes(int N) { return (size_t)512 * K6_LSTR + (size_t)N * 4 + 128 * 4 + 512 * 4 + 64 + (size_t)4 * K6_WL * 4 + 512 * 4; }

// shapes v6 takes (the rest stays on knn.hip's kernels)
bool knn6_supported(int B, int N, int C, int k) {
    return B > 0 && N >= 128 && N % 128 == 0 && N <= 4096 && C >= 1 && C <= 128 && k >= 1 && k <= K6_KMAX && k <= N &&
           knn6_lds_bytes(N) <= 160 * 1024;
}
int knn6_padded_channels(int C) { return C <= 16 ? 16 : C <= 64 ? 64 : 128; }
size_t knn6_plane_bytes(int P, int C) { return (size_t)P * 2 * knn6_padded_channels(C) * sizeof(__bf16) + (size_t)P * sizeof(float); }    // image + centred norms

template <int CT>
static int knn6_go(hipStream_t st, const float* x, int ld, int B, int N, int C, int k, int* idx, float* xx, float* xc, char* planes) {
    const int P = B * N;
    hipLaunchKernelGGL((knn6_prep_kernel<CT>), dim3((P + 255) / 256), dim3(256), 0, st, x, ld, P, N, C, xx, xc, planes);
    const size_t lds = knn6_lds_bytes(N);
    hipError_t e = hipFuncSetAttribute((const void*)knn6_kernel<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((knn6_kernel<CT>), dim3((N / 128) * B), dim3(256), lds, st, x, ld, xx, xc, planes, N, C, k, idx, B);
    return mlsp_launch_status();
}

// xx [B*N] floats and planes (knn6_plane_bytes) are workspace; both are written here (xx = canonical squared norms, as sqnorm_kernel)
int launch_knn6(hipStream_t st, const float* x, int ld, int B, int N, int C, int k, int* idx, float* xx, void* planes) {
    if (!knn6_supported(B, N, C, k) || !planes || (((uintptr_t)planes) & 15)) return MLSP_ERR_UNSUPPORTED;
    const int CT = knn6_padded_channels(C);
    float* xc = (float*)((char*)planes + (size_t)B * N * CT * 4);            // centred norms behind the fragment image (knn6_plane_bytes covers them)
    if (CT == 16) return knn6_go<16>(st, x, ld, B, N, C, k, idx, xx, xc, (char*)planes);
    if (CT == 64) return knn6_go<64>(st, x, ld, B, N, C, k, idx, xx, xc, (char*)planes);
    return knn6_go<128>(st, x, ld, B, N, C, k, idx, xx, xc, (char*)planes);
}
