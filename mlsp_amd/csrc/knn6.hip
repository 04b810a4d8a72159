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
//     orders like the distance and pd' = 2 a - xx_q.  A wave carries ONE query group (32 queries) against half the candidates, two waves
//     per SIMD (two groups per wave against the same fragments was measured and rejected: see the kernel's comment).
//   * BOTH sweeps use the split-bf16 products hi hi + lo hi + hi lo.  Error budget (worst case, not typical): bf16 has 8 significant
//     bits, so |x - hi| <= 2^-8 |x|, |x - hi - lo| <= 2^-16 |x|; the dropped terms (lo lo, the two residual products) are at most
//     3 x 2^-16 |x_q,c x_j,c| per channel, i.e. |2 dot' - 2 dot| <= 3 x 2^-16 (xx_q + xx_j) = 2^-14.4 (xx_q + xx_j); the fp32 accumulation
//     inside and between the 3 C / 16 MFMAs adds at most 2^-16.1 (xx_q + xx_j) at C = 128; the canonical value's own two roundings
//     2^-22.  Sum < 2^-13.9 (xx_q + xx_j): E = 2^-13 (xx_q + X), X >= xx_j, bounds it with a factor 1.9 to spare.
//     Pass A: 64 running maxima per query (16 registers x 2 half-wave lanes x 2 candidate halves), tau = their k-th largest: at least k
//     candidates have pd' >= tau, so the true k-th best is >= tau - E and every true neighbour has pd' >= tau - 2 E (X = the cloud's
//     largest squared norm): pass B keeps exactly those (~1.4 k of N).
//   * Coordinates: the sweeps run on x - x_first(cloud) (distances are translation invariant, the split products' error scales with the
//     norms of what is multiplied); the canonical value's own rounding on the RAW coordinates is budgeted separately:
//     E = 2^-13 (xc_q + Xc) + (C + 4) 2^-23 (xx_q + Xx), xc = centred, xx = raw squared norms.
//   * Exactness: a survivor whose pd' is farther than 2 E from its neighbours in rank (Xc = the largest centred norm among the query's
//     survivors) has its rank decided by pd' alone.  Four lanes per query sort the query's <= 32 survivors in registers (bitonic network, the
//     exchanges at distance 8 / 16 through v_permlane16 / 32_swap),
//     flags the gaps within 2 E, the flagged (~10-40 %) get their canonical distance from an fmaf chain over the fp32 rows (pairs packed
//     densely over the wave), and odd-even passes under the full order (value desc, index asc) settle the flagged runs -- safe, because
//     an unflagged value differs from its neighbours by more than 2 E and an exact one from its approximation by at most E.  Queries
//     with more than 32 survivors (or exact lists) take a counting path with the same logic.
//   * Any list overflow (massive ties: more than 24 survivors in a quarter of a query's candidates) or NaN / inf bound sends the whole
//     workgroup through an exact path: f32 MFMA tiles from the fp32 rows, per-lane sorted top-k lists in registers, the same final.
// Measured on MI355X (B = 32, N = 1024, k = 20, one call incl. the prep kernel; uniform random clouds): C = 64 71 us (v5 113), C = 128 103 us
// (v5 176), N = 2048 C = 64 95 us (v5 176); C = 3 54 us (v5 49: the selection phases dominate when the sweeps are trivial, so C <= 16 stays
// on v5).  Inside the DGCNN step (features of a 3-D manifold: small gaps between neighbours relative to the norms, 2-3x more ambiguous
// pairs): 89 / 129 us against 118 / 165.  Phase cycles per wave at C = 64, two waves per SIMD (tools/knn6_stamps.py): sweeps 17 k + 27 k,
// tau 5.5 k, list conversion 5 k, sort / flag / settle 26 k, exact distances 16 k (bound by cache-line transactions: every 16-byte piece
// of a gathered row is its own transaction; a cooperative fetch through LDS lost to its chain of dependent steps as written here).
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
                                                        float* __restrict__ xc, char* __restrict__ planes, int* __restrict__ idx, int k,
                                                        int* __restrict__ cloud_flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    if (cloud_flag && i < P / N) cloud_flag[i] = 0;           // (knn6w_kernel: which clouds go to the v5 kernel)
    // Every output row starts as k zeros: a query with fewer than k comparable candidates (NaN coordinates -- its own or its cloud's) keeps
    // index 0 in the positions the main kernel does not write, as oracle/knn_canon.c does; the gathers downstream never see an
    // uninitialised index.
    for (int s = 0; s < k; ++s) idx[(size_t)i * k + s] = 0;
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

// VEX mode (C <= 3: the raw and the transformed cloud of DGCNN): no image, no centring -- the sweeps compute the CANONICAL distance itself
// with vector FMAs, so every survivor's value is exact and nothing has to be re-resolved.  Per point {x0, x1, x2, xx} (zero-padded
// coordinates: fmaf(0, 0, acc) == acc), xx = the canonical chain; the output rows start as zeros (see knn6_prep_kernel).
__global__ __launch_bounds__(256) void knn6_prep_vex_kernel(const float* __restrict__ x, int ld, int P, int C, float* __restrict__ xx,
                                                            f32x4* __restrict__ cand, int* __restrict__ idx, int k) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    for (int s = 0; s < k; ++s) idx[(size_t)i * k + s] = 0;
    const float* r = x + (size_t)i * ld;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    float acc = 0.f;
    for (int c = 0; c < C; ++c) { v[c] = r[c]; acc = fmaf(r[c], r[c], acc); }
    v[3] = acc;
    xx[i] = acc;
    cand[i] = v;
}

template <int I, int Nn, class F>
__device__ __forceinline__ void k6_static_for(F&& f) {
    if constexpr (I < Nn) { f(std::integral_constant<int, I>{}); k6_static_for<I + 1, Nn>(f); }
}

// candidate beats list entry (value desc, index asc); false for a NaN candidate
__device__ __forceinline__ bool k6_beats(float d, int j, float pv, int pi) { return d > pv || (d == pv && j < pi); }

// Workgroup = 8 waves = 128 queries x all N candidates of their cloud.  Wave w: qg = w & 3 picks a group of 32 queries, ch = w >> 2 the
// half of the candidates it sweeps: two waves per SIMD (the selection phases are vector-issue bound, and a lone wave issues at half rate),
// no barrier inside a sweep.  (Two query groups per wave against the same fragments -- half the fragment traffic -- was built and measured:
// the sweeps gain 10 %, every selection phase loses 2x at one wave per SIMD.)
// Measured (B = 32, N = 1024, k = 20): 37.7 us per call against 41.5 on knn_mfma5_kernel<4>; stamps (tools/knn6_stamps.py, clocks per
// workgroup): pass A 11.7 k, pass B 24.7 k, tau 5.5 k, final 11.4 k of 64 k -- the sweeps are LDS-issue bound, not vector bound: every
// ds_read_b128 of a candidate row serves 64 (query, candidate) pairs whatever the arithmetic (16 reads per tile and wave, 8 waves), and
// the masked survivor append doubles that in pass B.  Going below needs several queries per lane against one candidate read (another
// wave <-> query map, i.e. another list layout): not built.
// VEX (C <= 3, N <= 1024): `planes` is the {x0, x1, x2, xx} array of knn6_prep_vex_kernel; the cloud's 16 KB of it are staged in LDS (in the
// final's work-list area, dead until then) and the two sweeps are plain vector code: three FMAs, the two norm terms -- the canonical value,
// bit for bit -- per pair.  Lists hold exact distances (E = 0): the final flags only exact ties, which the full order settles by index.
template <int CT, bool VEX = false>
__global__ __launch_bounds__(512) void knn6_kernel(const float* __restrict__ x, int ld, const float* __restrict__ xx_all, const float* __restrict__ xc_all,
                                                   const char* __restrict__ planes, int N, int C, int k, int* __restrict__ idx, int B) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NKB = CT / 16;                              // 16-channel blocks = bf16 MFMA K steps per tile
    constexpr int PF = NKB >= 8 ? 1 : NKB >= 4 ? 2 : 4;      // tiles of fragments in flight (the ring holds PF * NKB blocks of 2 x 4 registers)
    constexpr int NR = PF * NKB;
    constexpr int UNR = NKB >= 8 ? 2 : 8;                     // tiles per loop trip (see `sweep`; 4 at C = 128 costs two spilled registers)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // (scalar: addresses below stay in SGPRs)
    const int l31 = lane & 31, h = lane >> 5, qg = wave & 3, ch = wave >> 2;
    int b, chunk;
    xcd_cloud_map(blockIdx.x, N / 128, B, b, chunk);
    const float* xxb = xx_all + (size_t)b * N;                // canonical squared norms (raw coordinates)
    const float* xcb = xc_all + (size_t)b * N;                // squared norms relative to the cloud's first point (the sweeps' coordinates)
    const size_t T0 = (size_t)b * (N / 32);                  // first 32-point tile of this cloud in the fragment-major image
    const float* xb = x + (size_t)b * N * ld;
    const int nt2 = N / 64;                                   // 32-candidate tiles of this wave's half

    char* lists = (char*)sm;                                  // [512 lists][K6_LSTR]; list = ((qg*2 + ch)*2 + h)*32 + query of the group
    float* xch = sm;                                          // tau exchange image [128 queries][K6_XS], dead before pass B
    float* nxx = (float*)(lists + 512 * K6_LSTR);             // [N]  -xc_j / 2
    float* tauv = nxx + N;                                    // [128]
    int* cnts = (int*)(tauv + 128);                           // [128 queries][4 quarters]
    float* red = (float*)(cnts + 512);                        // [16]
    unsigned* wlbase = (unsigned*)(red + 16);                 // [8 waves][512 words]: fast final: [16 queries][32] candidate / exact value of the ambiguous; slow final: work list
    float* lmn = (float*)(wlbase + 8 * 512);                  // [128 queries][4 quarters] min of -xc_j / 2 over the list's survivors
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
        // (fmaxf drops a NaN operand: a NaN norm -- a NaN coordinate, or an infinite first point of the cloud, the origin of the centred
        // image -- becomes an infinite bound here, and an infinite bound sends the workgroup through the exact path below)
        if (VEX) {
            const f32x4* cg = (const f32x4*)planes + (size_t)b * N;
            f32x4* cl = (f32x4*)wlbase;                       // the cloud {x, xx}: the final's work lists live here later
            for (int j = tid; j < N; j += 512) { const f32x4 v = cg[j]; cl[j] = v; mr = fmaxf(mr, v[3] == v[3] ? v[3] : INFINITY); }
        } else
        for (int j = tid; j < N; j += 512) {
            const float v = xcb[j], w = xxb[j];
            nxx[j] = -0.5f * v; m = fmaxf(m, v == v ? v : INFINITY); mr = fmaxf(mr, w == w ? w : INFINITY);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); mr = fmaxf(mr, __shfl_xor(mr, o, 64)); }
        if (lane == 0) { red[wave] = m; red[8 + wave] = mr; }
        __syncthreads();
        xcmax = red[0]; xxmax = red[8];
#pragma unroll
        for (int w = 1; w < 8; ++w) { xcmax = fmaxf(xcmax, red[w]); xxmax = fmaxf(xxmax, red[8 + w]); }
    }
    // |pd' - pd_canonical| <= K6_EPS (xc_q + xc_j)  [split products on the centred coordinates, header]
    //                        + K6_CANON (xx_q + xx_j)  [the canonical value's own distance from -|x_q - x_j|^2: an fmaf chain of C terms, the two
    //                          norms and two more roundings on the RAW coordinates: < (C + 4) 2^-23 (xx_q + xx_j)]
    const float K6_CANON = (float)(C + 4) * 1.1920929e-07f;
    const int q0 = chunk * 128 + qg * 32;                     // first query of this wave's group, local to the cloud
    const float xxq = xxb[q0 + l31];                          // raw (exact path, canonical bound)
    const float xcq = VEX ? 0.f : xcb[q0 + l31];              // centred (sweeps)
    // VEX: the lists hold the canonical values themselves -- no error to budget; a non-finite norm still asks for the exact path
    const float Eq = VEX ? (xxmax < INFINITY ? 0.f : INFINITY) : K6_EPS * (xcq + xcmax) + K6_CANON * (xxq + xxmax);

    // ---------------------------------------------------------------------------------------------------------------- approximate sweeps
    k6bf16x8 qh[NKB], ql[NKB];                                // B operand: query row l31 of the group, channels 16 kb + 8 h .. + 7
    f32x4 qv = {0.f, 0.f, 0.f, 0.f};                          // VEX: this lane's query {x0, x1, x2, xx}
    if constexpr (VEX) {
        qv = ((const f32x4*)wlbase)[q0 + l31];
    } else {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            qh[kb] = *(const k6bf16x8*)(planes + k6_piece(T0 + (q0 >> 5), NKB, kb, 0, h, l31));
            ql[kb] = *(const k6bf16x8*)(planes + k6_piece(T0 + (q0 >> 5), NKB, kb, 1, h, l31));
        }
    }
    const unsigned voff = (unsigned)(h * 512 + l31 * 16);     // this lane's 16 bytes inside a fragment KiB
    const char* cand0 = planes + (T0 + (size_t)ch * nt2) * (NKB * 2048);     // (uniform) first tile of this wave's half
    auto frag_load = [&](int tl, int kb, k6bf16x8& ah, k6bf16x8& al) {       // A operand: candidate row l31 of tile tl of this half
        const char* p = cand0 + (size_t)tl * (NKB * 2048) + kb * 2048;                       // uniform; one contiguous KiB per wave and piece
        ah = *(const k6bf16x8*)(p + voff);
        al = *(const k6bf16x8*)(p + 1024 + voff);
    };
    // one sweep over the half: sel(acc, tl) sees the finished 32 x 32 tile: acc[r] = a(query l31, candidate (r & 3) + 8 (r >> 2) + 4 h of the tile).
    // Fragment ring: slot (tile % PF, block) is refilled right after its use with the tile PF ahead; a scheduling barrier after every
    // refill keeps the loads where they are written (the scheduler otherwise sinks them next to their use: prefetch distance zero).
    // The compiler drains the memory counter once per loop trip (its wait-count analysis merges pessimistically at the back edge), so a
    // trip covers UNR tiles: the drain exposes one L2 latency per UNR tiles of MFMA work.
    auto sweep = [&](auto&& sel) {
        if constexpr (VEX) {
            // vector sweep: acc[r] = the CANONICAL distance of (query l31, candidate (r & 3) + 8 (r >> 2) + 4 h of the tile): dot = fmaf chain
            // over the channels from +0, t = fl(2 dot - xx_j), pd = fl(t - xx_i) (oracle/knn_canon.c).  The 32 lanes of a half read the same
            // candidate: one LDS broadcast per row.
            const f32x4* cl = (const f32x4*)wlbase;
            for (int tl = 0; tl < nt2; ++tl) {
                const f32x4* cp = cl + (ch * nt2 + tl) * 32 + 4 * h;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x4 c = cp[(r & 3) + 8 * (r >> 2)];
                    float d = fmaf(qv[0], c[0], 0.f);
                    d = fmaf(qv[1], c[1], d);
                    d = fmaf(qv[2], c[2], d);
                    acc[r] = fmaf(2.0f, d, -c[3]) - qv[3];
                }
                sel(acc, tl);
            }
            return;
        }
        k6bf16x8 fh[NR], fl[NR];
        auto tile = [&](auto SLOT, auto RING, int tl, int tn, auto&& sel_) {  // tn: the tile that refills this ring slot (always loaded: no branch)
            constexpr int s0 = decltype(SLOT)::value * NKB;
            constexpr bool ring = decltype(RING)::value;
            f32x16 acc;
            const float* p = nxx + (ch * nt2 + tl) * 32 + 4 * h;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 v = *(const f32x4*)(p + 8 * g4);
                acc[4 * g4] = v[0]; acc[4 * g4 + 1] = v[1]; acc[4 * g4 + 2] = v[2]; acc[4 * g4 + 3] = v[3];
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[s0 + kb], qh[kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[s0 + kb], qh[kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[s0 + kb], ql[kb], acc, 0, 0, 0);
                if (ring) {
                    frag_load(tn, kb, fh[s0 + kb], fl[s0 + kb]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            sel_(acc, tl);
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
    // ---- pass A: 16 running maxima per lane (acc domain: a = dot' - xc_j / 2 is monotone in the distance for a fixed query)
    {
        float cm[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) cm[r] = -INFINITY;
        sweep([&](const f32x16& acc, int) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cm[r] = fmaxf(cm[r], acc[r]);
        });
        K6_T(2);
        float* dst = xch + (qg * 32 + l31) * K6_XS + (ch * 2 + h) * 16;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) { const f32x4 v = {cm[4 * g4], cm[4 * g4 + 1], cm[4 * g4 + 2], cm[4 * g4 + 3]}; *(f32x4*)(dst + 4 * g4) = v; }
    }
    __syncthreads();
#if defined(K6_PROBE) && K6_PROBE == 1
    return;
#endif
    K6_T(3);
    // tau = k-th largest of a query's 64 maxima: one lane per query (the ch = 0 wave of a group: one wave per SIMD; lanes 32-63 mirror), bitonic
    // network in registers
    if (ch == 0) {
        const int qs = qg * 32 + l31;
        float v[64];
        const float* src = xch + qs * K6_XS;
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
        if (lane < 32) tauv[qs] = t;
    }
    __syncthreads();                                          // tau complete; the exchange image (aliases the lists) is dead from here on
#if defined(K6_PROBE) && K6_PROBE == 2
    return;
#endif
    K6_T(4);
    float thr = tauv[qg * 32 + l31] - Eq;                     // acc domain: pd' >= tau_pd - 2 E  <=>  a >= a_tau - E
    thr = thr == thr ? thr : -INFINITY;                       // (a non-finite bound Eq sends the workgroup through the exact path: `ovf` below)

    // ---- pass B: survivors -> this lane's private list (acc-domain value, candidate index)
    // per pair: v_cmpx (exec = survivors) / ds_write2_b32 {value, index} at the cursor / cursor += 8 under that mask / exec back to all lanes:
    // three vector instructions and no branch; the cursor is clamped every 4 appends (lists have 4 entries of slack).
    char* LB = lists + (size_t)((((qg * 2 + ch) * 2 + h) * 32 + l31) * K6_LSTR);
    const unsigned base = lds0 + (unsigned)((((qg * 2 + ch) * 2 + h) * 32 + l31) * K6_LSTR);
    unsigned ad = base, top = base;
    sweep([&](const f32x16& acc, int tl) {
        const int jb = (ch * nt2 + tl) * 32 + 4 * h;
        int jv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) jv[r] = jb + (r & 3) + 8 * (r >> 2);
        // The compiler does not place the MFMA -> VALU / LDS read wait states for instructions INSIDE an asm statement: a visible VALU read of
        // the accumulator (gate) comes first -- the hazard recognizer pads in front of it -- and every append depends on the gate.
        const float gate = fmaxf(acc[0], acc[15]);
#define K6_APPEND(ad_, val_, thr_, j_, gate_) asm volatile("v_cmpx_ge_f32_e32 vcc, %1, %2\n\tds_write2_b32 %0, %1, %3 offset1:1\n\tv_add_u32_e32 %0, 8, %0\n\ts_mov_b64 exec, -1" \
                                                           : "+v"(ad_) : "v"(val_), "v"(thr_), "v"(j_), "v"(gate_) : "vcc", "memory")
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            K6_APPEND(ad, acc[r], thr, jv[r], gate);
            if ((r & 3) == 3) { top = max(top, ad); ad = min(ad, base + K6_CAP * 8); }
        }
#undef K6_APPEND
    });
    int cnt = (int)(ad - base) >> 3;
    // overflow (massive ties), or no usable bound: a non-finite norm in the cloud (the approximate values of a NaN row are NaN and survive
    // no comparison, so they would not overflow anything: ask for the exact path outright)
    const int ovf = (top > base + K6_CAP * 8 || !(Eq < INFINITY)) ? 1 : 0;
    K6_T(5);
    bool exact_lists = false;
    if (__syncthreads_or(ovf)) {
        // ------------------------------------------------------------------------------------------------------------ exact path (rare)
        // f32 MFMA tiles straight from the fp32 rows (same transposed layout; the MFMA chain is the canonical fmaf chain, channels
        // ascending), every lane keeps the exact top-K6_KMAX of ITS quarter of the candidates in a sorted register list.
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
                for (int s_ = 0; s_ < 16; ++s_) f[s_] = (c0 + 2 * s_ + h < C) ? row[c0 + 2 * s_ + h] : 0.f;
            }
        };
        const float* qrowp = xb + (size_t)(q0 + l31) * ld;
        float tv[K6_KMAX];
        int ti[K6_KMAX];
#pragma unroll
        for (int s_ = 0; s_ < K6_KMAX; ++s_) { tv[s_] = -INFINITY; ti[s_] = 0x7fffffff; }
        for (int tl = 0; tl < nt2; ++tl) {
            const int j0 = (ch * nt2 + tl) * 32;
            const float* crowp = xb + (size_t)(j0 + l31) * ld;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            for (int c0 = 0; c0 < C; c0 += 32) {               // channels ascending: the canonical chain (query fragments re-read: rare path)
                float ca[16], qa[16];
                load_frag(crowp, c0, ca);
                load_frag(qrowp, c0, qa);
#pragma unroll
                for (int s_ = 0; s_ < 16; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[s_], qa[s_], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + 4 * h + (r & 3) + 8 * (r >> 2);
                const float xxj = xxb[j];                                    // canonical norm of the raw row
                const float pd = fmaf(2.0f, acc[r], -xxj) - xxq;
                if (k6_beats(pd, j, tv[K6_KMAX - 1], ti[K6_KMAX - 1])) {
                    bool bs[K6_KMAX];
#pragma unroll
                    for (int s_ = 0; s_ < K6_KMAX; ++s_) bs[s_] = k6_beats(pd, j, tv[s_], ti[s_]);
#pragma unroll
                    for (int s_ = K6_KMAX - 1; s_ > 0; --s_) {
                        tv[s_] = bs[s_ - 1] ? tv[s_ - 1] : (bs[s_] ? pd : tv[s_]);
                        ti[s_] = bs[s_ - 1] ? ti[s_ - 1] : (bs[s_] ? j : ti[s_]);
                    }
                    tv[0] = bs[0] ? pd : tv[0];
                    ti[0] = bs[0] ? j : ti[0];
                }
            }
        }
        cnt = 0;
#pragma unroll
        for (int s_ = 0; s_ < K6_KMAX; ++s_) {
            if (s_ < k && ti[s_] != 0x7fffffff) {
                const k6u32x2 e = {(unsigned)__float_as_int(tv[s_]), (unsigned)ti[s_]};
                *(k6u32x2*)(LB + s_ * 8) = e;
                cnt = s_ + 1;
            }
        }
        exact_lists = true;
    }
    K6_T(6);
    // ---- F0: every lane moves ITS list to the pd domain (pd' = 2 a - xc_q; the exact path stored pd itself) and fills it to the end with
    //      {-inf, 0} (never ahead of, never close to a real entry: the counting loops of the slow final read whole blocks unmasked)
    {
        float mn = 0.f;                                        // min of -xc_j / 2 = -(largest centred squared norm among this list's survivors) / 2
#pragma unroll
        for (int p0 = 0; p0 < K6_LENT; p0 += 4) {
            k6u32x2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const k6u32x2*)(LB + (p0 + u) * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a = __int_as_float((int)v[u][0]);
                const float p = (exact_lists || VEX) ? a : fmaf(2.0f, a, -xcq);
                const bool live = p0 + u < cnt;
                const k6u32x2 w = {(unsigned)__float_as_int(live ? p : -INFINITY), live ? v[u][1] : 0u};
                *(k6u32x2*)(LB + (p0 + u) * 8) = w;
                if (!VEX && live) mn = fminf(mn, nxx[v[u][1] & 4095u]);
            }
        }
        cnts[(qg * 32 + l31) * 4 + ch * 2 + h] = cnt;
        lmn[(qg * 32 + l31) * 4 + ch * 2 + h] = mn;
    }
    __syncthreads();
#if defined(K6_PROBE) && K6_PROBE == 3
    return;
#endif

    // ---------------------------------------------------------------------------------------------------------------- final: exact ranks
    K6_T(7);
    const bool xvec = (ld & 3) == 0 && (((uintptr_t)x) & 15) == 0 && (C & 3) == 0;
    auto list_at = [&](int qlc_, int t) -> char* { return lists + (size_t)((((qg * 2 + (t >> 1)) * 2 + (t & 1)) * 32 + qlc_) * K6_LSTR); };
#ifdef K6_STAMP
    auto put_stamps = [&]() {
        K6_T(8);
        __syncthreads();
        if (lane == 0 && blockIdx.x < 32) {       // diagnostic build only (tools/knn6_stamps.py): stamps go behind the fragment image in the caller's workspace
            int* o = (int*)(const_cast<char*>(planes) + (size_t)B * N * CT * 4 + (size_t)B * N * 4) + (blockIdx.x * 8 + wave) * 16;
            for (int i = 1; i <= 8; ++i) o[i - 1] = (int)(stamp[i] - stamp[i - 1]);
            o[8] = (int)st_f2; o[9] = st_items; o[10] = st_flushes; o[11] = (int)(stamp[8] - stamp[0]);
        }
    };
#endif
    // ---- fast final (every query of the group has at most 32 survivors -- the normal case): FOUR LANES PER QUERY.  Wave (qg, ch) finishes queries
    // ch * 16 .. + 15 of its group; lane l serves query l & 15 and holds rank positions 8 m .. 8 m + 7 of its 32 (m = l >> 4).  The four lanes pull the
    // query's survivors into registers and sort them by pd' with a bitonic network whose exchanges at distance 8 / 16 cross lanes (v_permlane16 /
    // 32_swap, no LDS): 72 compare-exchanges per lane instead of the 240 of one lane per query.  Then as before: the ones whose gap to a neighbour in
    // rank is within 2 E are flagged, get their canonical distances (all flagged pairs of the wave packed densely over its lanes: fmaf chains over
    // the fp32 rows), and the flagged runs are settled with odd-even passes under the full order (value desc, index asc); the pair across a lane
    // boundary goes through two shuffles.  An unflagged survivor never moves: its gaps exceed 2 E.
    {
        const int ql = lane & 15, m = lane >> 4;
        const int qlc = ch * 16 + ql;
        const int qq = qg * 32 + qlc, qrow = chunk * 128 + qq;
        const k6i32x4 c4 = *(const k6i32x4*)(cnts + qq * 4);
        const int p1 = c4[0], p2 = p1 + c4[1], p3 = p2 + c4[2], n = p3 + c4[3];
        int nmaxw = n;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nmaxw = max(nmaxw, __shfl_xor(nmaxw, o, 64));
        if (!exact_lists && nmaxw <= 32) {
            const f32x4 m4 = *(const f32x4*)(lmn + qq * 4);
            const float xm = -2.0f * fminf(fminf(m4[0], m4[1]), fminf(m4[2], m4[3]));
            const float xqr = xxb[qrow], xqc = VEX ? 0.f : xcb[qrow];
            const float E2 = VEX ? 0.0f : 2.0f * (K6_EPS * (xqc + xm) + K6_CANON * (xqr + xxmax));     // (VEX: exact values, only exact ties are flagged)
            const char* Lb0 = list_at(qlc, 0);
            const char* Lb1 = list_at(qlc, 1);
            const char* Lb2 = list_at(qlc, 2);
            const char* Lb3 = list_at(qlc, 3);
            float pv[8];
            int jv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = 8 * m + i;
                const int t = (e >= p1) + (e >= p2) + (e >= p3);
                const char* Lt = t == 0 ? Lb0 : t == 1 ? Lb1 : t == 2 ? Lb2 : Lb3;
                const int pos = e - (t == 0 ? 0 : t == 1 ? p1 : t == 2 ? p2 : p3);
                const k6u32x2 v = *(const k6u32x2*)(Lt + (e < n ? pos : 0) * 8);
                pv[i] = e < n ? __int_as_float((int)v[0]) : -INFINITY;
                jv[i] = e < n ? (int)v[1] : 0x7fffffff;
            }
            // value of the same register in lane ^ 16 / lane ^ 32
            auto x16 = [&](unsigned v) -> unsigned { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (m & 1) ? r[0] : r[1]; };
            auto x32 = [&](unsigned v) -> unsigned { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (m & 2) ? r[0] : r[1]; };
            auto cex_local = [&](int jj, bool desc) {          // positions a, a ^ jj inside the lane; desc: the larger value to the lower position
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const int c = a ^ jj;
                    if (c > a) {
                        const bool sw = desc ? pv[a] < pv[c] : pv[a] > pv[c];
                        const float ta = sw ? pv[c] : pv[a], tc = sw ? pv[a] : pv[c];
                        const int ja = sw ? jv[c] : jv[a], jc = sw ? jv[a] : jv[c];
                        pv[a] = ta; pv[c] = tc; jv[a] = ja; jv[c] = jc;
                    }
                }
            };
            auto cex_cross = [&](bool by32, bool desc) {       // positions 8 m + i and 8 (m ^ 1 | m ^ 2) + i: the lane with the lower m is the lower position
                const bool low = by32 ? (m & 2) == 0 : (m & 1) == 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float ov = __uint_as_float(by32 ? x32(__float_as_uint(pv[i])) : x16(__float_as_uint(pv[i])));
                    const int oj = (int)(by32 ? x32((unsigned)jv[i]) : x16((unsigned)jv[i]));
                    const bool sw = low ? (desc ? pv[i] < ov : pv[i] > ov) : (desc ? ov < pv[i] : ov > pv[i]);
                    pv[i] = sw ? ov : pv[i]; jv[i] = sw ? oj : jv[i];
                }
            };
            // bitonic network over positions a = 8 m + i; a pair keeps the larger value at the lower position iff (a & k2) == 0
#pragma unroll
            for (int a = 0; a < 8; a += 2) {                    // k2 = 2
                const bool desc = (a & 2) == 0;
                const bool sw = desc ? pv[a] < pv[a + 1] : pv[a] > pv[a + 1];
                const float ta = sw ? pv[a + 1] : pv[a], tc = sw ? pv[a] : pv[a + 1];
                const int ja = sw ? jv[a + 1] : jv[a], jc = sw ? jv[a] : jv[a + 1];
                pv[a] = ta; pv[a + 1] = tc; jv[a] = ja; jv[a + 1] = jc;
            }
            {                                                   // k2 = 4: jj = 2, 1; direction by bit 2 of the position
#pragma unroll
                for (int jj = 2; jj > 0; jj >>= 1)
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        const int c = a ^ jj;
                        if (c > a) {
                            const bool desc = (a & 4) == 0;
                            const bool sw = desc ? pv[a] < pv[c] : pv[a] > pv[c];
                            const float ta = sw ? pv[c] : pv[a], tc = sw ? pv[a] : pv[c];
                            const int ja = sw ? jv[c] : jv[a], jc = sw ? jv[a] : jv[c];
                            pv[a] = ta; pv[c] = tc; jv[a] = ja; jv[c] = jc;
                        }
                    }
            }
            { const bool d8 = (m & 1) == 0; cex_local(4, d8); cex_local(2, d8); cex_local(1, d8); }                      // k2 = 8
            { const bool d16 = (m & 2) == 0; cex_cross(false, d16); cex_local(4, d16); cex_local(2, d16); cex_local(1, d16); }   // k2 = 16
            { cex_cross(true, true); cex_cross(false, true); cex_local(4, true); cex_local(2, true); cex_local(1, true); }       // k2 = 32
            // flag: gap to the next survivor in rank not provably larger than 2 E (dead tail entries: -inf, never flagged)
            const float nxt0 = __shfl(pv[0], lane + 16, 64), prv7 = __shfl(pv[7], lane - 16, 64);
            unsigned amb = 0u;
#pragma unroll
            for (int i = 0; i < 7; ++i) amb |= (8 * m + i + 1 < n && !(pv[i] - pv[i + 1] > E2)) ? (3u << i) : 0u;
            amb |= (m < 3 && 8 * m + 8 < n && !(pv[7] - nxt0 > E2)) ? 0x80u : 0u;
            amb |= (m > 0 && 8 * m < n && !(prv7 - pv[0] > E2)) ? 1u : 0u;
#if defined(K6_PROBE) && K6_PROBE == 4
            amb = 0u;
#endif
            const int nfl = __builtin_popcount(amb);           // this lane's flagged entries; the query's: the four lanes' sum
            const int f0 = __shfl(nfl, ql, 64), f1 = __shfl(nfl, ql + 16, 64), f2 = __shfl(nfl, ql + 32, 64), f3 = __shfl(nfl, ql + 48, 64);
            const int nflag = f0 + f1 + f2 + f3;
            const int foff = m == 0 ? 0 : m == 1 ? f0 : m == 2 ? f0 + f1 : f0 + f1 + f2;
            unsigned* slots = wlbase + wave * 512 + ql * 32 + foff;      // this lane's part of the query's [32]: candidate index in, canonical distance out
            {
                int c = 0;
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    if ((amb >> a) & 1u) { slots[c] = (unsigned)jv[a]; ++c; }
                }
            }
            int fmaxw = nflag;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) fmaxw = max(fmaxw, __shfl_xor(fmaxw, o, 64));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#ifdef K6_STAMP
            const long long f0_ = (long long)__builtin_amdgcn_s_memtime();
            st_items += nfl; st_flushes = fmaxw;
#endif
            // Canonical distances of the flagged pairs, DENSELY packed over the wave's lanes: pair i of the wave = (query, slot) by a prefix sum of
            // the 16 queries' flag counts (the counts differ a lot from query to query: slot-by-slot rounds ran at a third of the lanes).  Both rows
            // fetched whole (2 x 16 x 16 bytes in flight per lane at C = 64), fmaf chain channels ascending.  (Staging the rows through LDS with
            // cooperative, fully coalesced fetches was built and measured: 2x slower -- the phase is bound by its chain of dependent steps, not by
            // cache-line transactions.)
            if (fmaxw > 0) {
                int offs[17];
                offs[0] = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) offs[q + 1] = offs[q] + __builtin_amdgcn_readlane(nflag, q);      // (scalar)
                const int total = offs[16];
                unsigned* wslots = wlbase + wave * 512;
                const int qbase = chunk * 128 + qg * 32 + ch * 16;
                for (int i0 = 0; i0 < total; i0 += 64) {
                    const int i = i0 + lane;
                    if (i < total) {
                        int q = 0, ob = 0;
#pragma unroll
                        for (int t = 1; t < 16; ++t) { const bool ge = i >= offs[t]; q = ge ? t : q; ob = ge ? offs[t] : ob; }
                        unsigned* sp = wslots + q * 32 + (i - ob);
                        const int j = (int)*sp;
                        const int qr = qbase + q;
                        const float* rq = xb + (size_t)qr * ld;
                        const float* rj = xb + (size_t)j * ld;
                        float acc = 0.f;
                        if (!xvec) {
                            for (int c = 0; c < C; ++c) acc = fmaf(rq[c], rj[c], acc);
                        } else {
                            constexpr int CHV = CT >= 128 ? 8 : 16;        // float4 pieces of both rows in flight (register budget at C = 128)
                            for (int c = 0; c < C; c += 4 * CHV) {
                                f32x4 a4[CHV], b4[CHV];
#pragma unroll
                                for (int u = 0; u < CHV; ++u)
                                    if (c + 4 * u < C) { a4[u] = *(const f32x4*)(rq + c + 4 * u); b4[u] = *(const f32x4*)(rj + c + 4 * u); }
#pragma unroll
                                for (int u = 0; u < CHV; ++u)
                                    if (c + 4 * u < C) {
#pragma unroll
                                        for (int e = 0; e < 4; ++e) acc = fmaf(a4[u][e], b4[u][e], acc);
                                    }
                            }
                        }
                        const float t2 = fmaf(2.0f, acc, -xxb[j]);
                        *sp = (unsigned)__float_as_int(t2 - xxb[qr]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
#ifdef K6_STAMP
            st_f2 += (long long)__builtin_amdgcn_s_memtime() - f0_;
#endif
            {
                int c = 0;
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    if ((amb >> a) & 1u) { pv[a] = __int_as_float((int)slots[c]); ++c; }
                }
            }
            // settle the flagged runs: odd-even transposition passes under the full order until nothing moves (a run of r entries needs <= r passes)
            if (fmaxw > 0) {
                for (int pass = 0; pass < 32; ++pass) {
                    bool moved = false;
#pragma unroll
                    for (int a = 0; a < 7; a += 2) {             // positions (8 m + a, + 1), a even: inside the lane
                        const bool sw = k6_beats(pv[a + 1], jv[a + 1], pv[a], jv[a]);
                        const float ta = sw ? pv[a + 1] : pv[a], tc = sw ? pv[a] : pv[a + 1];
                        const int ja = sw ? jv[a + 1] : jv[a], jc = sw ? jv[a] : jv[a + 1];
                        pv[a] = ta; pv[a + 1] = tc; jv[a] = ja; jv[a + 1] = jc;
                        moved |= sw;
                    }
#pragma unroll
                    for (int a = 1; a < 7; a += 2) {
                        const bool sw = k6_beats(pv[a + 1], jv[a + 1], pv[a], jv[a]);
                        const float ta = sw ? pv[a + 1] : pv[a], tc = sw ? pv[a] : pv[a + 1];
                        const int ja = sw ? jv[a + 1] : jv[a], jc = sw ? jv[a] : jv[a + 1];
                        pv[a] = ta; pv[a + 1] = tc; jv[a] = ja; jv[a + 1] = jc;
                        moved |= sw;
                    }
                    {                                            // the odd pair across the lane boundary: (8 m + 7, 8 (m + 1))
                        const float nv = __shfl(pv[0], lane + 16, 64), pvv = __shfl(pv[7], lane - 16, 64);
                        const int nj = __shfl(jv[0], lane + 16, 64), pj = __shfl(jv[7], lane - 16, 64);
                        const bool swh = m < 3 && k6_beats(nv, nj, pv[7], jv[7]);       // the next lane's first entry beats my last one
                        const bool swl = m > 0 && k6_beats(pv[0], jv[0], pvv, pj);      // my first entry beats the previous lane's last one
                        pv[7] = swh ? nv : pv[7]; jv[7] = swh ? nj : jv[7];
                        pv[0] = swl ? pvv : pv[0]; jv[0] = swl ? pj : jv[0];
                        moved |= swh | swl;
                    }
                    if (!__any(moved)) break;
                }
            }
            {
                int* out = idx + ((size_t)b * N + qrow) * k;
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const int pos = 8 * m + a;
                    if (pos < k && pos < n) out[pos] = jv[a];
                }
            }
#ifdef K6_STAMP
            put_stamps();
#endif
            return;
        }
    }
    // ---- slow final (a query with more than 32 survivors, or exact lists): rank by counting, two queries per trip (one per half-wave); lane l31
    // owns entries l31 + 32 s of the query's concatenated quarter lists.  Per entry of the query: rank += (pd'_e > mine), near += (|pd'_e - mine|
    // <= 2 E).  near > 1 (I count myself): AMBIGUOUS -> the wave's work list; everyone else is written out at once.  Flush: canonical distance
    // of every listed pair back into the lists, then a recount under the full order (value desc, index asc).
    unsigned* wl = wlbase + wave * 512;                        // items: entry offset / 8 | query of the workgroup << 16
    int wcnt = 0;                                              // wave-uniform
    auto flush = [&]() {
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
    };
    for (int it = 0; it < 8; ++it) {
        const int qlc = ch * 16 + it * 2 + h;                  // query of the group
        const int qq = qg * 32 + qlc;                          // query of the workgroup
        const int qrow = chunk * 128 + qq;                     // query of the cloud
        const float xq = xxb[qrow], xqc = VEX ? 0.f : xcb[qrow];
        const char* L0 = list_at(qlc, 0);
        const char* L1 = list_at(qlc, 1);
        const char* L2 = list_at(qlc, 2);
        const char* L3 = list_at(qlc, 3);
        const k6i32x4 c4 = *(const k6i32x4*)(cnts + qq * 4);
        const f32x4 m4 = *(const f32x4*)(lmn + qq * 4);
        const float xm = -2.0f * fminf(fminf(m4[0], m4[1]), fminf(m4[2], m4[3]));       // largest centred squared norm among the query's survivors
        const float E2 = (exact_lists || VEX) ? 0.0f : 2.0f * (K6_EPS * (xqc + xm) + K6_CANON * (xq + xxmax));     // (exact values: only exact ties go through the full-order recount)
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
            for (int p0 = 0; p0 < cmx; p0 += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float a0 = *(const float*)(L0 + (p0 + u) * 8), a1 = *(const float*)(L1 + (p0 + u) * 8);
                    const float a2 = *(const float*)(L2 + (p0 + u) * 8), a3 = *(const float*)(L3 + (p0 + u) * 8);
                    rank += (a0 > pm ? 1 : 0) + (a1 > pm ? 1 : 0) + (a2 > pm ? 1 : 0) + (a3 > pm ? 1 : 0);
                    near += (fabsf(a0 - pm) <= E2 ? 1 : 0) + (fabsf(a1 - pm) <= E2 ? 1 : 0) + (fabsf(a2 - pm) <= E2 ? 1 : 0) + (fabsf(a3 - pm) <= E2 ? 1 : 0);
                }
            }
            const bool amb = valid && !(near <= 1);            // another survivor inside my 2 E window (I count once myself)
            if (valid && !amb && rank < k) idx[((size_t)b * N + qrow) * k + rank] = j;
            const unsigned long long m = __ballot(amb);
            if (m) {
                const int before = __builtin_popcountll(m & ((1ull << lane) - 1ull));
                if (amb) wl[wcnt + before] = (unsigned)((mine - lists) >> 3) | ((unsigned)qq << 16);
                wcnt += __builtin_popcountll(m);
            }
        }
        if (wcnt > 512 - 192 || (it == 7 && wcnt > 0)) {       // (a trip adds at most 2 x 96 items)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            flush();
        }
    }
#ifdef K6_STAMP
    put_stamps();
#endif
}


// ------------------------------------------------------------------------------------------------------------------------------------
// v6 WIDE (24 < k <= 40, N % 128 == 0, C <= 128): PointSegDA's graphs (k = 40, PointSegDA/Models.py:6-15; BASELINE.json configs[4]).
// The same two sweeps, lists and final as knn6_kernel with the WORKGROUP re-cut so that the 232-byte survivor lists still fit the CU:
//   * workgroup = 64 queries (two groups of 32) x all N candidates; wave w: qg = w & 1, ch = w >> 1 sweeps a QUARTER of the candidates.
//     A query's candidates are cut into 8 sub-ranges (quarter x half-wave lane): 512 lists of K6_CAP = 24 entries as before, but a query
//     now owns 8 of them.  With k = 40 a query keeps ~50 survivors, ~6 per list: K6_CAP is six standard deviations away.
//   * pass A leaves 128 running maxima per query; tau = their k-th largest.  Two lanes per query sort 64 each (the bitonic network of
//     knn6_kernel), exchange them through v_permlane32_swap (max(a[i], b[63 - i]) = the 64 largest of the union, a bitonic sequence) and
//     merge.  Against 64 maxima the bound is tighter: ~1.2 k survivors instead of ~1.6 k.
//   * fast final: EIGHT lanes per query (64 rank positions; exchanges at lane distance 8 through DPP row_ror, 16 / 32 through the
//     permlane swaps); the counting final takes queries with more than 64 survivors.
//   * no exact path inside: a workgroup whose lists overflow (massive ties: padded clouds) or whose bound is not finite raises its CLOUD's
//     flag and leaves; the v5 kernel (knn.hip, KB = 1) runs right behind on the flagged clouds only (its workgroups of other clouds return at
//     once) and overwrites their rows.  Indices bit-exact either way: both kernels rank by the canonical distance under the same total order.
#define K6W_XS 132          // floats per query of the tau exchange image (128 maxima, 16-byte aligned rows, 4-bank skew)
#define K6W_KMAX 40

__device__ __forceinline__ unsigned k6w_x8(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true); }     // lane ^ 8 (row_ror:8)

template <int CT>
__global__ __launch_bounds__(512) void knn6w_kernel(const float* __restrict__ x, int ld, const float* __restrict__ xx_all, const float* __restrict__ xc_all,
                                                    const char* __restrict__ planes, int N, int C, int k, int* __restrict__ idx, int B,
                                                    int* __restrict__ cloud_flag) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NKB = CT / 16;
    constexpr int PF = NKB >= 8 ? 1 : NKB >= 4 ? 2 : 4;
    constexpr int NR = PF * NKB;
    constexpr int UNR = NKB >= 8 ? 2 : 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5, qg = wave & 1, ch = wave >> 1;
    int b, chunk;
    xcd_cloud_map(blockIdx.x, N / 64, B, b, chunk);
    const float* xxb = xx_all + (size_t)b * N;
    const float* xcb = xc_all + (size_t)b * N;
    const size_t T0 = (size_t)b * (N / 32);
    const float* xb = x + (size_t)b * N * ld;
    const int nt4 = N / 128;                                  // 32-candidate tiles of this wave's quarter

    char* lists = (char*)sm;                                  // [512 lists][K6_LSTR]; list = ((qg*4 + ch)*2 + h)*32 + query of the group = (qg*8 + t)*32 + query, t = 2 ch + h
    float* xch = sm;                                          // tau exchange image [64 queries][K6W_XS], dead before pass B
    float* nxx = (float*)(lists + 512 * K6_LSTR);             // [N]  -xc_j / 2
    float* tauv = nxx + N;                                    // [64] (+64 unused)
    int* cnts = (int*)(tauv + 128);                           // [64 queries][8 sub-ranges]
    float* red = (float*)(cnts + 512);                        // [16]
    unsigned* wlbase = (unsigned*)(red + 16);                 // [8 waves][512 words]: fast final [8 queries][64]; counting final: work list
    float* lmn = (float*)(wlbase + 8 * 512);                  // [64 queries][8 sub-ranges]
    const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)sm);

    float xcmax, xxmax;
    {
        float m = 0.f, mr = 0.f;
        for (int j = tid; j < N; j += 512) {
            const float v = xcb[j], w = xxb[j];
            nxx[j] = -0.5f * v; m = fmaxf(m, v == v ? v : INFINITY); mr = fmaxf(mr, w == w ? w : INFINITY);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); mr = fmaxf(mr, __shfl_xor(mr, o, 64)); }
        if (lane == 0) { red[wave] = m; red[8 + wave] = mr; }
        __syncthreads();
        xcmax = red[0]; xxmax = red[8];
#pragma unroll
        for (int w = 1; w < 8; ++w) { xcmax = fmaxf(xcmax, red[w]); xxmax = fmaxf(xxmax, red[8 + w]); }
    }
    const float K6_CANON = (float)(C + 4) * 1.1920929e-07f;
    const int q0 = chunk * 64 + qg * 32;                      // first query of this wave's group, local to the cloud
    const float xxq = xxb[q0 + l31];
    const float xcq = xcb[q0 + l31];
    const float Eq = K6_EPS * (xcq + xcmax) + K6_CANON * (xxq + xxmax);      // (error budget: knn6_kernel / the file header)

    k6bf16x8 qh[NKB], ql[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        qh[kb] = *(const k6bf16x8*)(planes + k6_piece(T0 + (q0 >> 5), NKB, kb, 0, h, l31));
        ql[kb] = *(const k6bf16x8*)(planes + k6_piece(T0 + (q0 >> 5), NKB, kb, 1, h, l31));
    }
    const unsigned voff = (unsigned)(h * 512 + l31 * 16);
    const char* cand0 = planes + (T0 + (size_t)ch * nt4) * (NKB * 2048);
    auto frag_load = [&](int tl, int kb, k6bf16x8& ah, k6bf16x8& al) {
        const char* p = cand0 + (size_t)tl * (NKB * 2048) + kb * 2048;
        ah = *(const k6bf16x8*)(p + voff);
        al = *(const k6bf16x8*)(p + 1024 + voff);
    };
    // (the sweep of knn6_kernel over nt4 tiles: fragment ring, a scheduling barrier after every refill, UNR tiles per trip)
    auto sweep = [&](auto&& sel) {
        k6bf16x8 fh[NR], fl[NR];
        auto tile = [&](auto SLOT, auto RING, int tl, int tn, auto&& sel_) {
            constexpr int s0 = decltype(SLOT)::value * NKB;
            constexpr bool ring = decltype(RING)::value;
            f32x16 acc;
            const float* p = nxx + (ch * nt4 + tl) * 32 + 4 * h;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 v = *(const f32x4*)(p + 8 * g4);
                acc[4 * g4] = v[0]; acc[4 * g4 + 1] = v[1]; acc[4 * g4 + 2] = v[2]; acc[4 * g4 + 3] = v[3];
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[s0 + kb], qh[kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[s0 + kb], qh[kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[s0 + kb], ql[kb], acc, 0, 0, 0);
                if (ring) {
                    frag_load(tn, kb, fh[s0 + kb], fl[s0 + kb]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            sel_(acc, tl);
        };
        const int nmain = (nt4 / UNR) * UNR;
        if (nmain > 0) {
#pragma unroll
            for (int tu = 0; tu < PF; ++tu)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) frag_load(min(tu, nt4 - 1), kb, fh[tu * NKB + kb], fl[tu * NKB + kb]);
            for (int t0 = 0; t0 < nmain; t0 += UNR) {
                k6_static_for<0, UNR>([&](auto TU) {
                    constexpr int tu = decltype(TU)::value;
                    const int tl = t0 + tu;
                    const int tn = min(tl + PF, nt4 - 1);
                    tile(std::integral_constant<int, tu % PF>{}, std::true_type{}, tl, tn, sel);
                });
            }
        }
        for (int tl = nmain; tl < nt4; ++tl) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) frag_load(tl, kb, fh[kb], fl[kb]);
            tile(std::integral_constant<int, 0>{}, std::false_type{}, tl, tl, sel);
        }
    };

    // ---- pass A: 16 running maxima per lane -> 128 per query
    {
        float cm[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) cm[r] = -INFINITY;
        sweep([&](const f32x16& acc, int) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cm[r] = fmaxf(cm[r], acc[r]);
        });
        float* dst = xch + (qg * 32 + l31) * K6W_XS + (ch * 2 + h) * 16;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) { const f32x4 v = {cm[4 * g4], cm[4 * g4 + 1], cm[4 * g4 + 2], cm[4 * g4 + 3]}; *(f32x4*)(dst + 4 * g4) = v; }
    }
    __syncthreads();
    // tau = k-th largest of the query's 128 maxima: lanes (l31, h = 0 / 1) of the group's ch = 0 wave each sort 64, then merge
    if (ch == 0) {
        const int qs = qg * 32 + l31;
        float v[64];
        const float* src = xch + qs * K6W_XS + h * 64;
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
        // the 64 largest of the two sorted halves: w[i] = max(mine[i], other[63 - i]) (a bitonic sequence; the partner lane holds it reversed)
        float w[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[63 - i]), __float_as_uint(v[63 - i]), false, false);
            w[i] = fmaxf(v[i], __uint_as_float(h ? r[0] : r[1]));
        }
#pragma unroll
        for (int j = 32; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const int l = i ^ j;
                if (l > i) { const float lo = fminf(w[i], w[l]), hi = fmaxf(w[i], w[l]); w[i] = hi; w[l] = lo; }
            }
        float t = w[0];
#pragma unroll
        for (int i = 1; i < K6W_KMAX; ++i) t = (i == k - 1) ? w[i] : t;
        if (lane < 32) tauv[qs] = t;
    }
    __syncthreads();                                          // tau complete; the exchange image (aliases the lists) is dead from here on
    float thr = tauv[qg * 32 + l31] - Eq;
    thr = thr == thr ? thr : -INFINITY;

    // ---- pass B: survivors -> this lane's private list
    const int lid = ((qg * 4 + ch) * 2 + h) * 32 + l31;
    char* LB = lists + (size_t)lid * K6_LSTR;
    const unsigned base = lds0 + (unsigned)(lid * K6_LSTR);
    unsigned ad = base, top = base;
    sweep([&](const f32x16& acc, int tl) {
        const int jb = (ch * nt4 + tl) * 32 + 4 * h;
        int jv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) jv[r] = jb + (r & 3) + 8 * (r >> 2);
        const float gate = fmaxf(acc[0], acc[15]);            // (MFMA -> VALU wait states: see knn6_kernel)
#define K6_APPEND(ad_, val_, thr_, j_, gate_) asm volatile("v_cmpx_ge_f32_e32 vcc, %1, %2\n\tds_write2_b32 %0, %1, %3 offset1:1\n\tv_add_u32_e32 %0, 8, %0\n\ts_mov_b64 exec, -1" \
                                                           : "+v"(ad_) : "v"(val_), "v"(thr_), "v"(j_), "v"(gate_) : "vcc", "memory")
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            K6_APPEND(ad, acc[r], thr, jv[r], gate);
            if ((r & 3) == 3) { top = max(top, ad); ad = min(ad, base + K6_CAP * 8); }
        }
#undef K6_APPEND
    });
    const int cnt = (int)(ad - base) >> 3;
    const int ovf = (top > base + K6_CAP * 8 || !(Eq < INFINITY)) ? 1 : 0;
    if (__syncthreads_or(ovf)) {                              // overflow / no usable bound: this cloud goes to the v5 kernel launched behind
        if (tid == 0) cloud_flag[b] = 1;
        return;
    }
    // ---- F0: lists to the pd domain (pd' = 2 a - xc_q), filled to the end with {-inf, 0}
    {
        float mn = 0.f;
#pragma unroll
        for (int p0 = 0; p0 < K6_LENT; p0 += 4) {
            k6u32x2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const k6u32x2*)(LB + (p0 + u) * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a = __int_as_float((int)v[u][0]);
                const float p = fmaf(2.0f, a, -xcq);
                const bool live = p0 + u < cnt;
                const k6u32x2 w = {(unsigned)__float_as_int(live ? p : -INFINITY), live ? v[u][1] : 0u};
                *(k6u32x2*)(LB + (p0 + u) * 8) = w;
                if (live) mn = fminf(mn, nxx[v[u][1] & 4095u]);
            }
        }
        cnts[(qg * 32 + l31) * 8 + ch * 2 + h] = cnt;
        lmn[(qg * 32 + l31) * 8 + ch * 2 + h] = mn;
    }
    __syncthreads();

    // ---------------------------------------------------------------------------------------------------------------- final: exact ranks
    const bool xvec = (ld & 3) == 0 && (((uintptr_t)x) & 15) == 0 && (C & 3) == 0;
    auto list_at = [&](int qlc_, int t) -> char* { return lists + (size_t)(((qg * 8 + t) * 32 + qlc_) * K6_LSTR); };
    // ---- fast final (every query of the wave has at most 64 survivors): eight lanes per query, lane l serves query l & 7 of the wave's eight
    // (queries ch * 8 .. + 7 of the group) and holds rank positions 8 m .. 8 m + 7 of its 64 (m = l >> 3)
    {
        const int ql8 = lane & 7, m = lane >> 3;
        const int qlc = ch * 8 + ql8;
        const int qq = qg * 32 + qlc, qrow = chunk * 64 + qq;
        const k6i32x4 ca = *(const k6i32x4*)(cnts + qq * 8), cb = *(const k6i32x4*)(cnts + qq * 8 + 4);
        int pp[9];
        pp[0] = 0; pp[1] = ca[0]; pp[2] = pp[1] + ca[1]; pp[3] = pp[2] + ca[2]; pp[4] = pp[3] + ca[3];
        pp[5] = pp[4] + cb[0]; pp[6] = pp[5] + cb[1]; pp[7] = pp[6] + cb[2]; pp[8] = pp[7] + cb[3];
        const int n = pp[8];
        int nmaxw = n;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nmaxw = max(nmaxw, __shfl_xor(nmaxw, o, 64));
        if (nmaxw <= 64) {
            const f32x4 ma = *(const f32x4*)(lmn + qq * 8), mb = *(const f32x4*)(lmn + qq * 8 + 4);
            const float xm = -2.0f * fminf(fminf(fminf(ma[0], ma[1]), fminf(ma[2], ma[3])), fminf(fminf(mb[0], mb[1]), fminf(mb[2], mb[3])));
            const float xqr = xxb[qrow], xqc = xcb[qrow];
            const float E2 = 2.0f * (K6_EPS * (xqc + xm) + K6_CANON * (xqr + xxmax));
            float pv[8];
            int jv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = 8 * m + i;
                int t = 0, pb = 0;
#pragma unroll
                for (int u = 1; u < 8; ++u) { const bool ge = e >= pp[u]; t = ge ? u : t; pb = ge ? pp[u] : pb; }
                const k6u32x2 v = *(const k6u32x2*)(list_at(qlc, e < n ? t : 0) + (e < n ? e - pb : 0) * 8);
                pv[i] = e < n ? __int_as_float((int)v[0]) : -INFINITY;
                jv[i] = e < n ? (int)v[1] : 0x7fffffff;
            }
            auto x16 = [&](unsigned v) -> unsigned { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (m & 2) ? r[0] : r[1]; };
            auto x32 = [&](unsigned v) -> unsigned { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (m & 4) ? r[0] : r[1]; };
            auto cex_local = [&](int jj, bool desc) {
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const int c = a ^ jj;
                    if (c > a) {
                        const bool sw = desc ? pv[a] < pv[c] : pv[a] > pv[c];
                        const float ta = sw ? pv[c] : pv[a], tc = sw ? pv[a] : pv[c];
                        const int ja = sw ? jv[c] : jv[a], jc = sw ? jv[a] : jv[c];
                        pv[a] = ta; pv[c] = tc; jv[a] = ja; jv[c] = jc;
                    }
                }
            };
            auto cex_cross = [&](int bit, bool desc) {       // positions 8 m + i and 8 (m ^ bit) + i: the lane whose m lacks `bit` is the lower position
                const bool low = (m & bit) == 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned uv = __float_as_uint(pv[i]), uj = (unsigned)jv[i];
                    const float ov = __uint_as_float(bit == 1 ? k6w_x8(uv) : bit == 2 ? x16(uv) : x32(uv));
                    const int oj = (int)(bit == 1 ? k6w_x8(uj) : bit == 2 ? x16(uj) : x32(uj));
                    const bool sw = low ? (desc ? pv[i] < ov : pv[i] > ov) : (desc ? ov < pv[i] : ov > pv[i]);
                    pv[i] = sw ? ov : pv[i]; jv[i] = sw ? oj : jv[i];
                }
            };
            // bitonic network over positions a = 8 m + i; a pair keeps the larger value at the lower position iff (a & k2) == 0
#pragma unroll
            for (int a = 0; a < 8; a += 2) {                    // k2 = 2
                const bool desc = (a & 2) == 0;
                const bool sw = desc ? pv[a] < pv[a + 1] : pv[a] > pv[a + 1];
                const float ta = sw ? pv[a + 1] : pv[a], tc = sw ? pv[a] : pv[a + 1];
                const int ja = sw ? jv[a + 1] : jv[a], jc = sw ? jv[a] : jv[a + 1];
                pv[a] = ta; pv[a + 1] = tc; jv[a] = ja; jv[a + 1] = jc;
            }
#pragma unroll
            for (int jj = 2; jj > 0; jj >>= 1)                  // k2 = 4: direction by bit 2 of the position
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const int c = a ^ jj;
                    if (c > a) {
                        const bool desc = (a & 4) == 0;
                        const bool sw = desc ? pv[a] < pv[c] : pv[a] > pv[c];
                        const float ta = sw ? pv[c] : pv[a], tc = sw ? pv[a] : pv[c];
                        const int ja = sw ? jv[c] : jv[a], jc = sw ? jv[a] : jv[c];
                        pv[a] = ta; pv[c] = tc; jv[a] = ja; jv[c] = jc;
                    }
                }
            { const bool d = (m & 1) == 0; cex_local(4, d); cex_local(2, d); cex_local(1, d); }                                              // k2 = 8
            { const bool d = (m & 2) == 0; cex_cross(1, d); cex_local(4, d); cex_local(2, d); cex_local(1, d); }                             // k2 = 16
            { const bool d = (m & 4) == 0; cex_cross(2, d); cex_cross(1, d); cex_local(4, d); cex_local(2, d); cex_local(1, d); }            // k2 = 32
            { cex_cross(4, true); cex_cross(2, true); cex_cross(1, true); cex_local(4, true); cex_local(2, true); cex_local(1, true); }      // k2 = 64
            // flag: gap to the next survivor in rank not provably larger than 2 E
            const float nxt0 = __shfl(pv[0], lane + 8, 64), prv7 = __shfl(pv[7], lane - 8, 64);
            unsigned amb = 0u;
#pragma unroll
            for (int i = 0; i < 7; ++i) amb |= (8 * m + i + 1 < n && !(pv[i] - pv[i + 1] > E2)) ? (3u << i) : 0u;
            amb |= (m < 7 && 8 * m + 8 < n && !(pv[7] - nxt0 > E2)) ? 0x80u : 0u;
            amb |= (m > 0 && 8 * m < n && !(prv7 - pv[0] > E2)) ? 1u : 0u;
            const int nfl = __builtin_popcount(amb);
            int nflag = 0, foff = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) { const int f = __shfl(nfl, ql8 + 8 * t, 64); nflag += f; foff += t < m ? f : 0; }
            unsigned* slots = wlbase + wave * 512 + ql8 * 64 + foff;     // this lane's part of the query's [64]: candidate index in, canonical distance out
            {
                int c = 0;
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    if ((amb >> a) & 1u) { slots[c] = (unsigned)jv[a]; ++c; }
                }
            }
            int fmaxw = nflag;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) fmaxw = max(fmaxw, __shfl_xor(fmaxw, o, 64));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (fmaxw > 0) {                                    // canonical distances of the flagged pairs, densely packed over the wave's lanes
                int offs[9];
                offs[0] = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) offs[q + 1] = offs[q] + __builtin_amdgcn_readlane(nflag, q);
                const int total = offs[8];
                unsigned* wslots = wlbase + wave * 512;
                const int qbase = chunk * 64 + qg * 32 + ch * 8;
                for (int i0 = 0; i0 < total; i0 += 64) {
                    const int i = i0 + lane;
                    if (i < total) {
                        int q = 0, ob = 0;
#pragma unroll
                        for (int t = 1; t < 8; ++t) { const bool ge = i >= offs[t]; q = ge ? t : q; ob = ge ? offs[t] : ob; }
                        unsigned* sp = wslots + q * 64 + (i - ob);
                        const int j = (int)*sp;
                        const int qr = qbase + q;
                        const float* rq = xb + (size_t)qr * ld;
                        const float* rj = xb + (size_t)j * ld;
                        float acc = 0.f;
                        if (!xvec) {
                            for (int c = 0; c < C; ++c) acc = fmaf(rq[c], rj[c], acc);
                        } else {
                            constexpr int CHV = CT >= 128 ? 8 : 16;
                            for (int c = 0; c < C; c += 4 * CHV) {
                                f32x4 a4[CHV], b4[CHV];
#pragma unroll
                                for (int u = 0; u < CHV; ++u)
                                    if (c + 4 * u < C) { a4[u] = *(const f32x4*)(rq + c + 4 * u); b4[u] = *(const f32x4*)(rj + c + 4 * u); }
#pragma unroll
                                for (int u = 0; u < CHV; ++u)
                                    if (c + 4 * u < C) {
#pragma unroll
                                        for (int e = 0; e < 4; ++e) acc = fmaf(a4[u][e], b4[u][e], acc);
                                    }
                            }
                        }
                        const float t2 = fmaf(2.0f, acc, -xxb[j]);
                        *sp = (unsigned)__float_as_int(t2 - xxb[qr]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            {
                int c = 0;
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    if ((amb >> a) & 1u) { pv[a] = __int_as_float((int)slots[c]); ++c; }
                }
            }
            if (fmaxw > 0) {                                    // settle the flagged runs under the full order (value desc, index asc)
                for (int pass = 0; pass < 64; ++pass) {
                    bool moved = false;
#pragma unroll
                    for (int a = 0; a < 7; a += 2) {
                        const bool sw = k6_beats(pv[a + 1], jv[a + 1], pv[a], jv[a]);
                        const float ta = sw ? pv[a + 1] : pv[a], tc = sw ? pv[a] : pv[a + 1];
                        const int ja = sw ? jv[a + 1] : jv[a], jc = sw ? jv[a] : jv[a + 1];
                        pv[a] = ta; pv[a + 1] = tc; jv[a] = ja; jv[a + 1] = jc;
                        moved |= sw;
                    }
#pragma unroll
                    for (int a = 1; a < 7; a += 2) {
                        const bool sw = k6_beats(pv[a + 1], jv[a + 1], pv[a], jv[a]);
                        const float ta = sw ? pv[a + 1] : pv[a], tc = sw ? pv[a] : pv[a + 1];
                        const int ja = sw ? jv[a + 1] : jv[a], jc = sw ? jv[a] : jv[a + 1];
                        pv[a] = ta; pv[a + 1] = tc; jv[a] = ja; jv[a + 1] = jc;
                        moved |= sw;
                    }
                    {                                            // the odd pair across the lane boundary: (8 m + 7, 8 (m + 1))
                        const float nv = __shfl(pv[0], lane + 8, 64), pvv = __shfl(pv[7], lane - 8, 64);
                        const int nj = __shfl(jv[0], lane + 8, 64), pj = __shfl(jv[7], lane - 8, 64);
                        const bool swh = m < 7 && k6_beats(nv, nj, pv[7], jv[7]);
                        const bool swl = m > 0 && k6_beats(pv[0], jv[0], pvv, pj);
                        pv[7] = swh ? nv : pv[7]; jv[7] = swh ? nj : jv[7];
                        pv[0] = swl ? pvv : pv[0]; jv[0] = swl ? pj : jv[0];
                        moved |= swh | swl;
                    }
                    if (!__any(moved)) break;
                }
            }
            {
                int* out = idx + ((size_t)b * N + qrow) * k;
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const int pos = 8 * m + a;
                    if (pos < k && pos < n) out[pos] = jv[a];
                }
            }
            return;
        }
    }
    // ---- counting final (a query of the wave has more than 64 survivors): two queries per trip (one per half-wave), lane l31 owns entries
    // l31 + 32 s of the query's eight concatenated lists; logic as in knn6_kernel
    unsigned* wl = wlbase + wave * 512;
    int wcnt = 0;
    auto flush = [&]() {
        for (int i0 = 0; i0 < wcnt; i0 += 64) {
            const int i = i0 + lane;
            if (i < wcnt) {
                const unsigned item = wl[i];
                const int qq_ = (int)(item >> 16);
                char* ent = lists + (size_t)(item & 0xffffu) * 8;
                const int j = *(const int*)(ent + 4);
                const int qrow_ = chunk * 64 + qq_;
                const float* rq = xb + (size_t)qrow_ * ld;
                const float* rj = xb + (size_t)j * ld;
                float acc = 0.f;
                if (xvec) {
                    for (int c = 0; c < C; c += 32) {
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
        for (int i0 = 0; i0 < wcnt; i0 += 64) {
            const int i = i0 + lane;
            const bool on = i < wcnt;
            const unsigned item = on ? wl[i] : 0u;
            const int qq_ = (int)(item >> 16), qlc_ = qq_ & 31;
            const k6u32x2 me = *(const k6u32x2*)(lists + (size_t)(item & 0xffffu) * 8);
            const float pm = __int_as_float((int)me[0]);
            const int jm = (int)me[1];
            int rank = 0;
            for (int t = 0; t < 8; ++t) {
                const char* Lq = list_at(qlc_, t);
#pragma unroll
                for (int p0 = 0; p0 < K6_CAP; p0 += 8) {
                    k6u32x2 ke[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) ke[u] = *(const k6u32x2*)(Lq + (p0 + u) * 8);
#pragma unroll
                    for (int u = 0; u < 8; ++u) rank += k6_beats(__int_as_float((int)ke[u][0]), (int)ke[u][1], pm, jm) ? 1 : 0;
                }
            }
            if (on && rank < k) idx[((size_t)b * N + chunk * 64 + qq_) * k + rank] = jm;
        }
        wcnt = 0;
    };
    for (int it = 0; it < 4; ++it) {
        const int qlc = ch * 8 + it * 2 + h;                   // query of the group
        const int qq = qg * 32 + qlc;                          // query of the workgroup
        const int qrow = chunk * 64 + qq;                      // query of the cloud
        const float xq = xxb[qrow], xqc = xcb[qrow];
        const k6i32x4 ca = *(const k6i32x4*)(cnts + qq * 8), cb = *(const k6i32x4*)(cnts + qq * 8 + 4);
        const f32x4 ma = *(const f32x4*)(lmn + qq * 8), mb = *(const f32x4*)(lmn + qq * 8 + 4);
        const float xm = -2.0f * fminf(fminf(fminf(ma[0], ma[1]), fminf(ma[2], ma[3])), fminf(fminf(mb[0], mb[1]), fminf(mb[2], mb[3])));
        const float E2 = 2.0f * (K6_EPS * (xqc + xm) + K6_CANON * (xq + xxmax));
        int pp[9];
        pp[0] = 0; pp[1] = ca[0]; pp[2] = pp[1] + ca[1]; pp[3] = pp[2] + ca[2]; pp[4] = pp[3] + ca[3];
        pp[5] = pp[4] + cb[0]; pp[6] = pp[5] + cb[1]; pp[7] = pp[6] + cb[2]; pp[8] = pp[7] + cb[3];
        const int n = pp[8];
        const int nmax = max(__builtin_amdgcn_readlane(n, 0), __builtin_amdgcn_readlane(n, 32));
        int cmx = max(max(max(ca[0], ca[1]), max(ca[2], ca[3])), max(max(cb[0], cb[1]), max(cb[2], cb[3])));
        cmx = max(__builtin_amdgcn_readlane(cmx, 0), __builtin_amdgcn_readlane(cmx, 32));
        const char* Lq0 = list_at(qlc, 0);                     // list t of the query: Lq0 + t * 32 * K6_LSTR
        for (int s0 = 0; s0 < nmax; s0 += 32) {
            const int e = s0 + l31;
            const bool valid = e < n;
            int t = 0, pb = 0;
#pragma unroll
            for (int u = 1; u < 8; ++u) { const bool ge = e >= pp[u]; t = ge ? u : t; pb = ge ? pp[u] : pb; }
            const char* mine = Lq0 + (size_t)(valid ? t : 0) * (32 * K6_LSTR) + (valid ? e - pb : 0) * 8;
            const k6u32x2 me = *(const k6u32x2*)mine;
            const float pm = __int_as_float((int)me[0]);
            const int j = (int)me[1];
            int rank = 0, near = 0;
            for (int p0 = 0; p0 < cmx; ++p0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float a0 = *(const float*)(Lq0 + (size_t)u * (32 * K6_LSTR) + p0 * 8);
                    rank += a0 > pm ? 1 : 0;
                    near += fabsf(a0 - pm) <= E2 ? 1 : 0;
                }
            }
            const bool amb = valid && !(near <= 1);
            if (valid && !amb && rank < k) idx[((size_t)b * N + qrow) * k + rank] = j;
            const unsigned long long mm = __ballot(amb);
            if (mm) {
                const int before = __builtin_popcountll(mm & ((1ull << lane) - 1ull));
                if (amb) wl[wcnt + before] = (unsigned)((mine - lists) >> 3) | ((unsigned)qq << 16);
                wcnt += __builtin_popcountll(mm);
            }
        }
        if (wcnt > 512 - 384 || (it == 3 && wcnt > 0)) {       // (a trip adds at most 2 x 192 items)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            flush();
        }
    }
}

size_t knn6_lds_bytes(int N) { return (size_t)512 * K6_LSTR + (size_t)N * 4 + 128 * 4 + 512 * 4 + 64 + (size_t)8 * 512 * 4 + 512 * 4; }

// shapes v6 takes (the rest stays on knn.hip's kernels)
// the vector-exact mode of the kernel: C <= 3 on whole 128-query chunks, the cloud's {x, xx} image in the 16 KB work-list area
bool knn6_vex_supported(int B, int N, int C, int k) {
    return B > 0 && N >= 128 && N % 128 == 0 && N <= 1024 && C >= 1 && C <= 3 && k >= 1 && k <= K6_KMAX && k <= N;
}
size_t knn6_vex_bytes(int P) { return (size_t)P * 16; }
int launch_knn6_vex(hipStream_t st, const float* x, int ld, int B, int N, int C, int k, int* idx, float* xx, void* cand) {
    if (!knn6_vex_supported(B, N, C, k) || !cand || (((uintptr_t)cand) & 15)) return MLSP_ERR_UNSUPPORTED;
    const int P = B * N;
    hipLaunchKernelGGL(knn6_prep_vex_kernel, dim3((P + 255) / 256), dim3(256), 0, st, x, ld, P, C, xx, (f32x4*)cand, idx, k);
    const size_t lds = knn6_lds_bytes(N);
    hipError_t e = mlsp_lds_limit((const void*)knn6_kernel<16, true>, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((knn6_kernel<16, true>), dim3((N / 128) * B), dim3(512), lds, st, x, ld, xx, xx, (const char*)cand, N, C, k, idx, B);
    return mlsp_launch_status();
}

bool knn6_supported(int B, int N, int C, int k) {
    return B > 0 && N >= 128 && N % 128 == 0 && N <= 4096 && C >= 1 && C <= 128 && k >= 1 && k <= K6_KMAX && k <= N &&
           knn6_lds_bytes(N) <= 160 * 1024;
}
int knn6_padded_channels(int C) { return C <= 16 ? 16 : C <= 64 ? 64 : 128; }
// image + centred norms + the wide kernel's cloud flags (B <= P / 128 ints)
size_t knn6_plane_bytes(int P, int C) { return (size_t)P * 2 * knn6_padded_channels(C) * sizeof(__bf16) + (size_t)P * sizeof(float) + ((size_t)P / 128 + 4) * sizeof(int); }
bool knn6w_supported(int B, int N, int C, int k) {
    return B > 0 && N >= 128 && N % 128 == 0 && N <= 4096 && C >= 1 && C <= 128 && k > K6_KMAX && k <= K6W_KMAX && k <= N &&
           knn6_lds_bytes(N) <= 160 * 1024;
}

template <int CT>
static int knn6_go(hipStream_t st, const float* x, int ld, int B, int N, int C, int k, int* idx, float* xx, float* xc, char* planes) {
    const int P = B * N;
    hipLaunchKernelGGL((knn6_prep_kernel<CT>), dim3((P + 255) / 256), dim3(256), 0, st, x, ld, P, N, C, xx, xc, planes, idx, k, (int*)nullptr);
    const size_t lds = knn6_lds_bytes(N);
    hipError_t e = mlsp_lds_limit((const void*)knn6_kernel<CT>, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((knn6_kernel<CT>), dim3((N / 128) * B), dim3(512), lds, st, x, ld, xx, xc, planes, N, C, k, idx, B);
    return mlsp_launch_status();
}

template <int CT>
static int knn6w_go(hipStream_t st, const float* x, int ld, int B, int N, int C, int k, int* idx, float* xx, float* xc, char* planes, int* flags) {
    const int P = B * N;
    hipLaunchKernelGGL((knn6_prep_kernel<CT>), dim3((P + 255) / 256), dim3(256), 0, st, x, ld, P, N, C, xx, xc, planes, idx, k, flags);
    const size_t lds = knn6_lds_bytes(N);
    hipError_t e = mlsp_lds_limit((const void*)knn6w_kernel<CT>, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((knn6w_kernel<CT>), dim3((N / 64) * B), dim3(512), lds, st, x, ld, xx, xc, planes, N, C, k, idx, B, flags);
    return mlsp_launch_status();
}
// 24 < k <= 40: prep + knn6w_kernel; *flags_out = the per-cloud flags ([B] ints inside `planes`) the caller hands to the v5 launch behind it
int launch_knn6w(hipStream_t st, const float* x, int ld, int B, int N, int C, int k, int* idx, float* xx, void* planes, int** flags_out) {
    if (!knn6w_supported(B, N, C, k) || !planes || (((uintptr_t)planes) & 15)) return MLSP_ERR_UNSUPPORTED;
    const int CT = knn6_padded_channels(C);
    float* xc = (float*)((char*)planes + (size_t)B * N * CT * 4);
    int* flags = (int*)(xc + (size_t)B * N);
    *flags_out = flags;
    if (CT == 16) return knn6w_go<16>(st, x, ld, B, N, C, k, idx, xx, xc, (char*)planes, flags);
    if (CT == 64) return knn6w_go<64>(st, x, ld, B, N, C, k, idx, xx, xc, (char*)planes, flags);
    return knn6w_go<128>(st, x, ld, B, N, C, k, idx, xx, xc, (char*)planes, flags);
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
