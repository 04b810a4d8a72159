// Experimental variants of the six-product bf16 split GEMM (see x6_bench.hip).  128x128x32 tiles, 256 threads (2 x 2 waves of 64 x 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

struct X6Args { const float* A; const float* B; float* C; int M, N, K, lda, ldb, ldc, nsplit, ksplit, ntm, ntn, xcd_map; };

#define X6_PLANE 10240            // bytes of one bf16 image of a 128 x 32 operand tile: [128 rows][80 B] or [32 k][320 B]
#define X6_RPITCH 80              // row-major-source image: bytes per row (32 k + 8 pad)
#define X6_KPITCH 320             // k-major-source image: bytes per k   (128 rows + 32 pad): the 4 k-rows of a transposed read fall in 4 different 64-byte bank windows

__device__ __forceinline__ void x6_split4(const f32x4& x, bf16x4& a, bf16x4& b, bf16x4& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v = x[e];
        const __bf16 ha = (__bf16)v;
        const float r1 = v - (float)ha;
        const __bf16 hb = (__bf16)r1;
        const float r2 = r1 - (float)hb;
        a[e] = ha; b[e] = hb; c[e] = (__bf16)r2;
    }
}

// global -> registers: one 128 x 32 fp32 operand tile, 4 x 16 bytes per thread
template <bool KMAJ>
__device__ __forceinline__ const float* x6_src(const float* base, int ld, int row0, int k0, int tid) {
    return KMAJ ? base + (size_t)(k0 + (tid >> 5)) * ld + row0 + (tid & 31) * 4
                : base + (size_t)(row0 + (tid >> 3)) * ld + k0 + (tid & 7) * 4;
}
template <bool KMAJ>
__device__ __forceinline__ void x6_g2r(f32x4 (&r)[4], const float* __restrict__ src, int ld) {
#pragma unroll
    for (int p = 0; p < 4; ++p) r[p] = *(const f32x4*)(src + (size_t)((KMAJ ? 8 : 32) * p) * ld);
}
// registers -> the three LDS images (split on the way)
template <bool KMAJ>
__device__ __forceinline__ void x6_r2s(const f32x4 (&r)[4], char* img, int tid) {
    const int off = KMAJ ? (tid >> 5) * X6_KPITCH + (tid & 31) * 8 : (tid >> 3) * X6_RPITCH + (tid & 7) * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        bf16x4 a, b, c;
        x6_split4(r[p], a, b, c);
        char* d = img + off + p * (KMAJ ? 8 * X6_KPITCH : 32 * X6_RPITCH);
        *(bf16x4*)(d) = a;
        *(bf16x4*)(d + X6_PLANE) = b;
        *(bf16x4*)(d + 2 * X6_PLANE) = c;
    }
}
// one MFMA operand fragment (rows rb .. rb+31 of the tile, k = 16 s2 + 8 h .. + 7) of image plane `img`
//   row-major image: one 16-byte read.  k-major image: two transposed 8-byte reads (ds_read_b64_tr_b16).
template <bool KMAJ>
__device__ __forceinline__ int x6_frag_base(int lane) {
    const int l31 = lane & 31, h = lane >> 5;
    if (!KMAJ) return l31 * X6_RPITCH + 16 * h;
    const int i = lane & 15, q = i >> 2, pp = i & 3, g1 = (lane >> 4) & 1;
    return (8 * h + q) * X6_KPITCH + (16 * g1 + 4 * pp) * 2;
}
template <bool KMAJ>
__device__ __forceinline__ bf16x8 x6_frag(const char* img, int fbase, int rb, int s2) {
    if (!KMAJ) return *(const bf16x8*)(img + fbase + rb * X6_RPITCH + 32 * s2);
    const char* a = img + fbase + rb * 2 + 16 * s2 * X6_KPITCH;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(a));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4*)(a + 4 * X6_KPITCH));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ void x6_tile_of_block(const X6Args& p, int& tm, int& tn, bool& live) {
    const int bid = blockIdx.x;
    live = true;
    if (p.xcd_map) {
        const int xcd = bid & 7, q = bid >> 3;
        tn = q % p.ntn;
        tm = (q / p.ntn) * 8 + xcd;
        live = tm < p.ntm;
    } else {
        tn = bid % p.ntn;
        tm = bid / p.ntn;
    }
}

__device__ __forceinline__ void x6_store(const X6Args& p, f32x16 (&acc)[2][2], int m0, int n0, int split, int l31, int h, int wm, int wn) {
    float* C = p.C + (size_t)split * p.M * p.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = n0 + wn * 64 + j * 32 + l31;
                C[(size_t)row * p.ldc + col] = acc[i][j][r];
            }
}

#define X6_MFMA6(ACC, FA, FB) do { \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1], FB[1], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], FB[2], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[2], FB[0], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], FB[1], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1], FB[0], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], FB[0], ACC, 0, 0, 0); } while (0)

template <int G, int NR>
__device__ __forceinline__ void x6_sched_steps() {
    if constexpr (G < 48) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        constexpr int nr = G < 24 ? (G + 1) * NR / 24 - G * NR / 24 : 0;
        if constexpr (nr > 0) __builtin_amdgcn_sched_group_barrier(0x100, nr, 0);
        if constexpr (G % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if constexpr (G % 6 == 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        x6_sched_steps<G + 1, NR>();
    }
}

// one K-tile step of the double-buffered kernels: MFMAs of the tile in `cur`, split + image write of the next tile into `nxt`, global loads
// of the tile after it.  `cur` and `nxt` are different __shared__ objects, so the compiler may move the image writes among the fragment reads.
template <bool KA, bool KB, int SCHED>
__device__ __forceinline__ void x6_step(const char* __restrict__ cur, char* __restrict__ nxt, f32x16 (&acc)[2][2], f32x4 (&ra)[4], f32x4 (&rb)[4], const float* pa, const float* pb,
                                        int lda, int ldb, int fa, int fb, int tid, int wm, int wn) {
    x6_r2s<KA>(ra, nxt, tid);
    x6_r2s<KB>(rb, nxt + 3 * X6_PLANE, tid);
    x6_g2r<KA>(ra, pa, lda);
    x6_g2r<KB>(rb, pb, ldb);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i][q] = x6_frag<KA>(cur + q * X6_PLANE, fa, wm * 64 + 32 * i, s2);
                b[i][q] = x6_frag<KB>(cur + (3 + q) * X6_PLANE, fb, wn * 64 + 32 * i, s2);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) X6_MFMA6(acc[i][j], a[i], b[j]);
    }
    if (SCHED == 1) {
        // one basic block: 48 MFMA, ~192 VALU (the split), 24 (48 transposed) fragment reads, 12 image writes, 8 global loads.
        // MFMA shadow = 32 cycles = 8 issue slots: 1 MFMA + 4 VALU + at most one memory instruction per slot group.
        constexpr int NR = (KA ? 12 : 6) + (KB ? 12 : 6);             // DS reads per k16 step
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);             // fragments of step 0
        x6_sched_steps<0, NR>();
    }
    __syncthreads();
}

// ---- variant 0: one LDS buffer, two barriers per K-tile, two workgroups per CU (the structure of the fp32 kernel) ----------------
// ---- variant 1: two LDS buffers, one barrier per K-tile, one workgroup per CU
// ---- variant 2: variant 1 + the split of tile t+1 interleaved with the MFMAs of tile t (sched_group_barrier)
template <bool KA, bool KB, int NBUF, int SCHED = 0>
__global__ __launch_bounds__(256, NBUF == 1 ? 2 : 1) void x6_kernel(X6Args p) {
    __shared__ __attribute__((aligned(16))) char simg[NBUF * 6 * X6_PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    int tm, tn; bool live;
    x6_tile_of_block(p, tm, tn, live);
    if (!live) return;
    const int split = blockIdx.y, m0 = tm * 128, n0 = tn * 128;
    const int kbeg = split * p.ksplit, T = p.ksplit / 32;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[4], rb[4];
    const float* pa = x6_src<KA>(p.A, p.lda, m0, kbeg, tid);
    const float* pb = x6_src<KB>(p.B, p.ldb, n0, kbeg, tid);
    const size_t sa = KA ? (size_t)32 * p.lda : 32, sb = KB ? (size_t)32 * p.ldb : 32;
    const int fa = x6_frag_base<KA>(lane), fb = x6_frag_base<KB>(lane);
    x6_g2r<KA>(ra, pa, p.lda);
    x6_g2r<KB>(rb, pb, p.ldb);
    if (NBUF == 1) {
        for (int t = 0; t < T; ++t) {
            x6_r2s<KA>(ra, simg, tid);
            x6_r2s<KB>(rb, simg + 3 * X6_PLANE, tid);
            __syncthreads();
            if (t + 1 < T) { pa += sa; pb += sb; x6_g2r<KA>(ra, pa, p.lda); x6_g2r<KB>(rb, pb, p.ldb); }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 a[2][3], b[2][3];
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        a[i][q] = x6_frag<KA>(simg + q * X6_PLANE, fa, wm * 64 + 32 * i, s2);
                        b[i][q] = x6_frag<KB>(simg + (3 + q) * X6_PLANE, fb, wn * 64 + 32 * i, s2);
                    }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) X6_MFMA6(acc[i][j], a[i], b[j]);
            }
            __syncthreads();
        }
    } else {
        // tile 0 -> buffer 0; registers <- tile 1.  Every step re-splits what the registers hold into the idle buffer and reloads them:
        // past the end that is a harmless repeat of the last tile (no branch inside a step).  T is even (launcher).
        x6_r2s<KA>(ra, simg, tid);
        x6_r2s<KB>(rb, simg + 3 * X6_PLANE, tid);
        const float* ea = pa + (size_t)(T - 1) * sa; const float* eb = pb + (size_t)(T - 1) * sb;      // last tile
        pa += sa; pb += sb;
        x6_g2r<KA>(ra, pa, p.lda);
        x6_g2r<KB>(rb, pb, p.ldb);
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            pa = pa + sa <= ea ? pa + sa : ea; pb = pb + sb <= eb ? pb + sb : eb;
            x6_step<KA, KB, SCHED>(simg + (t & 1) * 6 * X6_PLANE, simg + ((t & 1) ^ 1) * 6 * X6_PLANE, acc, ra, rb, pa, pb, p.lda, p.ldb, fa, fb, tid, wm, wn);
        }
    }
    x6_store(p, acc, m0, n0, split, l31, h, wm, wn);
}

// ---- variant 3: one LDS buffer, two workgroups per CU; all 24 fragments of tile t are read first (barrier), then the 48 MFMAs of tile t run with
// the split + image write of tile t+1 and the global loads of tile t+2 hand-placed between them: one MFMA + 3-4 vector instructions per slot,
// pinned by sched_barrier(0) (the MFMA shadow is 32 cycles = 8 issue slots of 4; the MFMA itself takes 2 of them).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t x6_cvt_pk(float lo, float hi) {
    const f32x2_ v = {lo, hi};
    const bf16x2 b = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, b);
}

template <bool KA, bool KB, int ABL = 0>
__global__ __launch_bounds__(256, 2) void x6_kernel_v3(X6Args p) {
    __shared__ __attribute__((aligned(16))) char simg[6 * X6_PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    int tm, tn; bool live;
    x6_tile_of_block(p, tm, tn, live);
    if (!live) return;
    const int split = blockIdx.y, m0 = tm * 128, n0 = tn * 128;
    const int kbeg = split * p.ksplit, T = p.ksplit / 32;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 raw[8];                                  // [0..3] A quads, [4..7] B quads of the tile being staged
    const float* pa = x6_src<KA>(p.A, p.lda, m0, kbeg, tid);
    const float* pb = x6_src<KB>(p.B, p.ldb, n0, kbeg, tid);
    const size_t sa = KA ? (size_t)32 * p.lda : 32, sb = KB ? (size_t)32 * p.ldb : 32;
    const size_t qa_ = (size_t)(KA ? 8 : 32) * p.lda, qb_ = (size_t)(KB ? 8 : 32) * p.ldb;       // quad p of a tile = + p * q
    const float* ea = pa + (size_t)(T - 1) * sa; const float* eb = pb + (size_t)(T - 1) * sb;      // last tile
    const int fa = x6_frag_base<KA>(lane), fb = x6_frag_base<KB>(lane);
    char* wa = simg + (KA ? (tid >> 5) * X6_KPITCH + (tid & 31) * 8 : (tid >> 3) * X6_RPITCH + (tid & 7) * 8);
    char* wb = simg + 3 * X6_PLANE + (KB ? (tid >> 5) * X6_KPITCH + (tid & 31) * 8 : (tid >> 3) * X6_RPITCH + (tid & 7) * 8);
    constexpr int WQA = KA ? 8 * X6_KPITCH : 32 * X6_RPITCH, WQB = KB ? 8 * X6_KPITCH : 32 * X6_RPITCH;
    {
        f32x4 ra[4], rb[4];
        x6_g2r<KA>(ra, pa, p.lda);
        x6_g2r<KB>(rb, pb, p.ldb);
        x6_r2s<KA>(ra, simg, tid);
        x6_r2s<KB>(rb, simg + 3 * X6_PLANE, tid);
        pa = pa + sa <= ea ? pa + sa : ea; pb = pb + sb <= eb ? pb + sb : eb;
#pragma unroll
        for (int q = 0; q < 4; ++q) { raw[q] = *(const f32x4*)(pa + q * qa_); raw[4 + q] = *(const f32x4*)(pb + q * qb_); }
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        bf16x8 a[2][2][3], b[2][2][3];                                   // [k16 step][32-row tile][piece]
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[s2][i][q] = x6_frag<KA>(simg + q * X6_PLANE, fa, wm * 64 + 32 * i, s2);
                    b[s2][i][q] = x6_frag<KB>(simg + (3 + q) * X6_PLANE, fb, wn * 64 + 32 * i, s2);
                }
        __syncthreads();                                                 // every wave holds its fragments: the images may be overwritten
        pa = pa + sa <= ea ? pa + sa : ea; pb = pb + sb <= eb ? pb + sb : eb;          // tile t+2 (clamped: a harmless repeat at the end)
        uint32_t pk[3][2];                                               // packed pieces of the quad in flight: [piece][pair]
        float r0 = 0.f, r1 = 0.f, a1 = 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 48; ++c) {
            constexpr int QA[6] = {1, 0, 2, 0, 1, 0}, QB[6] = {1, 2, 0, 1, 0, 0};
            const int s2 = c / 24, ij = (c % 24) / 6, i = ij >> 1, j = ij & 1, pr6 = c % 6;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s2][i][QA[pr6]], b[s2][j][QB[pr6]], acc[i][j], 0, 0, 0);
            const int pr = c / 3, qd = pr >> 1, hh = pr & 1, ms = c % 3;
            const float x0 = raw[qd][2 * hh], x1 = raw[qd][2 * hh + 1];
            const bool nosplit = (ABL & 1) || ((ABL & 32) && qd >= 4);
            if (nosplit) {
                if (ms == 0) { pk[0][hh] = __float_as_uint(x0); pk[1][hh] = __float_as_uint(x1); pk[2][hh] = __float_as_uint(x0); }
            } else if (ms == 0) {
                pk[0][hh] = x6_cvt_pk(x0, x1);
                const float a0 = __uint_as_float(pk[0][hh] << 16);
                a1 = __uint_as_float(pk[0][hh] & 0xffff0000u);
                r0 = x0 - a0;
            } else if (ms == 1) {
                r1 = x1 - a1;
                pk[1][hh] = x6_cvt_pk(r0, r1);
            } else {
                const float b0 = __uint_as_float(pk[1][hh] << 16), b1 = __uint_as_float(pk[1][hh] & 0xffff0000u);
                pk[2][hh] = x6_cvt_pk(r0 - b0, r1 - b1);
            }
            if (ms == 2) {
                if (hh == 1) {
                    char* d = qd < 4 ? wa + qd * WQA : wb + (qd - 4) * WQB;
#pragma unroll
                    for (int q = 0; q < 3; ++q) { const u32x2_ v = {pk[q][0], pk[q][1]}; *(u32x2_*)(d + q * X6_PLANE) = v; }
                    raw[qd] = qd < 4 ? *(const f32x4*)(pa + qd * qa_) : *(const f32x4*)(pb + (qd - 4) * qb_);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    x6_store(p, acc, m0, n0, split, l31, h, wm, wn);
}

// ---- variant 4: two LDS buffers, ONE workgroup per CU (one wave per SIMD, so nothing contends for the SIMD's issue slots); hand-placed stream:
// 48 MFMAs of tile t; between them the reads of tile t's second-half fragments, the split + image write of tile t+1 (other buffer), the global
// loads of tile t+2; barrier after MFMA CB; then the first-half fragments of tile t+1 are read under the last 48 - CB MFMAs of tile t.
template <bool KA, bool KB, int CB, int ABL = 0>
__global__ __launch_bounds__(256, 1) void x6_kernel_v4(X6Args p) {
    __shared__ __attribute__((aligned(16))) char simg[2 * 6 * X6_PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    int tm, tn; bool live;
    x6_tile_of_block(p, tm, tn, live);
    if (!live) return;
    const int split = blockIdx.y, m0 = tm * 128, n0 = tn * 128;
    const int kbeg = split * p.ksplit, T = p.ksplit / 32;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 raw[8];
    const float* pa = x6_src<KA>(p.A, p.lda, m0, kbeg, tid);
    const float* pb = x6_src<KB>(p.B, p.ldb, n0, kbeg, tid);
    const size_t sa = KA ? (size_t)32 * p.lda : 32, sb = KB ? (size_t)32 * p.ldb : 32;
    const size_t qa_ = (size_t)(KA ? 8 : 32) * p.lda, qb_ = (size_t)(KB ? 8 : 32) * p.ldb;
    const float* ea = pa + (size_t)(T - 1) * sa; const float* eb = pb + (size_t)(T - 1) * sb;
    const int fa = x6_frag_base<KA>(lane) + (wm * 64) * (KA ? 2 : X6_RPITCH), fb = x6_frag_base<KB>(lane) + (wn * 64) * (KB ? 2 : X6_RPITCH) + 3 * X6_PLANE;
    const int woa = KA ? (tid >> 5) * X6_KPITCH + (tid & 31) * 8 : (tid >> 3) * X6_RPITCH + (tid & 7) * 8;
    const int wob = 3 * X6_PLANE + (KB ? (tid >> 5) * X6_KPITCH + (tid & 31) * 8 : (tid >> 3) * X6_RPITCH + (tid & 7) * 8);
    constexpr int WQA = KA ? 8 * X6_KPITCH : 32 * X6_RPITCH, WQB = KB ? 8 * X6_KPITCH : 32 * X6_RPITCH;
    constexpr int BUF = 6 * X6_PLANE;
    {
        f32x4 ra[4], rb[4];
        x6_g2r<KA>(ra, pa, p.lda);
        x6_g2r<KB>(rb, pb, p.ldb);
        x6_r2s<KA>(ra, simg, tid);
        x6_r2s<KB>(rb, simg + 3 * X6_PLANE, tid);
        pa = pa + sa <= ea ? pa + sa : ea; pb = pb + sb <= eb ? pb + sb : eb;
#pragma unroll
        for (int q = 0; q < 4; ++q) { raw[q] = *(const f32x4*)(pa + q * qa_); raw[4 + q] = *(const f32x4*)(pb + q * qb_); }
    }
    __syncthreads();
    bf16x8 a[2][2][3], b[2][2][3];                                   // [k16 step][32-row tile][piece]
    uint32_t junk = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[0][i][q] = x6_frag<KA>(simg + q * X6_PLANE, fa, 32 * i, 0);
            b[0][i][q] = x6_frag<KB>(simg + q * X6_PLANE, fb, 32 * i, 0);
            if (ABL & 8) { a[1][i][q] = x6_frag<KA>(simg + q * X6_PLANE, fa, 32 * i, 1); b[1][i][q] = x6_frag<KB>(simg + q * X6_PLANE, fb, 32 * i, 1); }
        }
    for (int t = 0; t < T; ++t) {
        const char* cur = simg + (t & 1) * BUF;
        char* nxt = simg + ((t & 1) ^ 1) * BUF;
        pa = pa + sa <= ea ? pa + sa : ea; pb = pb + sb <= eb ? pb + sb : eb;          // tile t+2 (clamped: a harmless repeat at the end)
        uint32_t pk0[2], pk1[2], pk2[2];
        float r0 = 0.f, r1 = 0.f, a1 = 0.f;
        __builtin_amdgcn_sched_barrier(0);
        if (CB == 42) {
#include "x6_body_cb42.inc"
        } else {
#include "x6_body_cb36.inc"
        }
    }
    if (ABL && junk == 0x12345u) acc[0][0][0] += 1.f;
    x6_store(p, acc, m0, n0, split, l31, h, wm, wn);
}

#define X6_NVARIANTS 14
static const char* x6_variant_name(int v) {
    switch (v) { case 0: return "1 buf, 2 barriers, 2 WG/CU"; case 1: return "2 bufs, 1 barrier, 1 WG/CU"; case 2: return "2 bufs, sched_group_barrier"; case 3: return "1 buf, hand-placed split"; case 4: return "2 bufs 1 WG/CU hand CB=42"; case 5: return "2 bufs 1 WG/CU hand CB=36"; case 6: return "v4 - split VALU"; case 7: return "v4 - LDS writes"; case 8: return "v4 - global loads"; case 9: return "v4 - frag reads"; case 10: return "v4 - barrier"; case 11: return "v4 - all (MFMA only)"; case 12: return "v3 - split VALU"; case 13: return "v3 - split of B only"; default: return "?"; }
}

template <int NBUF, int SCHED = 0>
static void x6_go(int ka, int kb, const X6Args& p, dim3 grid, hipStream_t st) {
    if (!ka && !kb) hipLaunchKernelGGL((x6_kernel<false, false, NBUF, SCHED>), grid, dim3(256), 0, st, p);
    else if (!ka && kb) hipLaunchKernelGGL((x6_kernel<false, true, NBUF, SCHED>), grid, dim3(256), 0, st, p);
    else if (ka && kb) hipLaunchKernelGGL((x6_kernel<true, true, NBUF, SCHED>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((x6_kernel<true, false, NBUF, SCHED>), grid, dim3(256), 0, st, p);
}

static bool x6_launch(int var, int ka, int kb, X6Args p, hipStream_t st) {
    p.ntm = p.M / 128; p.ntn = p.N / 128;
    p.xcd_map = (p.ntm % 8 == 0) ? 1 : 0;
    dim3 grid(p.ntm * p.ntn, p.nsplit);
    switch (var) {
        case 0: x6_go<1>(ka, kb, p, grid, st); return true;
        case 1: x6_go<2>(ka, kb, p, grid, st); return true;
        case 2: x6_go<2, 1>(ka, kb, p, grid, st); return true;
        case 3:
            if (!ka && !kb) hipLaunchKernelGGL((x6_kernel_v3<false, false>), grid, dim3(256), 0, st, p);
            else if (!ka && kb) hipLaunchKernelGGL((x6_kernel_v3<false, true>), grid, dim3(256), 0, st, p);
            else if (ka && kb) hipLaunchKernelGGL((x6_kernel_v3<true, true>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((x6_kernel_v3<true, false>), grid, dim3(256), 0, st, p);
            return true;
        case 4:
            if (!ka && !kb) hipLaunchKernelGGL((x6_kernel_v4<false, false, 42>), grid, dim3(256), 0, st, p);
            else if (!ka && kb) hipLaunchKernelGGL((x6_kernel_v4<false, true, 42>), grid, dim3(256), 0, st, p);
            else if (ka && kb) hipLaunchKernelGGL((x6_kernel_v4<true, true, 42>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((x6_kernel_v4<true, false, 42>), grid, dim3(256), 0, st, p);
            return true;
        case 5:
            if (!ka && !kb) hipLaunchKernelGGL((x6_kernel_v4<false, false, 36>), grid, dim3(256), 0, st, p);
            else if (!ka && kb) hipLaunchKernelGGL((x6_kernel_v4<false, true, 36>), grid, dim3(256), 0, st, p);
            else if (ka && kb) hipLaunchKernelGGL((x6_kernel_v4<true, true, 36>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((x6_kernel_v4<true, false, 36>), grid, dim3(256), 0, st, p);
            return true;
        case 6:
            if (!ka && !kb) hipLaunchKernelGGL((x6_kernel_v4<false, false, 42, 1>), grid, dim3(256), 0, st, p);
            else if (!ka && kb) hipLaunchKernelGGL((x6_kernel_v4<false, true, 42, 1>), grid, dim3(256), 0, st, p);
            else if (ka && kb) hipLaunchKernelGGL((x6_kernel_v4<true, true, 42, 1>), grid, dim3(256), 0, st, p);
            else return false;
            return true;
        case 7:
            if (!ka && !kb) hipLaunchKernelGGL((x6_kernel_v4<false, false, 42, 2>), grid, dim3(256), 0, st, p);
            else if (!ka && kb) hipLaunchKernelGGL((x6_kernel_v4<false, true, 42, 2>), grid, dim3(256), 0, st, p);
            else if (ka && kb) hipLaunchKernelGGL((x6_kernel_v4<true, true, 42, 2>), grid, dim3(256), 0, st, p);
            else return false;
            return true;
        case 8:
            if (!ka && !kb) hipLaunchKernelGGL((x6_kernel_v4<false, false, 42, 4>), grid, dim3(256), 0, st, p);
            else if (!ka && kb) hipLaunchKernelGGL((x6_kernel_v4<false, true, 42, 4>), grid, dim3(256), 0, st, p);
            else if (ka && kb) hipLaunchKernelGGL((x6_kernel_v4<true, true, 42, 4>), grid, dim3(256), 0, st, p);
            else return false;
            return true;
        case 9:
            if (!ka && !kb) hipLaunchKernelGGL((x6_kernel_v4<false, false, 42, 8>), grid, dim3(256), 0, st, p);
            else if (!ka && kb) hipLaunchKernelGGL((x6_kernel_v4<false, true, 42, 8>), grid, dim3(256), 0, st, p);
            else if (ka && kb) hipLaunchKernelGGL((x6_kernel_v4<true, true, 42, 8>), grid, dim3(256), 0, st, p);
            else return false;
            return true;
        case 10:
            if (!ka && !kb) hipLaunchKernelGGL((x6_kernel_v4<false, false, 42, 16>), grid, dim3(256), 0, st, p);
            else if (!ka && kb) hipLaunchKernelGGL((x6_kernel_v4<false, true, 42, 16>), grid, dim3(256), 0, st, p);
            else if (ka && kb) hipLaunchKernelGGL((x6_kernel_v4<true, true, 42, 16>), grid, dim3(256), 0, st, p);
            else return false;
            return true;
        case 11:
            if (!ka && !kb) hipLaunchKernelGGL((x6_kernel_v4<false, false, 42, 31>), grid, dim3(256), 0, st, p);
            else if (!ka && kb) hipLaunchKernelGGL((x6_kernel_v4<false, true, 42, 31>), grid, dim3(256), 0, st, p);
            else if (ka && kb) hipLaunchKernelGGL((x6_kernel_v4<true, true, 42, 31>), grid, dim3(256), 0, st, p);
            else return false;
            return true;
        case 12:
            if (!ka && !kb) hipLaunchKernelGGL((x6_kernel_v3<false, false, 1>), grid, dim3(256), 0, st, p);
            else if (!ka && kb) hipLaunchKernelGGL((x6_kernel_v3<false, true, 1>), grid, dim3(256), 0, st, p);
            else if (ka && kb) hipLaunchKernelGGL((x6_kernel_v3<true, true, 1>), grid, dim3(256), 0, st, p);
            else return false;
            return true;
        case 13:
            if (!ka && !kb) hipLaunchKernelGGL((x6_kernel_v3<false, false, 32>), grid, dim3(256), 0, st, p);
            else if (!ka && kb) hipLaunchKernelGGL((x6_kernel_v3<false, true, 32>), grid, dim3(256), 0, st, p);
            else if (ka && kb) hipLaunchKernelGGL((x6_kernel_v3<true, true, 32>), grid, dim3(256), 0, st, p);
            else return false;
            return true;
    }
    return false;
}
