// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, exact f32,
// 157 TFLOP/s dense peak == the fp32 roof of the chip).  Every dense contraction of the hot path
// (1x1 convs, Linears, their dgrad and wgrad) goes through this one kernel family.
//
//   C[M,N] = opA(A) * opB(B) (+ bias[N]) (+ gbias[row / rows_per_group][N])
//   opA(A)[m][k] = TA ? A[k*lda + m] : A[m*lda + k]
//   opB(B)[k][n] = TB ? B[n*ldb + k] : B[k*ldb + n]
//
//   forward  Y  = X  * W^T : TA=0 TB=1        (X [P,Cin] point-major, W [Cout,Cin] as torch stores it)
//   dgrad    dX = dY * W   : TA=0 TB=0
//   wgrad    dW = dY^T * X : TA=1 TB=0, K = P (split-K over the grid, slab + reduce: deterministic)
//
// Tiling: 128x128x32 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles of 32x32.
// LDS images follow the global layout of each operand, so staging is 16-byte loads and 16-byte LDS writes only:
// row-major sources -> [row][32+4] (an operand fragment is ONE ds_read_b64 per two MFMA steps), k-major sources ->
// [k][128] (ds_read_b32 per value).  K is consumed in groups of four (k = 4m+2h, then 4m+2h+1) so that both images
// feed the same k to the A and the B side.
// Global loads of tile t+1 are issued before the MFMA loop of tile t (register prefetch).
// Block -> tile mapping keeps all column tiles of one 128-row panel on one XCD (blockIdx % 8 is the
// XCD label under round-robin dispatch), so the activation panel is fetched into one L2 only.
#include "common.h"
#include "../../include/mlsp_hip.h"
#include <cstdlib>

#define BM 128
#define BN 128
#define BK 32
#define SROW 36    // LDS row pitch (floats) for operands whose global source is row-major: image [rows][BK+4]
#define SKMJ 128   // LDS stride for operands whose global source is k-major    [K][rows]

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* gbias;
    int M, N, K, lda, ldb, ldc, rows_per_group;
    int ntm, ntn, nsplit, ksplit;   // tiles; split-K count; K range per split (multiple of BK)
    int a_vec, b_vec;               // 1 if float4 global loads are legal for that operand
    int xcd_map;                    // 1: XCD-grouped block->tile map (grid padded to a multiple of 8 panels)
    int accumulate;                 // 1: C += result (beta = 1): several consumers of one activation sum their input gradients in place
    int fast_out;                   // 1: interior fp32 tiles may use the lean output pass (gemm_out_fast): set by launch_gemm
    int c_bf16;                     // 1: C holds bf16 (activation storage of BASELINE.json configs[4]); A / B element types are template arguments
    // operand transform (XF != 0): the operand is the PRE-BatchNorm output of the previous layer and is turned into that layer's
    // activated output while it is staged: act(x * scale[c] + shift[c]), dropout by the counter hash of element row * x_ld + c.
    // XF == 1: A is [M][K] row-major, c = k.  XF == 2: B is [K][N] k-major (the wgrad's X), c = n.
    // x_col: the operand's first column inside the [rows][x_ld] matrix the previous layer wrote (a column slice of a merged layer's output):
    // the dropout stream is indexed by the element's place in THAT matrix; x_scale / x_shift already point at the slice's first channel.
    const float* x_scale; const float* x_shift; int x_act; float x_slope; uint32_t x_thresh; float x_inv_keep; uint64_t x_seed; int x_ld; int x_col;
    // groups (FAST fp32 kernel only; launch_gemm `grp`): a block-diagonal product in ONE launch.  gmode 1 (forward / dgrad): column tiles
    // [g * gtiles, (g + 1) * gtiles) form group g, whose A columns start at A + g * a_gs and whose B operand is Bg[g]; C, bias and
    // the statistics keep the launch-wide column index.  gmode 2 (wgrad): ROW tiles are grouped, B = B + g * b_gs.
    int gmode, gtiles; long a_gs, b_gs; const float* Bg[4];
    // bs_y != null (a dgrad whose result C is the gradient w.r.t. the ACTIVATED output of the previous layer, gemm_split_kernel): the output
    // pass multiplies C by that layer's activation derivative and dropout mask -- from its pre-BatchNorm output bs_y, same coordinates as C,
    // row pitch bs_ldy -- stores the MASKED gradient d', and leaves per-row-panel column sums of d' and d' * yhat in stat_part: the
    // streaming reduction of that layer's BatchNorm backward happens here, where the gradient is produced.  bs_bn: its scale | shift |
    // mean | invstd rows (pitch bs_bnld) at C's column 0; bs_ld / bs_col: row pitch of its matrix and C's column 0 in it (dropout stream).
    const float* bs_y; int bs_ldy; const float* bs_bn; int bs_bnld; float bs_slope; uint32_t bs_thresh; float bs_ik; uint32_t bs_xH; int bs_ld4; int bs_col;
    // dy_y != null (gemm_split_kernel<.., DY>): the A operand is a layer's OUTPUT GRADIENT formed on the fly -- A holds the masked gradient
    // d' (what the consumer's dgrad left, see bs_*), dy_y the layer's pre-BatchNorm output at the same coordinates (same pitch), and the
    // BatchNorm backward dY = (d' + y * nk2[c] + c0[c]) * sc[c] is applied while the tile is staged: the streaming "apply" pass and the
    // dY tensor disappear.  dy_coef: rows c0 | nk2 | sc (pitch dy_cld) at A's channel 0 (dgrad: channel = k; wgrad: channel = m).
    const float* dy_y; const float* dy_coef; int dy_cld;
    // bs_amax (with bs_y; nullable): per-row-panel column maxima of |d'| [ntm][stat_ld] floats at C's column 0 -- a by-product for the
    // PRODUCER's backward, whose two-piece f16 products need a bound of d' (dy_amax: its coefficient rows carry a fourth row, max |d'| per channel)
    float* bs_amax; int dy_amax;
    // gemm_split_kernel<.., NPC = 2> (two f16 pieces): device-side upper bounds of |A| and |B| (after the operand transforms), nullable = scale 1
    // Partial maxima of |A| / |B| as they lie in memory (amax_partials_kernel; null / 0 where the bound is analytic), and what the analytic
    // bounds need: a transformed operand (XF, batch statistics over stat_rows rows) is bounded per channel by |scale| sqrt(rows) / invstd +
    // |shift + mean scale| (|yhat| <= sqrt(rows)), a gradient formed on the fly (DY) by max|sc| amax(d') (2 + sqrt(rows)).
    const float* a_amax; int a_amax_n; const float* b_amax[4]; int b_amax_n[4]; const float* x_mean; const float* x_invstd; float stat_sqrt_rows;   // b_amax[g]: group g's B (gmode 1), else [0]
    double* stat_part;              // nullable: per-row-panel column sums of C and C^2, [ntm][2][stat_ld] (BN batch statistics)
    int stat_ld;                    // columns of a statistics row (N, or the width of the wider matrix C is a column slice of)
    const float* sel_gamma;         // nullable: per-column sign selects max (>= 0) or min; enables the fused column-extreme epilogue
    float* sel_val; int* sel_row;   // [ntm][N] extreme of each 128-row panel and the global row attaining it (first occurrence)
};

// ---- global -> registers for one 128x32 operand tile -----------------------------------------
// SRC_KMAJOR = false: source [rows][K] (ld = row pitch). thread t: k-quad (t&7)*4, rows (t>>3)+32p.
// SRC_KMAJOR = true : source [K][rows] (ld = k pitch).   thread t: row-quad (t&31)*4, k (t>>5)+8p.
template <bool SRC_KMAJOR, int NP>
__device__ __forceinline__ void g2r(f32x4 (&r)[4], const float* __restrict__ src, int ld, int row0, int nrows,
                                    int k0, int kend, int vec_ok, int tid) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!SRC_KMAJOR) {
            int row = row0 + (tid >> 3) + 32 * p;
            int k = k0 + (tid & 7) * 4;
            if (row < nrows) {
                const float* g = src + (size_t)row * ld + k;
                if (vec_ok && k + 3 < kend) {
                    v = *(const f32x4*)g;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < kend) v[e] = g[e];
                }
            }
        } else {
            // NP == 4: 128 rows (32 threads per k-row, 8 k-rows per pass); NP == 2: 64 rows (16 threads per k-row, 16 per pass)
            int k = k0 + (NP == 4 ? (tid >> 5) + 8 * p : (tid >> 4) + 16 * p);
            int row = row0 + (NP == 4 ? (tid & 31) : (tid & 15)) * 4;
            if (k < kend) {
                const float* g = src + (size_t)k * ld + row;
                if (vec_ok && row + 3 < nrows) {
                    v = *(const f32x4*)g;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (row + e < nrows) v[e] = g[e];
                }
            }
        }
        r[p] = v;
    }
}

// FAST path (every tile interior, 16-byte loads legal): no predicates, the per-thread source pointer just advances by one
// K-tile per iteration.  `base` already points at this thread's first element of the tile at k0.
template <bool SRC_KMAJOR, int NP>
__device__ __forceinline__ void g2r_fast(f32x4 (&r)[4], const float* __restrict__ base, int ld) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (!SRC_KMAJOR) r[p] = *(const f32x4*)(base + (size_t)(32 * p) * ld);
        else r[p] = *(const f32x4*)(base + (size_t)((NP == 4 ? 8 : 16) * p) * ld);
    }
}

// The same loads through a buffer descriptor: the tile origin is the descriptor's base (scalar), the thread's place inside the tile
// one register that never changes, the K-tile / pass advance a scalar offset -- no vector instruction in the K loop computes an address
// (a vector instruction of a wave whose neighbours keep the matrix pipe full waits for an MFMA slot: see gemm_out_fast).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool SRC_KMAJOR, int NP>
__device__ __forceinline__ void g2r_buf(f32x4 (&r)[4], __amdgpu_buffer_rsrc_t rs, int voff, int soff, int ld4) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int step = (!SRC_KMAJOR ? 32 : (NP == 4 ? 8 : 16)) * p * ld4;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + step, 0);
        r[p][0] = __uint_as_float(v[0]); r[p][1] = __uint_as_float(v[1]); r[p][2] = __uint_as_float(v[2]); r[p][3] = __uint_as_float(v[3]);
    }
}

template <bool SRC_KMAJOR, int NP>
__device__ __forceinline__ void r2s(const f32x4 (&r)[4], float* __restrict__ s, int tid) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (!SRC_KMAJOR) {
            int row = (tid >> 3) + 32 * p;
            int k = (tid & 7) * 4;
            *(f32x4*)(s + row * SROW + k) = r[p];
        } else {
            int k = NP == 4 ? (tid >> 5) + 8 * p : (tid >> 4) + 16 * p;
            int row = (NP == 4 ? (tid & 31) : (tid & 15)) * 4;
            *(f32x4*)(s + k * SKMJ + row) = r[p];
        }
    }
}

// ---- lean output pass of the interior-tile fp32 kernels -------------------------------------------------------------------------
// Measured (tools/gemm_probe.py, GP_TIMELINE): while the other workgroups of a CU keep the matrix pipe full, every vector instruction
// of a workgroup that has left its K loop waits about one MFMA issue slot (64+ clocks).  The generic epilogue below spends 10+ vector
// and branch instructions per output element (64-bit address arithmetic, per-element predicates): 6-21 us per workgroup in which its
// slot does no matrix work (1.5 us when it runs alone).  Here an element costs ONE buffer store (+ one add per bias term, two FMAs for
// the BatchNorm sums): the wave's 32 x 64 (64 x 64) region gets a buffer descriptor whose base is the region's first element (scalar
// arithmetic), the lane's offset inside it is one register, the row of accumulator register r is a scalar offset.
// C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#define MLSP_BUF_FLAGS 0x00020000          // raw dword buffer, gfx94x / gfx950 (DATA_FORMAT = 32 bit)
template <int WM, bool GB, bool ACC, bool ST>
__device__ __forceinline__ void gemm_out_fast(const GemmArgs& p, f32x16 (&acc)[2][2], float* Cw, const float (&bv)[2], const float (&gv)[2],
                                              int l31, int h, float (&cs)[2], float (&cq)[2]) {
    const int ldc4 = p.ldc * 4;                                            // bytes per row (scalar)
    const int voff = 4 * h * ldc4 + 4 * l31;                               // this lane inside the wave's region
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Cw, 0, 0x7ffffff0, MLSP_BUF_FLAGS);
    const bool store = Cw != nullptr;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float old[16];
            if (ACC) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    old[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (i * 32 + (r & 3) + 8 * (r >> 2)) * ldc4 + j * 128, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + bv[j];
                if (GB) v += gv[j];
                if (ACC) v += old[r];
                if (ST) { cs[j] += v; cq[j] = fmaf(v, v, cq[j]); }
                acc[i][j][r] = v;
            }
            if (store) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                {   // (a __builtin_bit_cast of the vector ELEMENT is miscompiled by this clang: every store took element 0)
                    const float vv = acc[i][j][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vv), rs, voff, (i * 32 + (r & 3) + 8 * (r >> 2)) * ldc4 + j * 128, 0);
                }
            }
        }
}

// ---- output pass of a dgrad with the previous layer's BatchNorm-backward reduction fused in (GemmArgs bs_*) --------------------------
// Per element: d = dz; dropout (byte of the quad's hash: the lanes of an aligned column quad share it -- each computes the hashes of four
// of its sixteen rows and takes the others from its neighbours with a DPP quad broadcast); d *= slope where the activation's argument
// a = y * scale + shift is not positive -- the same expressions, in the same order, as multi_dz_prime / dz_prime_q of the streaming
// passes this replaces, so the stored d' is bit-identical; cs += d', cq += d' * (y - mean) * invstd in fp32 over the wave's rows, fp64
// across waves and panels (the streaming reduction summed in fp64 throughout: the sums agree to ~1e-7 relative).
template <int WM>
__device__ __forceinline__ void gemm_out_bs(const GemmArgs& p, f32x16 (&acc)[2][2], float* Cw, const float* Yw, int colw, int row0, int l31, int h,
                                            float (&cs)[2], float (&cq)[2], float (&cm)[2]) {
    const int ldc4 = p.ldc * 4, ldy4 = p.bs_ldy * 4;
    const int voff = 4 * h * ldc4 + 4 * l31, voffy = 4 * h * ldy4 + 4 * l31;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Cw, 0, 0x7ffffff0, MLSP_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)Yw, 0, 0x7ffffff0, MLSP_BUF_FLAGS);
    const float slope = p.bs_slope, ik = p.bs_ik;
    const uint32_t th = p.bs_thresh, xH = p.bs_xH;
    const int sh8 = 8 * (l31 & 3);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = colw + j * 32 + l31;                                // column of C (launch-wide)
        const float sc = p.bs_bn[col], sf = p.bs_bn[p.bs_bnld + col], mu = p.bs_bn[2 * p.bs_bnld + col], is = p.bs_bn[3 * p.bs_bnld + col];
        const uint32_t qcol = (uint32_t)(p.bs_col + col) >> 2;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            float yv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                yv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, voffy, (i * 32 + (r & 3) + 8 * (r >> 2)) * ldy4 + j * 128, 0));
            uint32_t hs[4] = {0u, 0u, 0u, 0u};
            if (th) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    hs[t] = mix32(((uint32_t)(row0 + i * 32 + (l31 & 3) + 8 * t + 4 * h) * (uint32_t)p.bs_ld4 + qcol) ^ xH);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float d = acc[i][j][r];
                if (th) {
                    const uint32_t own = hs[r >> 2];
                    const uint32_t hq = (r & 3) == 0 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0x00, 0xf, 0xf, false)
                                      : (r & 3) == 1 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0x55, 0xf, 0xf, false)
                                      : (r & 3) == 2 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xaa, 0xf, 0xf, false)
                                                     : (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xff, 0xf, 0xf, false);
                    d = ((hq >> sh8) & 255u) >= th ? d * ik : 0.f;
                }
                const float a = fmaf(yv[r], sc, sf);
                if (!(a > 0.f)) d *= slope;
                cs[j] += d;
                cq[j] = fmaf(d, (yv[r] - mu) * is, cq[j]);
                cm[j] = fmaxf(cm[j], fabsf(d));
                acc[i][j][r] = d;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float vv = acc[i][j][r];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vv), rs, voff, (i * 32 + (r & 3) + 8 * (r >> 2)) * ldc4 + j * 128, 0);
            }
        }
    }
}

// ---- epilogue shared by the fp32 and the bf16-operand kernels --------------------------------------------------------
// acc: this wave's WM x 2 MFMA tiles of the block tile at (m0, n0); smem: the operand tiles, dead by now (scratch).
template <int WM, bool FAST, bool CBF = false, bool BSOK = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[2][2], float* smem, int tm, int m0, int n0, int split,
                                              int tid, int l31, int h, int wm, int wn) {
    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cout = p.C + (p.nsplit > 1 ? (size_t)split * p.M * p.ldc : 0);
    __bf16* Cb = (__bf16*)p.C;                            // CBF: bf16 output (never split: the slab is fp32)
    const bool epi = (p.nsplit == 1);
    float cs[2] = {0.f, 0.f}, cq[2] = {0.f, 0.f};     // column sums of this wave's 64 rows (BN statistics)
    float cm[2] = {0.f, 0.f};                         // column maxima of |d'| (gemm_out_bs only)
    // interior fp32 tiles whose rows share one per-cloud bias row: the lean output pass (launch_gemm sets p.fast_out)
    if (BSOK && p.bs_y) {               // (launch_gemm: interior tiles, one K pass, no bias / beta; statistics rows in stat_part)
        float* Cw = Cout + (size_t)(m0 + wm * (32 * WM)) * p.ldc + n0 + wn * 64;
        const float* Yw = p.bs_y + (size_t)(m0 + wm * (32 * WM)) * p.bs_ldy + n0 + wn * 64;
        gemm_out_bs<WM>(p, acc, Cw, Yw, n0 + wn * 64, m0 + wm * (32 * WM), l31, h, cs, cq, cm);
    } else if (FAST && !CBF && p.fast_out) {
        float bv[2] = {0.f, 0.f}, gv[2] = {0.f, 0.f};
        if (epi && p.bias) { bv[0] = p.bias[n0 + wn * 64 + l31]; bv[1] = p.bias[n0 + wn * 64 + 32 + l31]; }
        const bool gb = epi && p.gbias;
        if (gb) {
            const float* g = p.gbias + (size_t)(m0 / p.rows_per_group) * p.N + n0 + wn * 64 + l31;
            gv[0] = g[0]; gv[1] = g[32];
        }
        float* Cw = p.C ? Cout + (size_t)(m0 + wm * (32 * WM)) * p.ldc + n0 + wn * 64 : nullptr;
        if (epi && p.accumulate) gemm_out_fast<WM, false, true, false>(p, acc, Cw, bv, gv, l31, h, cs, cq);
        else if (gb) gemm_out_fast<WM, true, false, true>(p, acc, Cw, bv, gv, l31, h, cs, cq);
        else if (p.stat_part) gemm_out_fast<WM, false, false, true>(p, acc, Cw, bv, gv, l31, h, cs, cq);
        else gemm_out_fast<WM, false, false, false>(p, acc, Cw, bv, gv, l31, h, cs, cq);
    } else
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int col = n0 + wn * 64 + j * 32 + l31;
            if (!FAST && col >= p.N) continue;
            float bv = (epi && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * (32 * WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (FAST || row < p.M) {
                    float v = acc[i][j][r] + bv;
                    if (epi && p.gbias) v += p.gbias[(size_t)(row / p.rows_per_group) * p.N + col];
                    if (epi && p.accumulate) v += CBF ? (float)Cb[(size_t)row * p.ldc + col] : Cout[(size_t)row * p.ldc + col];
#ifdef GP_NOSTORE
                    if (p.C && v == 12345.678f) Cout[(size_t)row * p.ldc + col] = v;
#else
                    if (CBF) { if (p.C) Cb[(size_t)row * p.ldc + col] = (__bf16)v; }
                    else if (p.C) Cout[(size_t)row * p.ldc + col] = v;
#endif
                    cs[j] += v; cq[j] = fmaf(v, v, cq[j]);
                    acc[i][j][r] = v;
                }
            }
        }
    if (p.sel_gamma) {       // fused column extreme (max over points follows this layer): per panel, per column
        float bv[2]; int br[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l31;
            const bool use_max = col < p.N ? p.sel_gamma[col] >= 0.f : true;
            float best = use_max ? -INFINITY : INFINITY;
            int brow = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * (32 * WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float v = acc[i][j][r];
                    const bool take = row < p.M && (use_max ? (v > best) : (v < best));
                    best = take ? v : best; brow = take ? row : brow;
                }
            const float ob = __shfl_xor(best, 32, 64);
            const int orow = __shfl_xor(brow, 32, 64);
            const bool better = use_max ? (ob > best) : (ob < best);
            if (better || (ob == best && orow < brow)) { best = ob; brow = orow; }
            bv[j] = best; br[j] = brow;
        }
        __syncthreads();                                  // smem is reused below (and by the statistics block)
        float* sv = smem + 1024;                          // [wm][128] values, then rows
        int* sr = (int*)(smem + 1024 + 256);
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { sv[wm * 128 + wn * 64 + j * 32 + l31] = bv[j]; sr[wm * 128 + wn * 64 + j * 32 + l31] = br[j]; }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            const bool use_max = p.sel_gamma[n0 + tid] >= 0.f;
            float a = sv[tid], b2 = sv[128 + tid];
            int ra = sr[tid], rb = sr[128 + tid];
            const bool better = use_max ? (b2 > a) : (b2 < a);
            if (better || (b2 == a && rb < ra)) { a = b2; ra = rb; }
            p.sel_val[(size_t)tm * p.N + n0 + tid] = a;
            p.sel_row[(size_t)tm * p.N + n0 + tid] = ra;
        }
    }
    if (p.stat_part) {       // fused BatchNorm statistics: one fp64 partial per 128-row panel and column
        float* red = smem;   // [wm][sum|sq][128]  (the operand tiles are dead: the k-loop ended on a barrier)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            cs[j] += __shfl_xor(cs[j], 32, 64);
            cq[j] += __shfl_xor(cq[j], 32, 64);
            if (BSOK) cm[j] = fmaxf(cm[j], __shfl_xor(cm[j], 32, 64));
            if (h == 0) {
                red[(wm * 2 + 0) * 128 + wn * 64 + j * 32 + l31] = cs[j];
                red[(wm * 2 + 1) * 128 + wn * 64 + j * 32 + l31] = cq[j];
                if (BSOK) red[(4 + wm) * 128 + wn * 64 + j * 32 + l31] = cm[j];
            }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            p.stat_part[((size_t)tm * 2 + 0) * p.stat_ld + n0 + tid] = (double)red[0 * 128 + tid] + (double)red[2 * 128 + tid];
            p.stat_part[((size_t)tm * 2 + 1) * p.stat_ld + n0 + tid] = (double)red[1 * 128 + tid] + (double)red[3 * 128 + tid];
            if (BSOK && p.bs_y && p.bs_amax) p.bs_amax[(size_t)tm * p.stat_ld + n0 + tid] = fmaxf(red[4 * 128 + tid], red[5 * 128 + tid]);
        }
    }
}

// WM = 32-row MFMA tiles per wave along M: 2 -> 128x128 block tile; 1 -> 64x128 (twice the workgroups, for launches whose
// 128-row grid is too small to keep ~3 workgroups per CU in flight and out of phase)
// one staged f32x4 of an operand under the transform: c0 = first channel of the quad (channels contiguous when XF == 1 / 2 alike),
// e0 = element index (row * x_ld + c0) of its first value
__device__ __forceinline__ f32x4 xf_quad(const GemmArgs& p, f32x4 v, const f32x4& xs, const f32x4& xh, uint64_t e0) {
    // x_slope here is the EFFECTIVE negative-side factor in [0, 1] (0 for ReLU, 1 for no activation: launch_gemm), so the activation
    // is one max: max(a, a * s) == (a > 0 ? a : a * s).  The dropout rescale is applied exactly as bn_act_fwd does (a * inv_keep).
    const uint32_t hq = p.x_thresh ? dropout_hash4(p.x_seed, e0 >> 2) : 0u;  // e0 is a multiple of 4: one hash for the quad
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a = fmaf(v[e], xs[e], xh[e]);
        a = fmaxf(a, a * p.x_slope);
        if (p.x_thresh) a = ((hq >> (8 * e)) & 255u) >= p.x_thresh ? a * p.x_inv_keep : 0.f;
        v[e] = a;
    }
    return v;
}

template <bool TA, bool TB, int WM, bool FAST, int XF = 0, int GRP = 0>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs p) {
    constexpr int BMT = 64 * WM, NPA = 2 * WM;
    // A tile is k-major in LDS either way; its GLOBAL source is k-major iff TA.  B's source is
    // k-major iff !TB.
#ifdef GP_LDSPAD
    __shared__ __attribute__((aligned(16))) float smem[BM * SROW * 2 + GP_LDSPAD];
#else
    __shared__ __attribute__((aligned(16))) float smem[BM * SROW * 2];
#endif
    float* As = smem;
    float* Bs = smem + BM * SROW;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the output pass addresses with it
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile mapping (only when there are >= 8 row panels): all ntn column tiles of a row
    // panel share blockIdx.x % 8, i.e. one XCD's L2 under round-robin dispatch.  With fewer panels
    // (wgrad, skinny GEMMs) plain order keeps every XCD busy.
    const int bid = blockIdx.x;
    int tm, tn;
    if (p.xcd_map) {
        const int xcd = bid & 7, q = bid >> 3;
        tn = q % p.ntn;
        tm = (q / p.ntn) * 8 + xcd;
        if (tm >= p.ntm) return;
    } else {
        tn = bid % p.ntn;
        tm = bid / p.ntn;
    }
    const int split = blockIdx.y;
    const int m0 = tm * BMT, n0 = tn * BN;
    const int kbeg = split * p.ksplit;
    const int kend = min(p.K, kbeg + p.ksplit);
    // operand origins of this tile (groups: see GemmArgs); n0b = the tile's first column inside its group's B operand
    const float* Ap = p.A;
    const float* Bp = p.B;
    int n0b = n0;
    static_assert(GRP == 0 || (FAST && XF == 0), "groups: interior-tile kernel without operand transform");
    if (GRP == 1) {
        const int g = tn / p.gtiles;
        Ap += (size_t)g * p.a_gs;
        Bp = g == 0 ? p.Bg[0] : g == 1 ? p.Bg[1] : g == 2 ? p.Bg[2] : p.Bg[3];
        n0b = (tn - g * p.gtiles) * BN;
    } else if (GRP == 2) {
        Bp += (size_t)(tm / p.gtiles) * p.b_gs;
    }

#ifdef GP_TIMELINE
    const long long tl0 = wall_clock64();
    long long tl1 = 0;
#endif
#ifdef GP_DEPHASE
    {   // co-resident workgroups get distinct issue priorities: they drift out of phase, one's C stores run under another's MFMAs
#if GP_DEPHASE == 1
        const unsigned slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 3u;        // HW_ID.wave_id[3:0] & 3
#elif GP_DEPHASE == 2
        const unsigned slot = ((unsigned)bid >> 3) & 3u;
#else
        const unsigned slot = ((unsigned)bid >> 8) & 3u;
#endif
        if (slot == 0) __builtin_amdgcn_s_setprio(0);
        else if (slot == 1) __builtin_amdgcn_s_setprio(1);
        else if (slot == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    }
#endif
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[4], rb[4];
    // XF (FAST, fp32 only): scale / shift of this thread's channel quad.  XF == 1: the k-quad of the A tile held in ra (reloaded with
    // every tile); XF == 2: the n-quad of B (fixed).
    f32x4 xs = {0.f, 0.f, 0.f, 0.f}, xh = {0.f, 0.f, 0.f, 0.f};
    static_assert(XF == 0 || FAST, "operand transform: interior tiles only");
    static_assert(XF != 1 || !TA, "XF == 1: A row-major");
    static_assert(XF != 2 || !TB, "XF == 2: B k-major");
    // FAST: buffer descriptors based at the tile's first element of this K range; voa / vob = this thread's byte offset inside the tile
    __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, MLSP_BUF_FLAGS), rsb = rsa;
    int voa = 0, vob = 0, soa = 0, sob = 0;
    const int lda4 = p.lda * 4, ldb4 = p.ldb * 4;
    if (FAST) {
        rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(TA ? Ap + (size_t)kbeg * p.lda + m0 : Ap + (size_t)m0 * p.lda + kbeg), 0, 0x7ffffff0, MLSP_BUF_FLAGS);
        rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(!TB ? Bp + (size_t)kbeg * p.ldb + n0b : Bp + (size_t)n0b * p.ldb + kbeg), 0, 0x7ffffff0, MLSP_BUF_FLAGS);
        voa = TA ? (NPA == 4 ? (tid >> 5) : (tid >> 4)) * lda4 + (NPA == 4 ? (tid & 31) : (tid & 15)) * 16 : (tid >> 3) * lda4 + (tid & 7) * 16;
        vob = !TB ? (tid >> 5) * ldb4 + (tid & 31) * 16 : (tid >> 3) * ldb4 + (tid & 7) * 16;
        g2r_buf<TA, NPA>(ra, rsa, voa, soa, lda4);
        g2r_buf<!TB, 4>(rb, rsb, vob, sob, ldb4);
        if (XF == 1) { xs = *(const f32x4*)(p.x_scale + kbeg + (tid & 7) * 4); xh = *(const f32x4*)(p.x_shift + kbeg + (tid & 7) * 4); }
        if (XF == 2) { xs = *(const f32x4*)(p.x_scale + n0 + (tid & 31) * 4); xh = *(const f32x4*)(p.x_shift + n0 + (tid & 31) * 4); }
    } else {
        g2r<TA, NPA>(ra, p.A, p.lda, m0, p.M, kbeg, kend, p.a_vec, tid);
        g2r<!TB, 4>(rb, p.B, p.ldb, n0, p.N, kbeg, kend, p.b_vec, tid);
    }

    // XF: the staged registers of the tile at kt become the previous layer's activated output.  The first tile is transformed here;
    // every later one in the MIDDLE of the MFMA loop of the tile before it (its loads have landed by then and the vector work runs
    // under the matrix pipe instead of in front of the LDS writes, where all four waves would wait on it).
    auto xf_tile = [&](int kt) {
        if (XF == 1) {                                    // rows m0 + (tid >> 3) + 32 q, channels kt + 4 (tid & 7) ..
#pragma unroll
            for (int q = 0; q < NPA; ++q)
                ra[q] = xf_quad(p, ra[q], xs, xh, (uint64_t)(m0 + (tid >> 3) + 32 * q) * p.x_ld + p.x_col + kt + (tid & 7) * 4);
        }
        if (XF == 2) {                                    // rows (points) kt + (tid >> 5) + 8 q, channels n0 + 4 (tid & 31) ..
#pragma unroll
            for (int q = 0; q < 4; ++q)
                rb[q] = xf_quad(p, rb[q], xs, xh, (uint64_t)(kt + (tid >> 5) + 8 * q) * p.x_ld + p.x_col + n0 + (tid & 31) * 4);
        }
    };
    if (XF) xf_tile(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#ifdef GP_NOR2S
        if (k0 == kbeg)
#endif
        {
        r2s<TA, NPA>(ra, As, tid);
        r2s<!TB, 4>(rb, Bs, tid);
        }
        __syncthreads();
#ifdef GP_TIMELINE
        if (k0 == kbeg) tl1 = wall_clock64();
#endif
#ifdef GP_NOGLOBAL
        if (false) {
#else
        if (k0 + BK < kend) {
#endif
            if (FAST) {
                soa += TA ? BK * lda4 : BK * 4;
                sob += !TB ? BK * ldb4 : BK * 4;
                g2r_buf<TA, NPA>(ra, rsa, voa, soa, lda4);
                g2r_buf<!TB, 4>(rb, rsb, vob, sob, ldb4);
                if (XF == 1) {
                    xs = *(const f32x4*)(p.x_scale + k0 + BK + (tid & 7) * 4); xh = *(const f32x4*)(p.x_shift + k0 + BK + (tid & 7) * 4);
                }
            } else {
                g2r<TA, NPA>(ra, p.A, p.lda, m0, p.M, k0 + BK, kend, p.a_vec, tid);
                g2r<!TB, 4>(rb, p.B, p.ldb, n0, p.N, k0 + BK, kend, p.b_vec, tid);
            }
        }
        // Fragment reads.  K is consumed in groups of 4: MFMA step 2m takes k = 4m + 2h, step 2m+1 takes k = 4m + 2h + 1
        // (h = lane >> 5), so a row-major image gives each lane its two values with ONE 8-byte read; a k-major image is
        // read per value.  Both operands use the same k assignment, so every (TA, TB) combination is consistent.
        const int arow = wm * (32 * WM) + l31, bcol = wn * 64 + l31;
#ifdef GP_PRIO
        __builtin_amdgcn_s_setprio(GP_PRIO);
#endif
#pragma unroll
        for (int m = 0; m < BK / 4; ++m) {
#ifdef GP_IGLP
            __builtin_amdgcn_iglp_opt(GP_IGLP);
#endif
            if (XF && m == BK / 8 && k0 + BK < kend) xf_tile(k0 + BK);
            const int kq = 4 * m + 2 * h;
            float a0s0, a0s1, a1s0, a1s1, b0s0, b0s1, b1s0, b1s1;
            a1s0 = 0.f; a1s1 = 0.f;
            if (!TA) {
                const float2 t0 = *(const float2*)(As + arow * SROW + kq);
                a0s0 = t0.x; a0s1 = t0.y;
                if (WM == 2) { const float2 t1 = *(const float2*)(As + (arow + 32) * SROW + kq); a1s0 = t1.x; a1s1 = t1.y; }
            } else {
                a0s0 = As[kq * SKMJ + arow]; a0s1 = As[(kq + 1) * SKMJ + arow];
                if (WM == 2) { a1s0 = As[kq * SKMJ + arow + 32]; a1s1 = As[(kq + 1) * SKMJ + arow + 32]; }
            }
            if (TB) {
                const float2 t0 = *(const float2*)(Bs + bcol * SROW + kq), t1 = *(const float2*)(Bs + (bcol + 32) * SROW + kq);
                b0s0 = t0.x; b0s1 = t0.y; b1s0 = t1.x; b1s1 = t1.y;
            } else {
                b0s0 = Bs[kq * SKMJ + bcol]; b0s1 = Bs[(kq + 1) * SKMJ + bcol];
                b1s0 = Bs[kq * SKMJ + bcol + 32]; b1s1 = Bs[(kq + 1) * SKMJ + bcol + 32];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0s0, b0s0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0s0, b1s0, acc[0][1], 0, 0, 0);
            if (WM == 2) {
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1s0, b0s0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1s0, b1s0, acc[1][1], 0, 0, 0);
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0s1, b0s1, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0s1, b1s1, acc[0][1], 0, 0, 0);
            if (WM == 2) {
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1s1, b0s1, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1s1, b1s1, acc[1][1], 0, 0, 0);
            }
        }
#ifdef GP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#ifndef GP_NOBAR
        __syncthreads();
#endif
    }

#ifdef GP_TIMELINE
    const long long tl2 = wall_clock64();
#endif
    gemm_epilogue<WM, FAST>(p, acc, smem, tm, m0, n0, split, tid, l31, h, wm, wn);
#ifdef GP_TIMELINE
    __syncthreads();
    if (tid == 0) {                                    // 100 MHz ticks: start, first tile staged, loop end, epilogue end
        const long long tl3 = wall_clock64();
        float* dbg = p.C + (size_t)m0 * p.ldc + n0;
        dbg[0] = (float)(tl0 & 0xffffff); dbg[1] = (float)(tl1 - tl0); dbg[2] = (float)(tl2 - tl1); dbg[3] = (float)(tl3 - tl2);
        dbg[4] = (float)(__builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4));      // HW_ID[15:0]: wave, simd, pipe, cu, sh, se
        dbg[5] = (float)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20));      // XCC_ID
    }
#endif
}

// ---- N = 64 variant: 128 x 64 x 32 tiles, the four waves stacked along M (32 rows x 64 columns each) -----------------------------
// The dgrads into a 64-channel input (dx = duv * Wd of EdgeConv 2 / 3) and their wgrads have N = 64: on the 128-column tile half of
// every B stage and half of the MFMA columns are padding (17-31 TF).  Interior shapes only (M % 128 == 0, K-range % 32 == 0, 16-byte
// aligned operands), plain / split-K output; epilogue options: bias, BatchNorm statistics per 128-row panel.
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_f32_n64_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[BM * SROW + 64 * SROW + 32 * 64];
    float* As = smem;                              // row-major image [128][36] or k-major [32][128]
    float* Bs = smem + BM * SROW;                  // row-major image [64][36]  or k-major [32][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tm = blockIdx.x, split = blockIdx.y;
    const int m0 = tm * 128;
    const int kbeg = split * p.ksplit, kend = min(p.K, kbeg + p.ksplit);
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f32x4 ra[4], rb[4];
    const float* pa = TA ? p.A + (size_t)(kbeg + (tid >> 5)) * p.lda + m0 + (tid & 31) * 4
                         : p.A + (size_t)(m0 + (tid >> 3)) * p.lda + kbeg + (tid & 7) * 4;
    const float* pb = !TB ? p.B + (size_t)(kbeg + (tid >> 4)) * p.ldb + (tid & 15) * 4
                          : p.B + (size_t)(tid >> 3) * p.ldb + kbeg + (tid & 7) * 4;
    g2r_fast<TA, 4>(ra, pa, p.lda);
    g2r_fast<!TB, 2>(rb, pb, p.ldb);
    constexpr int BSK = 64;                        // k-major B image stride
    const int arow = wave * 32 + l31;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        r2s<TA, 4>(ra, As, tid);
        if (TB) r2s<false, 2>(rb, Bs, tid);
        else {
#pragma unroll
            for (int q = 0; q < 2; ++q) *(f32x4*)(Bs + ((tid >> 4) + 16 * q) * BSK + (tid & 15) * 4) = rb[q];
        }
        __syncthreads();
        if (k0 + BK < kend) {
            pa += TA ? (size_t)BK * p.lda : BK;
            pb += !TB ? (size_t)BK * p.ldb : BK;
            g2r_fast<TA, 4>(ra, pa, p.lda);
            g2r_fast<!TB, 2>(rb, pb, p.ldb);
        }
#pragma unroll
        for (int m = 0; m < BK / 4; ++m) {
            const int kq = 4 * m + 2 * h;
            float a0, a1, b00, b01, b10, b11;
            if (!TA) { const float2 t = *(const float2*)(As + arow * SROW + kq); a0 = t.x; a1 = t.y; }
            else { a0 = As[kq * SKMJ + arow]; a1 = As[(kq + 1) * SKMJ + arow]; }
            if (TB) {
                const float2 t0 = *(const float2*)(Bs + l31 * SROW + kq), t1 = *(const float2*)(Bs + (l31 + 32) * SROW + kq);
                b00 = t0.x; b01 = t0.y; b10 = t1.x; b11 = t1.y;
            } else {
                b00 = Bs[kq * BSK + l31]; b01 = Bs[(kq + 1) * BSK + l31];
                b10 = Bs[kq * BSK + l31 + 32]; b11 = Bs[(kq + 1) * BSK + l31 + 32];
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b00, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b10, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b01, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b11, acc[1], 0, 0, 0);
        }
        __syncthreads();
    }
    float* Cout = p.C + (p.nsplit > 1 ? (size_t)split * p.M * p.ldc : 0);
    const bool epi = p.nsplit == 1;
    float bv[2] = {0.f, 0.f};
    if (epi && p.bias) { bv[0] = p.bias[l31]; bv[1] = p.bias[32 + l31]; }
    float cs[2] = {0.f, 0.f}, cq[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[j][r] + bv[j];
            if (epi && p.accumulate) v += Cout[(size_t)row * p.ldc + j * 32 + l31];       // beta = 1 (never with split K)
            Cout[(size_t)row * p.ldc + j * 32 + l31] = v;
            cs[j] += v; cq[j] = fmaf(v, v, cq[j]);
        }
    if (p.stat_part) {       // fused BatchNorm statistics of the 128-row panel (the 64-channel layers with a bias: PointNet / set abstraction)
        float* red = smem;   // [4 waves][sum | sq][64]  (the operand tiles are dead: the K loop ended on a barrier)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            cs[j] += __shfl_xor(cs[j], 32, 64);
            cq[j] += __shfl_xor(cq[j], 32, 64);
            if (h == 0) { red[(wave * 2 + 0) * 64 + j * 32 + l31] = cs[j]; red[(wave * 2 + 1) * 64 + j * 32 + l31] = cq[j]; }
        }
        __syncthreads();
        if (tid < 64) {
            p.stat_part[((size_t)tm * 2 + 0) * p.stat_ld + tid] = ((double)red[0 * 64 + tid] + (double)red[2 * 64 + tid]) + ((double)red[4 * 64 + tid] + (double)red[6 * 64 + tid]);
            p.stat_part[((size_t)tm * 2 + 1) * p.stat_ld + tid] = ((double)red[1 * 64 + tid] + (double)red[3 * 64 + tid]) + ((double)red[5 * 64 + tid] + (double)red[7 * 64 + tid]);
        }
    }
}

// ---- bf16-operand variant (opt-in: `precision` = MLSP_PREC_BF16 of the calling entry point; BASELINE.json configs[4]) ------
// Same tiling, epilogues and split-K protocol as gemm_f32_kernel<..., FAST>, but the fp32 operands are rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32) on their way into LDS and multiplied by v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 8x fewer MFMA
// issues per K-tile and half the LDS traffic.  Both LDS images are row-major [row][32 + 8] bf16 (80-byte pitch: every
// fragment is one 16-byte read) for row-major global sources; k-major sources keep their layout in LDS ([k][rows], 320-byte pitch) and
// their fragments are read with ds_read_b64_tr_b16 (round 3: the transposing 2-byte stores they replace were 16-way bank conflicts --
// the dgrad / wgrad launches of configs[4] ran at a quarter of the forward's rate).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define BROW 40    // bf16 elements per LDS row

// LDS images of a 128 x 32 bf16 operand tile, shared by gemm_bf16_kernel and gemm_split_kernel:
//   global source row-major [rows][K]: image [row][32 k + 8 pad] (80-byte pitch), fragment = one ds_read_b128;
//   global source k-major  [K][rows]: image [k][128 rows + 32 pad] (320-byte pitch) written as it is loaded (8 / 16-byte pieces of a k-row),
//   fragment = two ds_read_b64_tr_b16 (the four k-rows of a transposed read fall in four different 64-byte bank windows).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define SX_PLANE 10240            // bytes of one bf16 image of a 128 x 32 operand tile: [128 rows][80 B] or [32 k][320 B]
#define SX_RPITCH 80
#define SX_KPITCH 320
#define SX_LDS __attribute__((address_space(3)))
#define SX_XF_KMAX 1024          // channels (K range of one workgroup) of an A-side operand transform in gemm_split_kernel

// this lane's byte offset of a fragment inside an image (rows rb .. rb+31 of the tile start at + rb * (KMAJ ? 2 : SX_RPITCH))
template <bool KMAJ>
__device__ __forceinline__ int sx_frag_base(int lane) {
    if (!KMAJ) return (lane & 31) * SX_RPITCH + 16 * (lane >> 5);
    const int i = lane & 15;            // ds_read_b64_tr_b16: lane 4q + p of a 16-lane group supplies row (= k) q, columns 4p .. 4p+3
    return (8 * (lane >> 5) + (i >> 2)) * SX_KPITCH + (16 * ((lane >> 4) & 1) + 4 * (i & 3)) * 2;
}
// fragment of the k16 step s2: rows = this lane's row of the 32-row window, k = 16 s2 + 8 h .. + 7
template <bool KMAJ>
__device__ __forceinline__ bf16x8 sx_frag(const char* img_at_window, int s2) {
    if (!KMAJ) return *(const bf16x8*)(img_at_window + 32 * s2);
    const char* a = img_at_window + 16 * s2 * SX_KPITCH;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SX_LDS bf16x4*)(a));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((SX_LDS bf16x4*)(a + 4 * SX_KPITCH));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}


template <bool SRC_KMAJOR, int NP>
__device__ __forceinline__ void r2s_bf16(const f32x4 (&r)[4], __bf16* __restrict__ s, int tid) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (!SRC_KMAJOR) {
            const int row = (tid >> 3) + 32 * p, k = (tid & 7) * 4;
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (__bf16)r[p][e];
            *(bf16x4*)(s + row * BROW + k) = v;
        } else {
            const int k = NP == 4 ? (tid >> 5) + 8 * p : (tid >> 4) + 16 * p;
            const int row = (NP == 4 ? (tid & 31) : (tid & 15)) * 4;
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (__bf16)r[p][e];
            *(bf16x4*)((char*)s + k * SX_KPITCH + row * 2) = v;          // k-major image: no transpose on the write
        }
    }
}

// Operand staging of the bf16-MFMA kernel for a 128x32 (or 64x32) tile.  fp32 sources are rounded to bf16 on the LDS write
// (operand mode); bf16 sources (activation storage) are copied: 16-byte global loads of 8 elements either way they lie.
//   row-major bf16 source: thread t -> rows (t>>2) + 64p, k = (t&3)*8        one 16-byte LDS write
//   k-major  bf16 source: thread t -> k = (t>>4) + 16p, rows (t&15)*8 ..+7    eight 2-byte LDS writes (transpose)
struct BfStage { bf16x8 v[2]; };
template <bool SRC_KMAJOR, int ROWS>
__device__ __forceinline__ void g2r_b16(BfStage& r, const __bf16* __restrict__ base, int ld) {
    constexpr int NPB = SRC_KMAJOR ? 2 : ROWS / 64;
#pragma unroll
    for (int p = 0; p < NPB; ++p) r.v[p] = *(const bf16x8*)(base + (size_t)((SRC_KMAJOR ? 16 : 64) * p) * ld);
}
template <bool SRC_KMAJOR, int ROWS>
__device__ __forceinline__ void r2s_b16(const BfStage& r, __bf16* __restrict__ s, int tid) {
    constexpr int NPB = SRC_KMAJOR ? 2 : ROWS / 64;
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
        if (!SRC_KMAJOR) {
            *(bf16x8*)(s + ((tid >> 2) + 64 * p) * BROW + (tid & 3) * 8) = r.v[p];
        } else {
            const int k = (tid >> 4) + 16 * p, row = (tid & 15) * 8;
            if (ROWS == 128 || row < ROWS) *(bf16x8*)((char*)s + k * SX_KPITCH + row * 2) = r.v[p];
        }
    }
}

// AB16 / BB16: the A / B operand is stored as bf16 in HBM; CB16: C is written as bf16.
// NEDGE (only with a k-major fp32 B, i.e. the dgrad dX = dY * W): N is a multiple of 4 but not of the 128-column tile (the 192-channel
// input of the PointSegDA heads): B columns beyond N are zero-filled on the load and the epilogue predicates its stores.
template <bool TA, bool TB, int WM, bool AB16, bool BB16, bool CB16, bool NEDGE = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
    constexpr int BMT = 64 * WM, NPA = 2 * WM;
    __shared__ __attribute__((aligned(16))) float smem[BM * SROW * 2];      // same footprint as the fp32 kernel (epilogue scratch)
    __bf16* As = (__bf16*)smem;
    __bf16* Bs = As + BM * BROW;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = blockIdx.x;
    int tm, tn;
    if (p.xcd_map) {
        const int xcd = bid & 7, q = bid >> 3;
        tn = q % p.ntn;
        tm = (q / p.ntn) * 8 + xcd;
        if (tm >= p.ntm) return;
    } else {
        tn = bid % p.ntn;
        tm = bid / p.ntn;
    }
    const int split = blockIdx.y;
    const int m0 = tm * BMT, n0 = tn * BN;
    const int kbeg = split * p.ksplit;
    const int kend = min(p.K, kbeg + p.ksplit);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[4], rb[4];
    BfStage sa, sb;
    const float* pa = nullptr; const float* pb = nullptr;
    const __bf16* qa = nullptr; const __bf16* qb = nullptr;
    if (AB16) {
        const __bf16* A16 = (const __bf16*)p.A;
        qa = TA ? A16 + (size_t)(kbeg + (tid >> 4)) * p.lda + m0 + (tid & 15) * 8
                : A16 + (size_t)(m0 + (tid >> 2)) * p.lda + kbeg + (tid & 3) * 8;
        if (!TA || BMT == 128 || (tid & 15) * 8 < BMT) g2r_b16<TA, BMT>(sa, qa, p.lda);
    } else {
        pa = TA ? p.A + (size_t)(kbeg + (NPA == 4 ? (tid >> 5) : (tid >> 4))) * p.lda + m0 + (NPA == 4 ? (tid & 31) : (tid & 15)) * 4
                : p.A + (size_t)(m0 + (tid >> 3)) * p.lda + kbeg + (tid & 7) * 4;
        g2r_fast<TA, NPA>(ra, pa, p.lda);
    }
    if (BB16) {
        const __bf16* B16 = (const __bf16*)p.B;
        qb = !TB ? B16 + (size_t)(kbeg + (tid >> 4)) * p.ldb + n0 + (tid & 15) * 8
                 : B16 + (size_t)(n0 + (tid >> 2)) * p.ldb + kbeg + (tid & 3) * 8;
        g2r_b16<!TB, 128>(sb, qb, p.ldb);
    } else {
        pb = !TB ? p.B + (size_t)(kbeg + (tid >> 5)) * p.ldb + n0 + (tid & 31) * 4
                 : p.B + (size_t)(n0 + (tid >> 3)) * p.ldb + kbeg + (tid & 7) * 4;
        if (!NEDGE || n0 + (tid & 31) * 4 < p.N) g2r_fast<!TB, 4>(rb, pb, p.ldb);
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) rb[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const bool b_lane_ok = !NEDGE || n0 + (tid & 31) * 4 < p.N;
    const bool a_lane_ok = !AB16 || !TA || BMT == 128 || (tid & 15) * 8 < BMT;     // 64-row k-major bf16 tiles use half the lanes
    const char* fa = (const char*)As + sx_frag_base<TA>(lane) + (wm * 32 * WM) * (TA ? 2 : SX_RPITCH);
    const char* fb = (const char*)Bs + sx_frag_base<!TB>(lane) + (wn * 64) * (!TB ? 2 : SX_RPITCH);
    constexpr int FWA = 32 * (TA ? 2 : SX_RPITCH), FWB = 32 * (!TB ? 2 : SX_RPITCH);               // next 32-row window
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        if (AB16) r2s_b16<TA, BMT>(sa, As, tid); else r2s_bf16<TA, NPA>(ra, As, tid);
        if (BB16) r2s_b16<!TB, 128>(sb, Bs, tid); else r2s_bf16<!TB, 4>(rb, Bs, tid);
        __syncthreads();
        if (k0 + BK < kend) {
            if (AB16) { qa += TA ? (size_t)BK * p.lda : BK; if (a_lane_ok) g2r_b16<TA, BMT>(sa, qa, p.lda); }
            else { pa += TA ? (size_t)BK * p.lda : BK; g2r_fast<TA, NPA>(ra, pa, p.lda); }
            if (BB16) { qb += !TB ? (size_t)BK * p.ldb : BK; g2r_b16<!TB, 128>(sb, qb, p.ldb); }
            else { pb += !TB ? (size_t)BK * p.ldb : BK; if (b_lane_ok) g2r_fast<!TB, 4>(rb, pb, p.ldb); }
        }
#pragma unroll
        for (int s2 = 0; s2 < BK / 16; ++s2) {                      // MFMA step: k = 16*s2 + 8*h .. +7
            const bf16x8 a0 = sx_frag<TA>(fa, s2);
            const bf16x8 b0 = sx_frag<!TB>(fb, s2), b1 = sx_frag<!TB>(fb + FWB, s2);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            if (WM == 2) {
                const bf16x8 a1 = sx_frag<TA>(fa + FWA, s2);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    gemm_epilogue<WM, !NEDGE, CB16>(p, acc, smem, tm, m0, n0, split, tid, l31, h, wm, wn);
}

// ---- fp32-accurate products on the bf16 matrix cores (`precision` = MLSP_PREC_BF16X6 of the calling entry point) ----------------------
// On gfx950 the f32 MFMA runs at 1/16 of the bf16 MFMA rate.  Every fp32 operand value is split, while its tile is staged, into three
// bf16 pieces x = a + b + c (a = bf16(x), b = bf16(x - a), c = bf16(x - a - b): 8 + 8 + 8 significand bits, RNE, the remainders are exact
// in fp32), and x y is taken as the six products a a' + (a b' + b a') + (a c' + c a' + b b') on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation.  Each piece product is exact (8 x 8 bits); the dropped terms (b c', c b', c c') are below 2^-25 |x y|: the result is as
// accurate as an fp32 fused-multiply-add chain (tests/test_gpu_kernels.py::test_gemm_split_bf16_accuracy measures both against
// float64) for 6/16 of the matrix time.  Interior-tile shapes with fp32 operands and output only; same epilogues as the fp32 kernel.
// NOT a k-ordered fmaf chain: the kNN kernels (bit-exact canonical distances) never use it.  Non-finite operands: an infinite (or > bf16-max,
// 3.39e38) operand value yields NaN where the f32 MFMA yields +-inf (inf - bf16(inf) is NaN); finite fp32 data -- everything this path
// feeds it -- is covered by the accuracy test.
//
// Structure (measured in tools/x6: 1.55-1.65x the fp32 kernel on the 32768-row layers; the variants that lost are listed there):
//   * one LDS buffer of six bf16 images (A: hi | mid | lo, B: hi | mid | lo), two workgroups per CU;
//   * an operand whose global source is row-major ([rows][K]) is imaged [row][32 k + 8 pad] and read with one ds_read_b128 per fragment;
//     one whose source is k-major ([K][rows]: the dgrad's W, both wgrad operands) is imaged [k][128 rows + 32 pad] -- written as it is
//     loaded (8-byte rows, no transposing 2-byte stores) and read with two ds_read_b64_tr_b16 per fragment; the 320-byte pitch puts the
//     four k-rows of a transposed read in four different 64-byte bank windows;
//   * per K-tile: all 24 fragments of tile t are read, barrier, then the 48 (24) MFMAs of tile t run with the split + image write of tile
//     t+1 and the global loads of tile t+2 hand-placed between them (gen_split_body.py -> gemm_split_body_wm*.inc), barrier.
__device__ __forceinline__ uint32_t sx_cvt_pk(float lo, float hi) {       // one v_cvt_pk_bf16_f32 (RNE)
    const f32x2 v = {lo, hi};
    const bf16x2 b = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, b);
}
// ---- two f16 pieces, three products (gemm_split_kernel<.., NPC = 2>; `precision` = MLSP_PREC_F16X3) ----------------------------------
// x s = h0 + h1 + e with h0 = f16(x s), h1 = f16(x s - h0) (RNE; the remainder is exact in fp32), |e| <= 2^-22 |x s|: 11 + 11 significand
// bits; x y ~ h0 h0' + (h0 h1' + h1 h0') on v_mfma_f32_32x32x16_f16 -- three MFMAs per product instead of six -- with fp32 accumulation;
// every piece product is exact (11 x 11 bits), the dropped term h1 h1' is < 2^-22 |x y|.  s is a per-OPERAND power of two (exact) that
// places the operand's largest magnitude in [2^14, 2^15) of the f16 range (max 65504): elements down to 2^-17 of the largest keep all 22
// bits, smaller ones an absolute error < 2^-40 of the largest (f16 subnormal spacing) -- nothing against the 2^-24 of every fp32
// accumulation.  The scale comes from a device-side bound of the operand's magnitude (GemmArgs a_amax / b_amax; launch_gemm computes it
// with one streaming pass when the caller has none), the accumulators are multiplied by 2^-(ea + eb) before the epilogue.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define SXH(v_) __builtin_bit_cast(f16x8, (v_))
__device__ __forceinline__ uint32_t sx_cvt_pk_h(float lo, float hi) {     // one v_cvt_pk_f16_f32 (RNE)
    const f32x2 v = {lo, hi};
    const f16x2 b = __builtin_convertvector(v, f16x2);
    return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ float sx_h_lo(uint32_t pk) { return (float)__builtin_bit_cast(f16x2, pk)[0]; }
__device__ __forceinline__ float sx_h_hi(uint32_t pk) { return (float)__builtin_bit_cast(f16x2, pk)[1]; }
// x * s - (low / high f16 half of pk) in ONE instruction (v_fma_mix_f32 reads the f16 operand as it lies in the packed register): the
// remainder of the first piece, exact in fp32 (s a power of two, pk's half = f16(x * s)).  The compiler's own lowering of the same
// expression is v_cvt_f32_f16 + v_add per element (tools/r6: 120 -> 88 vector instructions per K-tile).
__device__ __forceinline__ float sx_rem_lo(float x, float s, uint32_t pk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "s"(s), "v"(pk));
    return r;
}
__device__ __forceinline__ float sx_rem_hi(float x, float s, uint32_t pk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "s"(s), "v"(pk));
    return r;
}
__device__ __forceinline__ void sx_split_store_h(const f32x4& x, float s, char* d) {
    uint32_t pk[2][2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        pk[0][hh] = sx_cvt_pk_h(x[2 * hh] * s, x[2 * hh + 1] * s);
        pk[1][hh] = sx_cvt_pk_h(sx_rem_lo(x[2 * hh], s, pk[0][hh]), sx_rem_hi(x[2 * hh + 1], s, pk[0][hh]));
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) *(u32x2*)(d + q * SX_PLANE) = (u32x2){pk[q][0], pk[q][1]};
}
// biased exponent field of the power-of-two scale for an operand whose magnitudes are bounded by `bound`:
// bound in [2^e, 2^(e+1)) -> s = 2^(14 - e), clamped to [2^-62, 2^62]; zero / denormal / non-finite bound -> 1.0
__device__ __forceinline__ int sx_scale_exp(float bound) {
    const int e = (int)((__float_as_uint(bound) >> 23) & 255u);
    if (e == 0 || e == 255) return 127;
    return min(max(268 - e, 127 - 62), 127 + 62);
}
// three-way split of one staged quad into the three images (prologue only: the K loop uses the hand-placed stream)
__device__ __forceinline__ void sx_split_store(const f32x4& x, char* d) {
    uint32_t pk[3][2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const float x0 = x[2 * hh], x1 = x[2 * hh + 1];
        pk[0][hh] = sx_cvt_pk(x0, x1);
        const float r0 = x0 - __uint_as_float(pk[0][hh] << 16), r1 = x1 - __uint_as_float(pk[0][hh] & 0xffff0000u);
        pk[1][hh] = sx_cvt_pk(r0, r1);
        pk[2][hh] = sx_cvt_pk(r0 - __uint_as_float(pk[1][hh] << 16), r1 - __uint_as_float(pk[1][hh] & 0xffff0000u));
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) *(u32x2*)(d + q * SX_PLANE) = (u32x2){pk[q][0], pk[q][1]};
}
__device__ __forceinline__ f32x4 sx_bufload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
// XF (operand transform, GemmArgs x_*): the staged fp32 values of ONE operand are the previous layer's PRE-BatchNorm output and become its
// activated output -- act(x * scale[c] + shift[c]), then that layer's dropout (XD) -- right before they are split: the same arithmetic, in
// the same order, as the streaming pass it replaces (bn_act_fwd_vec_kernel / multi_act_fwd_kernel), so the product is bit-identical to the
// one over the materialised tensor.  XF == 1: A row-major [M][K], channel = k (forward of the consumer layer); XF == 2: B k-major [K][N],
// channel = n (the consumer's weight gradient, X^T side).  Both also in block-diagonal launches (the group's channels start at
// g * a_gs / g * b_gs).  The vector work rides in the split stream's slots (gen_split_body.py variants xa / xad / xb / xbd).
// DY (GemmArgs dy_*): the A operand is d' and the layer's BatchNorm backward is applied to it while it is staged (two loads per staged
// quad: d' and y), for the layer's dgrad (A row-major: coefficients per K-tile from LDS) and its weight gradient (A k-major: the thread's
// channel quad is fixed, coefficients in registers).
template <bool TA, bool TB, int WM, int XF = 0, bool XD = false, bool DY = false, int NPC = 3>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(GemmArgs p) {
    static_assert(NPC == 3 || NPC == 2, "three bf16 pieces (six products) or two f16 pieces (three products)");
    static_assert(XF == 0 || (XF == 1 && !TA) || (XF == 2 && !TB), "XF == 1: A row-major; XF == 2: B k-major");
    static_assert(XF != 0 || !XD, "dropout only with a transform");
    static_assert(!(DY && XF == 1), "one transform per operand");
    constexpr bool KA = TA, KB = !TB;                                 // operand's global source is k-major
    constexpr int BMT = 64 * WM, NQA = 2 * WM;                        // A quads (16-byte loads) of a tile per thread; B: 4
    __shared__ __attribute__((aligned(16))) char simg[2 * NPC * SX_PLANE];  // 61,440 B (NPC = 3) / 40,960 B
    constexpr int XFS = XF == 1 ? 2 * SX_XF_KMAX : (DY && !TA) ? 3 * SX_XF_KMAX : 4;
    __shared__ __attribute__((aligned(16))) float xfs[XFS];           // XF == 1: scale | shift of this workgroup's K range (<= SX_XF_KMAX channels); DY dgrad: c0 | nk2 | sc
    float* smem = (float*)simg;                                       // epilogue scratch (the images are dead by then)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = blockIdx.x;
    int tm, tn, split = blockIdx.y;
    if (p.xcd_map == 3) {
        // several row panels AND column tiles per K split (weight gradients of the wide layers; 1-D grid): ALL tiles of a split are
        // consecutive slots of ONE XCD -- they walk the same K range in step, so both the A range and the B range of a split come out of HBM
        // once (under the (split, panel) order below every B range was fetched by each of the ntm XCDs that held one of its panels:
        // 820 MB HBM-side for 335 MB of operands at 1024 x 512 x 32768, profiles/pmc_r6)
        const int xcd = bid & 7, q = bid >> 3, tps = p.ntm * p.ntn;
        const int t = q % tps;
        split = (q / tps) * 8 + xcd;
        tm = t / p.ntn; tn = t - tm * p.ntn;
    } else if (p.xcd_map == 2) {
        // few row panels, many K splits (weight gradients; 1-D grid): the ntn tiles that read the same A panel of the same K range are
        // consecutive slots of ONE XCD -- dispatched together, they walk their K range in step and the panel is fetched from HBM once
        // instead of once per XCD (the plain order spread them over ntn XCDs: 1.24 GB HBM-side for 335 MB of operands at 1024 x 512 x 32768)
        const int xcd = bid & 7, q = bid >> 3;
        tn = q % p.ntn;
        const int G = (q / p.ntn) * 8 + xcd;               // (split, row panel) pair
        tm = G % p.ntm;
        split = G / p.ntm;
    } else if (p.xcd_map) {
        const int xcd = bid & 7, q = bid >> 3;
        tn = q % p.ntn;
        tm = (q / p.ntn) * 8 + xcd;
        if (tm >= p.ntm) return;
    } else {
        tn = bid % p.ntn;
        tm = bid / p.ntn;
    }
    const int m0 = tm * BMT, n0 = tn * BN;
    const int kbeg = split * p.ksplit;
    const int kend = min(p.K, kbeg + p.ksplit);
    const int T = (kend - kbeg) / BK;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // operand origins of this tile (block-diagonal launches: GemmArgs groups, as gemm_f32_kernel<.., GRP>); n0b = first column inside the group's B
    const float* Ap = p.A;
    const float* Bp = p.B;
    int n0b = n0;
    if (p.gmode == 1) {
        const int g = tn / p.gtiles;
        Ap += (size_t)g * p.a_gs;
        Bp = g == 0 ? p.Bg[0] : g == 1 ? p.Bg[1] : g == 2 ? p.Bg[2] : p.Bg[3];
        n0b = (tn - g * p.gtiles) * BN;
    } else if (p.gmode == 2) {
        Bp += (size_t)(tm / p.gtiles) * p.b_gs;
    }
    // first channel of the transformed operand's group inside x_scale / x_shift (block-diagonal launches)
    const int xgc = XF == 1 ? (p.gmode == 1 ? (tn / p.gtiles) * (int)p.a_gs : 0) : XF == 2 ? (p.gmode == 2 ? (tm / p.gtiles) * (int)p.b_gs : 0) : 0;
    // global loads through buffer descriptors based at the tile's first element of this K range (scalar); voa / vob = this thread's byte
    // offset inside the tile (one register each, fixed); the K-tile advance and the quad step are scalar offsets: no vector instruction of
    // the K loop computes an address.  soa / sob are clamped at the last tile (the stream always loads "tile t+2").
    const int lda4 = p.lda * 4, ldb4 = p.ldb * 4;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(KA ? Ap + (size_t)kbeg * p.lda + m0 : Ap + (size_t)m0 * p.lda + kbeg), 0, 0x7ffffff0, MLSP_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(KB ? Bp + (size_t)kbeg * p.ldb + n0b : Bp + (size_t)n0b * p.ldb + kbeg), 0, 0x7ffffff0, MLSP_BUF_FLAGS);
    const int voa = KA ? (NQA == 4 ? (tid >> 5) : (tid >> 4)) * lda4 + (NQA == 4 ? (tid & 31) : (tid & 15)) * 16 : (tid >> 3) * lda4 + (tid & 7) * 16;
    const int vob = KB ? (tid >> 5) * ldb4 + (tid & 31) * 16 : (tid >> 3) * ldb4 + (tid & 7) * 16;
    const int sta = KA ? BK * lda4 : BK * 4, stb = KB ? BK * ldb4 : BK * 4;                       // one K-tile on
    const int qa_ = (KA ? (NQA == 4 ? 8 : 16) : 32) * lda4, qb_ = (KB ? 8 : 32) * ldb4;           // one quad on
    const int enda = (T - 1) * sta, endb = (T - 1) * stb;
    int soa = 0, sob = 0;
#define SX_LOAD_A(q) sx_bufload(rsa, voa, soa + (q) * qa_)
#define SX_LOAD_B(q) sx_bufload(rsb, vob, sob + (q) * qb_)
    // DY: the layer's pre-BN output at the coordinates of A (same pitch, same group offset)
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(DY ? (KA ? p.dy_y + (Ap - p.A) + (size_t)kbeg * p.lda + m0 : p.dy_y + (Ap - p.A) + (size_t)m0 * p.lda + kbeg) : Ap), 0, DY ? 0x7ffffff0 : 0, MLSP_BUF_FLAGS);
#define SX_LOAD_AY(q) sx_bufload(rsy, voa, soa + (q) * qa_)
    // this thread's place in the images
    char* wa = simg + (KA ? (NQA == 4 ? (tid >> 5) * SX_KPITCH + (tid & 31) * 8 : (tid >> 4) * SX_KPITCH + (tid & 15) * 8) : (tid >> 3) * SX_RPITCH + (tid & 7) * 8);
    char* wb = simg + NPC * SX_PLANE + (KB ? (tid >> 5) * SX_KPITCH + (tid & 31) * 8 : (tid >> 3) * SX_RPITCH + (tid & 7) * 8);
    constexpr int WQA = KA ? (NQA == 4 ? 8 : 16) * SX_KPITCH : 32 * SX_RPITCH, WQB = KB ? 8 * SX_KPITCH : 32 * SX_RPITCH;
    const char* fa = simg + sx_frag_base<KA>(lane) + (wm * 32 * WM) * (KA ? 2 : SX_RPITCH);
    const char* fb = simg + NPC * SX_PLANE + sx_frag_base<KB>(lane) + (wn * 64) * (KB ? 2 : SX_RPITCH);
    constexpr int FWA = 32 * (KA ? 2 : SX_RPITCH), FWB = 32 * (KB ? 2 : SX_RPITCH);               // next 32-row window
    int sxe_a = 127, sxe_b = 127;                                     // NPC == 2: the operands' power-of-two scales (exponent fields; uniform)
    float sxs_a = 1.f, sxs_b = 1.f;
    f32x4 raw[8];                                                     // [0, NQA) A quads, then 4 B quads of the tile being staged
    // ---- operand transform state (XF): xsc / xsh = scale / shift of this thread's channel quad of the tile being staged (XF == 1: reloaded
    // from LDS per K-tile; XF == 2: fixed), xq = index of the aligned element quad (row * x_ld + x_col + channel) / 4 of this thread's
    // quad 0 of that tile in the producer's matrix (one dropout hash per quad: common.h dropout_hash4), xqs / xqt = its step per staged
    // quad / per K-tile (uniform).  Host: x_ld % 4 == 0, x_col % 4 == 0, rows * x_ld < 2^34 (the hash's high word is launch-uniform).
    f32x4 xsc = {0.f, 0.f, 0.f, 0.f}, xsh = {0.f, 0.f, 0.f, 0.f};
    uint32_t xq = 0, xhq = 0, xH = 0;
    int xqs = 0, xqt = 0;
    const float xslope = p.x_slope, xik = p.x_inv_keep;
    const uint32_t xth = p.x_thresh;
    if (XD) xH = mix32((uint32_t)p.x_seed) ^ (uint32_t)(p.x_seed >> 32) * 0x9e3779b9U;
    if (XF == 1) {
        for (int i = tid * 4; i < T * BK; i += 1024) {
            *(f32x4*)(xfs + i) = *(const f32x4*)(p.x_scale + xgc + kbeg + i);
            *(f32x4*)(xfs + SX_XF_KMAX + i) = *(const f32x4*)(p.x_shift + xgc + kbeg + i);
        }
        __syncthreads();
        xsc = *(const f32x4*)(xfs + (tid & 7) * 4); xsh = *(const f32x4*)(xfs + SX_XF_KMAX + (tid & 7) * 4);
        xq = (uint32_t)(((uint64_t)(m0 + (tid >> 3)) * (uint64_t)p.x_ld + (uint64_t)(p.x_col + xgc + kbeg + (tid & 7) * 4)) >> 2);
        xqs = 8 * p.x_ld; xqt = BK / 4;
    } else if (XF == 2) {
        const int ch = xgc + n0b + (tid & 31) * 4;
        xsc = *(const f32x4*)(p.x_scale + ch); xsh = *(const f32x4*)(p.x_shift + ch);
        xq = (uint32_t)(((uint64_t)(kbeg + (tid >> 5)) * (uint64_t)p.x_ld + (uint64_t)(p.x_col + ch)) >> 2);
        xqs = 2 * p.x_ld; xqt = 8 * p.x_ld;
    }
    auto xf_pair = [&](f32x4& v, int hh) {                            // elements 2 hh, 2 hh + 1 of a staged quad
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 2 * hh + e;
            float a = fmaf(v[c], xsc[c], xsh[c]);
            a = fmaxf(a, a * xslope);                                 // x_slope in [0, 1]: == (a > 0 ? a : a * slope)
            if (XD) a = ((xhq >> (8 * c)) & 255u) >= xth ? a * xik : 0.f;
            v[c] = a;
        }
    };
#define SX_XF_HASH_A(q_) do { if constexpr (XF == 1 && XD) xhq = mix32((xq + (uint32_t)((q_) * xqs)) ^ xH); } while (0)
#define SX_XF_HASH_B(q_) do { if constexpr (XF == 2 && XD) xhq = mix32((xq + (uint32_t)((q_) * xqs)) ^ xH); } while (0)
#define SX_XF_A(q_, hh_) do { if constexpr (XF == 1) xf_pair(raw[q_], hh_); } while (0)
#define SX_XF_B(q_, hh_) do { if constexpr (XF == 2) xf_pair(raw[NQA + (q_)], hh_); } while (0)
    // after a tile's quads are transformed: on to the tile after it (the stream always stages "tile t+1"; past the end a harmless repeat)
#define SX_XF_NEXT(tnext_) do { if constexpr (XF != 0) { xq += (uint32_t)xqt; if constexpr (XF == 1) { const int tc_ = min((tnext_), T - 1) * BK + (tid & 7) * 4; \
        xsc = *(const f32x4*)(xfs + tc_); xsh = *(const f32x4*)(xfs + SX_XF_KMAX + tc_); } } } while (0)
    // ---- DY state: rawy = the y quads of the tile being staged; dc0 / dnk2 / dsc = the coefficient quads of this thread's channels
    f32x4 rawy[DY ? 4 : 1];                                  // (sized for WM = 2: both bodies are compiled into every instantiation)
    f32x4 dc0 = {0.f, 0.f, 0.f, 0.f}, dnk2 = {0.f, 0.f, 0.f, 0.f}, dsc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (DY) {
        if constexpr (KA) {             // weight gradient: channel = this thread's four tile rows, fixed
            const int ch = m0 + (NQA == 4 ? (tid & 31) : (tid & 15)) * 4;
            dc0 = *(const f32x4*)(p.dy_coef + ch); dnk2 = *(const f32x4*)(p.dy_coef + p.dy_cld + ch); dsc = *(const f32x4*)(p.dy_coef + 2 * p.dy_cld + ch);
        } else {                        // dgrad: channel = k: the K range's coefficients through LDS, one quad per K-tile
            const int gc0 = (p.gmode == 1 ? (tn / p.gtiles) * (int)p.a_gs : 0) + kbeg;
            for (int i = tid * 4; i < T * BK; i += 1024) {
                *(f32x4*)(xfs + i) = *(const f32x4*)(p.dy_coef + gc0 + i);
                *(f32x4*)(xfs + SX_XF_KMAX + i) = *(const f32x4*)(p.dy_coef + p.dy_cld + gc0 + i);
                *(f32x4*)(xfs + 2 * SX_XF_KMAX + i) = *(const f32x4*)(p.dy_coef + 2 * p.dy_cld + gc0 + i);
            }
            __syncthreads();
            dc0 = *(const f32x4*)(xfs + (tid & 7) * 4); dnk2 = *(const f32x4*)(xfs + SX_XF_KMAX + (tid & 7) * 4); dsc = *(const f32x4*)(xfs + 2 * SX_XF_KMAX + (tid & 7) * 4);
        }
    }
#define SX_DY_A(q_, hh_) do { if constexpr (DY) { \
        raw[q_][2 * (hh_)] = (raw[q_][2 * (hh_)] + fmaf(rawy[q_][2 * (hh_)], dnk2[2 * (hh_)], dc0[2 * (hh_)])) * dsc[2 * (hh_)]; \
        raw[q_][2 * (hh_) + 1] = (raw[q_][2 * (hh_) + 1] + fmaf(rawy[q_][2 * (hh_) + 1], dnk2[2 * (hh_) + 1], dc0[2 * (hh_) + 1])) * dsc[2 * (hh_) + 1]; } } while (0)
#define SX_DY_LOAD(q_) do { if constexpr (DY) rawy[q_] = SX_LOAD_AY(q_); } while (0)
#define SX_DY_NEXT(tnext_) do { if constexpr (DY && !KA) { const int tc_ = min((tnext_), T - 1) * BK + (tid & 7) * 4; \
        dc0 = *(const f32x4*)(xfs + tc_); dnk2 = *(const f32x4*)(xfs + SX_XF_KMAX + tc_); dsc = *(const f32x4*)(xfs + 2 * SX_XF_KMAX + tc_); } } while (0)
#pragma unroll
    for (int q = 0; q < NQA; ++q) { raw[q] = SX_LOAD_A(q); SX_DY_LOAD(q); }
#pragma unroll
    for (int q = 0; q < 4; ++q) raw[NQA + q] = SX_LOAD_B(q);
    if constexpr (NPC == 2) {
        // bounds of the two operands as the split sees them (see GemmArgs a_amax) -> per-workgroup power-of-two scales; the first tile's
        // loads above are in flight meanwhile
        __shared__ float sred[12];
        float va = 0.f, vb = 0.f, ca = 0.f;
        if (XF != 1) for (int i = tid; i < p.a_amax_n; i += 256) va = fmaxf(va, p.a_amax[i]);
        if (XF != 2) {
            const int gb = p.gmode == 1 ? tn / p.gtiles : 0;
            const float* bam = gb == 0 ? p.b_amax[0] : gb == 1 ? p.b_amax[1] : gb == 2 ? p.b_amax[2] : p.b_amax[3];
            const int bn_ = gb == 0 ? p.b_amax_n[0] : gb == 1 ? p.b_amax_n[1] : gb == 2 ? p.b_amax_n[2] : p.b_amax_n[3];
            for (int i = tid; i < bn_; i += 256) vb = fmaxf(vb, bam[i]);
        }
        const float sqr = p.stat_sqrt_rows;
        if constexpr (XF == 1) {
            for (int i = tid * 4; i < T * BK; i += 1024) {
                const f32x4 sc = *(const f32x4*)(xfs + i), sh = *(const f32x4*)(xfs + SX_XF_KMAX + i);
                const f32x4 mu = *(const f32x4*)(p.x_mean + xgc + kbeg + i), is = *(const f32x4*)(p.x_invstd + xgc + kbeg + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) va = fmaxf(va, fabsf(sc[e]) * sqr / is[e] + fabsf(fmaf(mu[e], sc[e], sh[e])));
            }
            if (XD) va *= xik;
        } else if constexpr (XF == 2) {
            const int ch = xgc + n0b + (tid & 31) * 4;
            const f32x4 mu = *(const f32x4*)(p.x_mean + ch), is = *(const f32x4*)(p.x_invstd + ch);
#pragma unroll
            for (int e = 0; e < 4; ++e) vb = fmaxf(vb, fabsf(xsc[e]) * sqr / is[e] + fabsf(fmaf(mu[e], xsc[e], xsh[e])));
            if (XD) vb *= xik;
        }
        if constexpr (DY) {
            if (p.dy_amax) {                               // per channel: |sc| max|d'| (the fourth coefficient row)
                if constexpr (KA) {
                    const int ch = m0 + (NQA == 4 ? (tid & 31) : (tid & 15)) * 4;
                    const f32x4 am = *(const f32x4*)(p.dy_coef + 3 * p.dy_cld + ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ca = fmaxf(ca, fabsf(dsc[e]) * am[e]);
                } else {
                    const int gc0 = (p.gmode == 1 ? (tn / p.gtiles) * (int)p.a_gs : 0) + kbeg;
                    for (int i = tid; i < T * BK; i += 256) ca = fmaxf(ca, fabsf(xfs[2 * SX_XF_KMAX + i]) * p.dy_coef[3 * p.dy_cld + gc0 + i]);
                }
                va = 1.f;
            } else if constexpr (KA) { ca = fmaxf(fmaxf(fabsf(dsc[0]), fabsf(dsc[1])), fmaxf(fabsf(dsc[2]), fabsf(dsc[3]))); }
            else for (int i = tid; i < T * BK; i += 256) ca = fmaxf(ca, fabsf(xfs[2 * SX_XF_KMAX + i]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { va = fmaxf(va, __shfl_xor(va, o, 64)); vb = fmaxf(vb, __shfl_xor(vb, o, 64)); ca = fmaxf(ca, __shfl_xor(ca, o, 64)); }
        if (lane == 0) { sred[wave * 3] = va; sred[wave * 3 + 1] = vb; sred[wave * 3 + 2] = ca; }
        __syncthreads();
        va = fmaxf(fmaxf(sred[0], sred[3]), fmaxf(sred[6], sred[9]));
        vb = fmaxf(fmaxf(sred[1], sred[4]), fmaxf(sred[7], sred[10]));
        ca = fmaxf(fmaxf(sred[2], sred[5]), fmaxf(sred[8], sred[11]));
        if constexpr (DY) va = ca * va * (2.f + sqr);
        sxe_a = __builtin_amdgcn_readfirstlane(sx_scale_exp(va));
        sxe_b = __builtin_amdgcn_readfirstlane(sx_scale_exp(vb));
        sxs_a = __uint_as_float((uint32_t)sxe_a << 23); sxs_b = __uint_as_float((uint32_t)sxe_b << 23);
    }
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
        SX_DY_A(q, 0); SX_DY_A(q, 1);
        SX_XF_HASH_A(q); SX_XF_A(q, 0); SX_XF_A(q, 1);
        if constexpr (NPC == 2) sx_split_store_h(raw[q], sxs_a, wa + q * WQA); else sx_split_store(raw[q], wa + q * WQA);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        SX_XF_HASH_B(q); SX_XF_B(q, 0); SX_XF_B(q, 1);
        if constexpr (NPC == 2) sx_split_store_h(raw[NQA + q], sxs_b, wb + q * WQB); else sx_split_store(raw[NQA + q], wb + q * WQB);
    }
    SX_XF_NEXT(1);
    SX_DY_NEXT(1);
    soa = min(soa + sta, enda); sob = min(sob + stb, endb);
#pragma unroll
    for (int q = 0; q < NQA; ++q) { raw[q] = SX_LOAD_A(q); SX_DY_LOAD(q); }
#pragma unroll
    for (int q = 0; q < 0; ++q) raw[q] = SX_LOAD_A(q);
#pragma unroll
    for (int q = 0; q < 4; ++q) raw[NQA + q] = SX_LOAD_B(q);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        bf16x8 a[2][2][NPC], b[2][2][NPC];                           // [k16 step][32-row window][piece] (NPC == 2: f16 bit patterns)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
#pragma unroll
                for (int i = 0; i < WM; ++i) a[s2][i][q] = sx_frag<KA>(fa + q * SX_PLANE + i * FWA, s2);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[s2][j][q] = sx_frag<KB>(fb + q * SX_PLANE + j * FWB, s2);
            }
        // every wave holds its fragments: the images may be overwritten.  (NPC == 2: that barrier sits in the MIDDLE of the generated body
        // -- the first k16 step's MFMAs run while the second step's fragments are still arriving, and only stage into registers)
        if constexpr (NPC == 3) __syncthreads();
        soa = min(soa + sta, enda); sob = min(sob + stb, endb);      // tile t+2 (past the end: a harmless repeat of the last tile)
        uint32_t pk0[2], pk1[2], pk2[2];
        uint32_t pkd[5][2][2];                                        // NPC == 2: the pieces of the quads staged before the barrier
        float r0, r1, a1;
        __builtin_amdgcn_sched_barrier(0);
        // (one generated body per (tile height, transformed operand, dropout): gen_split_body.py spreads the vector work over the MFMA slots)
        if constexpr (NPC == 2) {
            if constexpr (DY && XF == 0) {
                if (WM == 2) {
#include "gemm_split_body_h_wm2_dy.inc"
                } else {
#include "gemm_split_body_h_wm1_dy.inc"
                }
            } else if constexpr (DY && !XD) {
                if (WM == 2) {
#include "gemm_split_body_h_wm2_dyxb.inc"
                } else {
#include "gemm_split_body_h_wm1_dyxb.inc"
                }
            } else if constexpr (DY) {
                if (WM == 2) {
#include "gemm_split_body_h_wm2_dyxbd.inc"
                } else {
#include "gemm_split_body_h_wm1_dyxbd.inc"
                }
            } else if constexpr (XF == 0) {
                if (WM == 2) {
#include "gemm_split_body_h_wm2.inc"
                } else {
#include "gemm_split_body_h_wm1.inc"
                }
            } else if constexpr (XF == 1 && !XD) {
                if (WM == 2) {
#include "gemm_split_body_h_wm2_xa.inc"
                } else {
#include "gemm_split_body_h_wm1_xa.inc"
                }
            } else if constexpr (XF == 1 && XD) {
                if (WM == 2) {
#include "gemm_split_body_h_wm2_xad.inc"
                } else {
#include "gemm_split_body_h_wm1_xad.inc"
                }
            } else if constexpr (XF == 2 && !XD) {
                if (WM == 2) {
#include "gemm_split_body_h_wm2_xb.inc"
                } else {
#include "gemm_split_body_h_wm1_xb.inc"
                }
            } else {
                if (WM == 2) {
#include "gemm_split_body_h_wm2_xbd.inc"
                } else {
#include "gemm_split_body_h_wm1_xbd.inc"
                }
            }
        } else if constexpr (DY && XF == 0) {
            if (WM == 2) {
#include "gemm_split_body_wm2_dy.inc"
            } else {
#include "gemm_split_body_wm1_dy.inc"
            }
        } else if constexpr (DY && !XD) {
            if (WM == 2) {
#include "gemm_split_body_wm2_dyxb.inc"
            } else {
#include "gemm_split_body_wm1_dyxb.inc"
            }
        } else if constexpr (DY) {
            if (WM == 2) {
#include "gemm_split_body_wm2_dyxbd.inc"
            } else {
#include "gemm_split_body_wm1_dyxbd.inc"
            }
        } else if constexpr (XF == 0) {
            if (WM == 2) {
#include "gemm_split_body_wm2.inc"
            } else {
#include "gemm_split_body_wm1.inc"
            }
        } else if constexpr (XF == 1 && !XD) {
            if (WM == 2) {
#include "gemm_split_body_wm2_xa.inc"
            } else {
#include "gemm_split_body_wm1_xa.inc"
            }
        } else if constexpr (XF == 1 && XD) {
            if (WM == 2) {
#include "gemm_split_body_wm2_xad.inc"
            } else {
#include "gemm_split_body_wm1_xad.inc"
            }
        } else if constexpr (XF == 2 && !XD) {
            if (WM == 2) {
#include "gemm_split_body_wm2_xb.inc"
            } else {
#include "gemm_split_body_wm1_xb.inc"
            }
        } else {
            if (WM == 2) {
#include "gemm_split_body_wm2_xbd.inc"
            } else {
#include "gemm_split_body_wm1_xbd.inc"
            }
        }
        SX_XF_NEXT(t + 2);
        SX_DY_NEXT(t + 2);
        __syncthreads();
    }
#undef SX_LOAD_A
#undef SX_LOAD_B
#undef SX_XF_A
#undef SX_XF_B
#undef SX_XF_HASH_A
#undef SX_XF_HASH_B
#undef SX_DY_A
#undef SX_DY_LOAD
#undef SX_DY_NEXT
#undef SX_LOAD_AY
#undef SX_XF_NEXT
    if constexpr (NPC == 2) {                                        // undo the operand scales (a power of two: exact)
        const float inv = __uint_as_float((uint32_t)(127 - (sxe_a - 127) - (sxe_b - 127)) << 23);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
    }
    gemm_epilogue<WM, true, false, (!TA && !TB && XF == 0 && !DY) || (!TA && !TB && DY)>(p, acc, smem, tm, m0, n0, split, tid, l31, h, wm, wn);
}

// the calling entry point's `precision` argument for the duration of that call (common.h GemmPrecisionScope): 0: fp32 MFMA (exact fp32
// products); 1: bf16 operands, fp32 accumulation; 2: fp32-accurate six-product bf16 split (gemm_split_kernel)
// 3 (MLSP_PREC_F16X3): as 2, with two f16 pieces / three products on the launches gemm_split_kernel<.., NPC = 2> covers (tl_split_half); the
// rest of the dispatch sees mode 2.
static thread_local int tl_call_precision = 0;
static thread_local bool tl_split_half = false;
int gemm_precision_mode() { return tl_call_precision; }

// ---- operand magnitudes for the two-piece f16 products (mode 3) ------------------------------------------------------------------
// AMAX_PARTS partial maxima of |X| per operand, written by ONE streaming launch (no atomics, nothing to initialise) into a slot of the
// call's workspace tail (MLSP_AMAX_TAIL_BYTES, common.h); every workgroup of the consuming GEMM reduces the partials in its prologue.
// Within one API call an operand is measured once (thread-local cache keyed by pointer and shape: dgrad and weight gradient share dY);
// the cache and the slot ring die with the call (GemmPrecisionScope).
#define AMAX_PARTS 256
#define AMAX_SLOTS ((MLSP_AMAX_TAIL_BYTES - 256) / (AMAX_PARTS * 4))
struct AmaxOp { const float* X; long rows; int cols, ld; float* out; };
struct AmaxArgs { AmaxOp op[5]; };
__global__ __launch_bounds__(512) void amax_partials_kernel(AmaxArgs a) {
    const AmaxOp o = a.op[blockIdx.x / AMAX_PARTS];
    const int b = blockIdx.x % AMAX_PARTS, tid = threadIdx.x;
    float m = 0.f;
    const int c4 = o.cols >> 2;
    const long nq = o.rows * c4;                                   // aligned quads (host: cols % 4 == 0, ld % 4 == 0, 16-byte aligned base)
    const long per = (nq + AMAX_PARTS - 1) / AMAX_PARTS;
    const long q0 = (long)b * per, q1 = min(nq, q0 + per);
    if (o.ld == o.cols) {
        const f32x4* p = (const f32x4*)o.X;
        long q = q0 + tid;
        for (; q + 3 * 512 < q1; q += 4 * 512) {                   // four 16-byte loads in flight per thread
            const f32x4 v0 = p[q], v1 = p[q + 512], v2 = p[q + 1024], v3 = p[q + 1536];
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))), fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3])))));
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(v2[0]), fabsf(v2[1])), fmaxf(fabsf(v2[2]), fabsf(v2[3]))), fmaxf(fmaxf(fabsf(v3[0]), fabsf(v3[1])), fmaxf(fabsf(v3[2]), fabsf(v3[3])))));
        }
        for (; q < q1; q += 512) { const f32x4 v = p[q]; m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])))); }
    } else {
        for (long q = q0 + tid; q < q1; q += 512) {
            const long r = q / c4; const int c = (int)(q - r * c4);
            const f32x4 v = *(const f32x4*)(o.X + r * o.ld + 4 * c);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
    __shared__ float sm[8];
#pragma unroll
    for (int s_ = 32; s_ > 0; s_ >>= 1) m = fmaxf(m, __shfl_xor(m, s_, 64));
    if ((tid & 63) == 0) sm[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) o.out[b] = fmaxf(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])), fmaxf(fmaxf(sm[4], sm[5]), fmaxf(sm[6], sm[7])));
}
static thread_local struct AmaxScratch {
    float* base = nullptr;          // the workspace tail of the current API call (null: none, mode 3 falls back to the bf16 pieces)
    int next = 0, ncache = 0;
    struct { const float* X; long rows; int cols, ld; float* out; } cache[AMAX_SLOTS];
    // caller-owned bounds of the current call (mlsp_operand_bounds_next): an operand that matches an entry uses the entry's partials --
    // as they are when `valid`, else this call measures INTO them and sets `valid` (the caller keeps them for later calls: the same
    // activation read by several layers, a weight read again by its layer's backward)
    mlsp_bound_t* offered = nullptr; int noffered = 0;
    mlsp_bound_t* pending = nullptr; int npending = 0;     // set by mlsp_operand_bounds_next, adopted by the next GemmPrecisionScope
} tl_amax;
extern "C" int mlsp_operand_bounds_next(mlsp_bound_t* tab, int n) {
    if (n < 0 || (n > 0 && !tab)) return MLSP_ERR_ARG;
    tl_amax.pending = n ? tab : nullptr; tl_amax.npending = n;
    return MLSP_OK;
}
// queue of operands to measure with the next launch (launch_gemm batches A and B -- and the groups' B operands -- into one launch)
struct AmaxBatch { AmaxArgs args; int n = 0; };
static float* amax_take_slot(const float* X, long rows, int cols, int ld) {
    const int slot = tl_amax.next;
    tl_amax.next = (tl_amax.next + 1) % AMAX_SLOTS;
    float* out = tl_amax.base + (size_t)slot * AMAX_PARTS;
    int k = 0;                                                   // the slot's previous tenant leaves the cache
    for (int i = 0; i < tl_amax.ncache; ++i) if (tl_amax.cache[i].out != out) tl_amax.cache[k++] = tl_amax.cache[i];
    tl_amax.ncache = k;
    tl_amax.cache[tl_amax.ncache++] = {X, rows, cols, ld, out};
    return out;
}
// -> the partials of X [rows][cols] (pitch ld), measured now (queued into `batch`) or earlier in this API call; null: cannot (no tail,
// unaligned, the batch is full)
static float* amax_get(AmaxBatch& batch, const float* X, long rows, int cols, int ld, int* n_out) {
    *n_out = AMAX_PARTS;
    if (!tl_amax.base || rows <= 0 || cols <= 0 || (cols & 3) || (ld & 3) || (((uintptr_t)X) & 15)) return nullptr;
    for (int i = 0; i < tl_amax.noffered; ++i) {
        mlsp_bound_t& o = tl_amax.offered[i];
        if (o.ptr != X || o.rows != rows || o.cols != cols || o.ld != ld || !o.partials || o.n < 0 || o.n > 4096) continue;
        if (!o.valid) {
            if (o.n != 0 && o.n != AMAX_PARTS) continue;            // (an unfilled OUTPUT-bounds buffer of another size: not ours to measure into)
            if (batch.n >= 5) return nullptr;
            batch.args.op[batch.n++] = {X, rows, cols, ld, o.partials};
            o.valid = 1; o.n = AMAX_PARTS;
        }
        *n_out = o.n ? o.n : AMAX_PARTS;
        return o.partials;
    }
    for (int i = 0; i < tl_amax.ncache; ++i) {
        const auto& c = tl_amax.cache[i];
        if (c.X == X && c.rows == rows && c.cols == cols && c.ld == ld) return c.out;
    }
    if (batch.n >= 5) return nullptr;
    float* out = amax_take_slot(X, rows, cols, ld);
    batch.args.op[batch.n++] = {X, rows, cols, ld, out};
    return out;
}
// Is a bound of X at hand (a valid entry of the caller's table, or measured / produced earlier in this API call)?  No side effects.
static bool amax_at_hand(const float* X, long rows, int cols, int ld) {
    for (int i = 0; i < tl_amax.noffered; ++i) {
        const mlsp_bound_t& o = tl_amax.offered[i];
        if (o.ptr == X && o.rows == rows && o.cols == cols && o.ld == ld && o.partials && o.valid && o.n > 0 && o.n <= 4096) return true;
    }
    for (int i = 0; i < tl_amax.ncache; ++i) {
        const auto& c = tl_amax.cache[i];
        if (c.X == X && c.rows == rows && c.cols == cols && c.ld == ld) return true;
    }
    return false;
}
// A slot for the partial maxima of X, FILLED BY THE CALLER's own kernels (the passes that write X, e.g. edge.hip edge_amax_raise) before
// the product that reads X is launched in this same API call: that product finds the slot like a measured one (an entry of the caller's
// table for X is filled instead and marked valid: the caller hands it to later calls).  null: no f16x3 scope / X is not an operand the
// products would look up / the caller's entry is valid already.
float* amax_reserve(const float* X, long rows, int cols, int ld) {
    if (!tl_amax.base || !tl_split_half || rows <= 0 || cols <= 0 || (cols & 3) || (ld & 3) || (((uintptr_t)X) & 15)) return nullptr;
    for (int i = 0; i < tl_amax.noffered; ++i) {                 // the caller's own entry for X: filled there (and handed on by the caller), or already valid
        mlsp_bound_t& o = tl_amax.offered[i];
        if (o.ptr != X || o.rows != rows || o.cols != cols || o.ld != ld) continue;
        if (!o.partials || o.valid || (o.n != 0 && o.n != AMAX_PARTS)) return nullptr;
        o.valid = 1; o.n = AMAX_PARTS;
        return o.partials;
    }
    return amax_take_slot(X, rows, cols, ld);
}
// An entry of the caller's table that describes an OUTPUT of the current call (same pointer / shape, valid == 0, room for `need` floats):
// the call fills its partials with a bound of what it writes -- e.g. the per-channel analytic bound of a BatchNorm'd layer -- and the
// caller hands them to the layers that read the tensor.  -> the partials (entry marked valid, n = need), or null.
float* amax_offered_output(const float* out, long rows, int cols, int ld, int need) {
    for (int i = 0; i < tl_amax.noffered; ++i) {
        mlsp_bound_t& o = tl_amax.offered[i];
        if (o.ptr != out || o.rows != rows || o.cols != cols || o.ld != ld || !o.partials || o.valid || o.n < need) continue;
        o.valid = 1; o.n = need;
        return o.partials;
    }
    return nullptr;
}
static void amax_flush(hipStream_t st, AmaxBatch& batch) {
    static const bool dump = getenv("MLSP_AMAX_DUMP") != nullptr;          // read-once diagnostic (tools/r6): what each measuring launch reads
    if (dump && batch.n) {
        fprintf(stderr, "amax launch:");
        for (int i = 0; i < batch.n; ++i) fprintf(stderr, " [%ld x %d ld %d = %.1f MB]", batch.args.op[i].rows, batch.args.op[i].cols, batch.args.op[i].ld, batch.args.op[i].rows * 4e-6 * batch.args.op[i].cols);
        fprintf(stderr, "\n");
    }
    if (batch.n) hipLaunchKernelGGL(amax_partials_kernel, dim3(batch.n * AMAX_PARTS), dim3(512), 0, st, batch.args);
    batch.n = 0;
}
GemmPrecisionScope::GemmPrecisionScope(int mode, void* ws, size_t ws_bytes) : prev(tl_call_precision | (tl_split_half ? 4 : 0)), prev_tail(tl_amax.base) {
    tl_split_half = mode == 3; tl_call_precision = mode == 3 ? 2 : mode;
    tl_amax.base = (mode == 3 && ws && ws_bytes >= 2 * MLSP_AMAX_TAIL_BYTES) ? (float*)((char*)ws + align_up(ws_bytes - MLSP_AMAX_TAIL_BYTES, 256)) : nullptr;
    tl_amax.next = tl_amax.ncache = 0;
    // the caller's bounds table is consumed by the OUTERMOST scope of the call (a nested scope -- an entry point asking one of the
    // shape queries -- leaves it, and the outer scope's measuring state, alone)
    prev_offered = tl_amax.offered; prev_noffered = tl_amax.noffered;
    if (tl_amax.pending) { tl_amax.offered = tl_amax.pending; tl_amax.noffered = tl_amax.npending; tl_amax.pending = nullptr; tl_amax.npending = 0; }
}
GemmPrecisionScope::~GemmPrecisionScope() {
    tl_call_precision = prev & 3; tl_split_half = (prev & 4) != 0;
    tl_amax.base = (float*)prev_tail; tl_amax.next = tl_amax.ncache = 0;
    tl_amax.offered = (mlsp_bound_t*)prev_offered; tl_amax.noffered = prev_noffered;
}

// sum the split-K slabs (fixed order -> bitwise reproducible) and apply the epilogue
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, int M, int N, int ldc,
                                     int nsplit, const float* __restrict__ bias, const float* __restrict__ gbias,
                                     int rows_per_group) {
    size_t total = (size_t)M * N;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int row = (int)(i / N), col = (int)(i % N);
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += slab[(size_t)z * M * N + i];
        if (bias) s += bias[col];
        if (gbias) s += gbias[(size_t)(row / rows_per_group) * N + col];
        C[(size_t)row * ldc + col] = s;
    }
}

static void launch_splitk_reduce_any(hipStream_t st, const float* slab, float* C, int M, int N, int ldc, int ns, const float* bias,
                                     const float* gbias, int rows_per_group);

int launch_slab_reduce(hipStream_t st, const float* slab, float* C, int M, int N, int ldc, int nsplit) {
    launch_splitk_reduce_any(st, slab, C, M, N, ldc, nsplit, nullptr, nullptr, 0);
    return mlsp_launch_status();
}

// vectorised slab reduce (contiguous C only): 16 bytes per lane.  A workgroup covers 256/ZG consecutive float4 outputs;
// its ZG thread groups each sum the slabs z = g, g+ZG, ... (8 loads in flight), then the groups are added in order through
// LDS.  The order is fixed by (nsplit, ZG) alone -> bitwise reproducible.
template <int ZG>
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const float* __restrict__ slab, float* __restrict__ C, size_t total4,
                                                                int N4, int nsplit, const float* __restrict__ bias,
                                                                const float* __restrict__ gbias, int rows_per_group) {
    constexpr int EPB = 256 / ZG;                 // float4 outputs per workgroup
    __shared__ f32x4 part[ZG > 1 ? 256 : 1];
    const int e = threadIdx.x % EPB, g = threadIdx.x / EPB;
    const size_t v = (size_t)blockIdx.x * EPB + e;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (v < total4) {
        const f32x4* p = (const f32x4*)slab + v;
        int z = g;
        for (; z + 7 * ZG < nsplit; z += 8 * ZG) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p[(size_t)(z + u * ZG) * total4];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = s + t[u];
        }
        for (; z < nsplit; z += ZG) s = s + p[(size_t)z * total4];
    }
    if (ZG > 1) {
        part[threadIdx.x] = s;
        __syncthreads();
        if (g != 0) return;
#pragma unroll
        for (int q = 1; q < ZG; ++q) s = s + part[q * EPB + e];
    }
    if (v >= total4) return;
    const int col = (int)(v % N4) * 4;
    if (bias) { const f32x4 bv = *(const f32x4*)(bias + col); s = s + bv; }
    if (gbias) {
        const size_t row = v / N4;
        const f32x4 gv = *(const f32x4*)(gbias + (row / rows_per_group) * (size_t)N4 * 4 + col);
        s = s + gv;
    }
    ((f32x4*)C)[v] = s;
}

// scalar outputs with many slabs (N = 3 wgrads ...): one wave per output element, lanes stride the slabs, fixed shuffle tree
__global__ __launch_bounds__(256) void splitk_reduce_wave_kernel(const float* __restrict__ slab, float* __restrict__ C, int M, int N,
                                                                 int ldc, int nsplit, const float* __restrict__ bias,
                                                                 const float* __restrict__ gbias, int rows_per_group) {
    const size_t total = (size_t)M * N;
    const int lane = threadIdx.x & 63;
    for (size_t i = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < total; i += (size_t)gridDim.x * 4) {
        float s = 0.f;
        for (int z = lane; z < nsplit; z += 64) s += slab[(size_t)z * total + i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) {
            const int row = (int)(i / N), col = (int)(i % N);
            if (bias) s += bias[col];
            if (gbias) s += gbias[(size_t)(row / rows_per_group) * N + col];
            C[(size_t)row * ldc + col] = s;
        }
    }
}

// The split-K sum of an EdgeConv weight gradient dWd = [dWd_u ; dWd_v] ([2 Cout][C], edge.hip build_wd_kernel's layout) written straight in
// the reference layout dW [Cout][2 C] = [dWd_u - dWd_v | dWd_v] (unbuild_wd_kernel's arithmetic on the same two sums, each taken in
// splitk_reduce_vec_kernel's order): the reduce and the unfold are one launch.  v indexes the float4s of the U half.
template <int ZG>
__global__ __launch_bounds__(256) void splitk_reduce_unfold_kernel(const float* __restrict__ slab, float* __restrict__ dW, size_t half4, int C4,
                                                                   int nsplit) {
    constexpr int EPB = 256 / ZG;
    __shared__ f32x4 part[ZG > 1 ? 512 : 1];
    const int e = threadIdx.x % EPB, g = threadIdx.x / EPB;
    const size_t v = (size_t)blockIdx.x * EPB + e;
    const size_t total4 = 2 * half4;
    f32x4 su = {0.f, 0.f, 0.f, 0.f}, sv = {0.f, 0.f, 0.f, 0.f};
    if (v < half4) {
        const f32x4* p = (const f32x4*)slab + v;
        int z = g;
        for (; z + 3 * ZG < nsplit; z += 4 * ZG) {
            f32x4 t[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { t[u] = p[(size_t)(z + u * ZG) * total4]; w[u] = p[(size_t)(z + u * ZG) * total4 + half4]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { su = su + t[u]; sv = sv + w[u]; }
        }
        for (; z < nsplit; z += ZG) { su = su + p[(size_t)z * total4]; sv = sv + p[(size_t)z * total4 + half4]; }
    }
    if (ZG > 1) {
        part[threadIdx.x] = su; part[256 + threadIdx.x] = sv;
        __syncthreads();
        if (g != 0) return;
#pragma unroll
        for (int q = 1; q < ZG; ++q) { su = su + part[q * EPB + e]; sv = sv + part[256 + q * EPB + e]; }
    }
    if (v >= half4) return;
    const size_t o = v / C4, c4 = v % C4;
    f32x4* out = (f32x4*)dW + o * 2 * C4 + c4;
    out[0] = su - sv;
    out[C4] = sv;
}
// request of the NEXT split-K reduction on this thread (launch_gemm of an EdgeConv weight gradient): consumed by launch_splitk_reduce_any,
// which clears it; gemm_unfold_take() tells the caller whether the reduce did the unfold (a launch without K splits did not)
static thread_local float* tl_unfold_dw = nullptr;
static thread_local bool tl_unfold_done = false;
void gemm_unfold_request(float* dW) { tl_unfold_dw = dW; tl_unfold_done = false; }
bool gemm_unfold_take() { const bool d = tl_unfold_done; tl_unfold_dw = nullptr; tl_unfold_done = false; return d; }

static void launch_splitk_reduce_any(hipStream_t st, const float* slab, float* C, int M, int N, int ldc, int ns, const float* bias,
                                     const float* gbias, int rows_per_group) {
    size_t total = (size_t)M * N;
    const bool vec = (N % 4 == 0) && ldc == N && ((((uintptr_t)slab | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)gbias) & 15) == 0);
    if (tl_unfold_dw && vec && !bias && !gbias && M % 2 == 0 && (((uintptr_t)tl_unfold_dw) & 15) == 0) {
        float* dW = tl_unfold_dw;
        tl_unfold_dw = nullptr; tl_unfold_done = true;
        const size_t h4 = total / 8;
        if (ns >= 32 && 2 * h4 <= 64 * 1024)
            hipLaunchKernelGGL((splitk_reduce_unfold_kernel<8>), dim3((unsigned)((h4 + 31) / 32)), dim3(256), 0, st, slab, dW, h4, N / 4, ns);
        else if (ns >= 8)
            hipLaunchKernelGGL((splitk_reduce_unfold_kernel<4>), dim3((unsigned)((h4 + 63) / 64)), dim3(256), 0, st, slab, dW, h4, N / 4, ns);
        else
            hipLaunchKernelGGL((splitk_reduce_unfold_kernel<1>), dim3((unsigned)((h4 + 255) / 256)), dim3(256), 0, st, slab, dW, h4, N / 4, ns);
        return;
    }
    tl_unfold_dw = nullptr;
    if (vec) {
        size_t t4 = total / 4;
        // enough workgroups to fill the chip: more slab groups per workgroup when the output is small
        if (ns >= 32 && t4 <= 64 * 1024)
            hipLaunchKernelGGL((splitk_reduce_vec_kernel<8>), dim3((unsigned)((t4 + 31) / 32)), dim3(256), 0, st, slab, C, t4, N / 4, ns, bias, gbias, rows_per_group);
        else if (ns >= 8)
            hipLaunchKernelGGL((splitk_reduce_vec_kernel<4>), dim3((unsigned)((t4 + 63) / 64)), dim3(256), 0, st, slab, C, t4, N / 4, ns, bias, gbias, rows_per_group);
        else
            hipLaunchKernelGGL((splitk_reduce_vec_kernel<1>), dim3((unsigned)((t4 + 255) / 256)), dim3(256), 0, st, slab, C, t4, N / 4, ns, bias, gbias, rows_per_group);
    } else if (ns >= 32 && total <= 256 * 1024) {
        int blocks = (int)((total + 3) / 4 < 4096 ? (total + 3) / 4 : 4096);
        hipLaunchKernelGGL(splitk_reduce_wave_kernel, dim3(blocks), dim3(256), 0, st, slab, C, M, N, ldc, ns, bias, gbias, rows_per_group);
    } else {
        int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, slab, C, M, N, ldc, ns, bias, gbias, rows_per_group);
    }
}

// How many K splits a launch will use (shared by the launcher and mlsp_workspace_bytes).
int gemm_pick_split(int M, int N, int K) {
    int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    long tiles = (long)ntm * ntn;
    int ktiles = (K + BK - 1) / BK;
    if (tiles >= 256 || ktiles < 4) return 1;
    long want = (512 + tiles - 1) / tiles;          // aim at ~2 blocks per CU (measured: 320 is 35 % slower on the wgrad shapes; 768 / 1024 and 64-row tiles for the split launches: no gain)
    int ns = (int)(want < ktiles / 2 ? want : ktiles / 2);
    if (ns < 1) ns = 1;
    if (ns > 256) ns = 256;
    return ns;
}

// number of BN-statistic partial rows a stats-fused launch writes (0: the launch would split K, use colstats)
// 64-row tiles when the 128-row grid is too small to keep ~3 workgroups per CU busy (and K is not split)
// mode 2: does a contraction of `ktiles` K-tiles per workgroup go to the split kernel?  Short K loops stay on the fp32 kernel (measured per
// launch on the headline step, tools/cmp_dump.py, and per shape, tools/x6/lib_bench: the split kernel's longer prologue loses at K = 64 and,
// at K = 128, on grids too small to hide it: few rows AND a single column tile)
static bool gemm_split_pays(int M, int N, int ktiles) {
    static const bool always = getenv("MLSP_GEMM_SPLIT_ALWAYS") != nullptr;        // read-once A/B switch (tools/x6/lib_bench)
    return always || ktiles >= 8 || (ktiles >= 4 && (N >= 256 || M >= 16384));
}
static int gemm_pick_bm(int M, int N, int K) {
    long tiles128 = (long)((M + 127) / 128) * ((N + BN - 1) / BN);
    // the split kernel amortises its operand split over the tile: 128 rows unless the grid would not fill the 512 workgroup slots
    const long few = (tl_call_precision == 2 && gemm_split_pays(M, N, (K + BK - 1) / BK)) ? 512 : 1536;
    static const bool force64 = getenv("MLSP_GEMM_BM64") != nullptr;            // read-once experiment switch (tools/x6/lib_bench): 64-row tiles wherever K is not split
    if (force64 && gemm_pick_split(M, N, K) == 1 && M >= 256) return 64;
    return (gemm_pick_split(M, N, K) == 1 && tiles128 < few && M >= 256) ? 64 : 128;
}
int gemm_stat_parts(int M, int N, int K) {
    if (gemm_pick_split(M, N, K) != 1) return 0;
    int bm = gemm_pick_bm(M, N, K);
    return (M + bm - 1) / bm;
}
int gemm_panel_rows(int M, int N, int K) { return gemm_pick_bm(M, N, K); }

size_t thin_tn_slab_floats(int M, int N, int K);
// A 64 x 64 weight gradient over many rows (dW = dY^T X of a 64 -> 64 layer over the edges of a set-abstraction level): neither the 128-row
// tile kernels (a quarter of the tile used, predicated path) nor the N = 64 kernel (M % 128) fit it.  With contiguous operands two
// consecutive rows are ONE row of a [K/2][128] matrix, and  A2^T B2  (128 x 128, an interior-tile launch) holds the even-row sum in its
// upper-left 64 x 64 block and the odd-row sum in its lower-right one: dW = their sum.  Twice the products on the fast kernel instead of four
// times on the slow one (configs[3]: 216 -> 70 us per step).
static bool gemm_fold64(bool ta, bool tb, int M, int N, int K, int lda, int ldb) {
    return ta && !tb && M == 64 && N == 64 && lda == 64 && ldb == 64 && K % 64 == 0 && K >= 8192;
}
size_t gemm_slab_floats(int M, int N, int K) {
    int ns = gemm_pick_split(M, N, K);
    size_t a = ns > 1 ? (size_t)ns * M * N : 0;
    const size_t b = thin_tn_slab_floats(M, N, K);      // thin.hip: A^T B with one side <= 16
    if (gemm_fold64(true, false, M, N, K, 64, 64)) {    // the folded launch's slab + its 128 x 128 result
        const size_t f = gemm_slab_floats(128, 128, K / 2) + 128 * 128;
        a = a > f ? a : f;
    }
    return a > b ? a : b;
}
__global__ __launch_bounds__(256) void gemm_fold64_sum_kernel(const float* __restrict__ C2, float* __restrict__ C, int ldc) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), j = threadIdx.x & 63;          // 16 workgroups x 4 rows x 64 columns
    C[(size_t)i * ldc + j] = C2[i * 128 + j] + C2[(64 + i) * 128 + 64 + j];
}

// ---- optional HIP-event profiling of the GEMM launches (bench.py roofline) -----------------------
// Off by default.  mlsp_profile_begin() arms it; every gemm_f32_kernel launch is then bracketed by two
// events recorded on the launch stream; mlsp_profile_end() returns {ms, launches, algorithmic FLOP}.
#include <vector>
#include <cstdlib>
#include <cstdio>
static struct GemmProf {
    bool on = false;
    std::vector<hipEvent_t> ev;     // pairs
    size_t used = 0;
    double flop = 0.0, bytes = 0.0;   // algorithmic: 2*M*N*K and the A + B + C bytes in their storage types
    struct Rec { int M, N, K, ta, tb, ns, bm, fused, split, extra; };       // extra: bit 0 = the A operand is a (d', y) PAIR (DY), bit 1 = C is read too (beta = 1)
    std::vector<Rec> rec;           // one per pair, for the MLSP_PROF_DUMP listing
} g_prof;
#define PROF_MAX_PAIRS 4096

// ---- the same for the other kernel families the bench line prices (kNN per channel count, LDS gather-reduce, T-Net stages):
// prof_cls_begin / prof_cls_end (common.h) bracket a launch with two events on its stream, `work` = its algorithmic bytes or FLOP.
static struct ClsProf {
    std::vector<hipEvent_t> ev;
    std::vector<int> cls;
    std::vector<double> work;
    size_t used = 0;
} g_cls;
int prof_cls_begin(hipStream_t st, int cls) {
    if (!g_prof.on || g_cls.used >= PROF_MAX_PAIRS || cls <= 0 || cls >= MLSP_PROF_NCLS || g_cls.ev.empty()) return -1;
    const int tok = (int)g_cls.used++;
    g_cls.cls[tok] = cls; g_cls.work[tok] = 0.0;
    (void)hipEventRecord(g_cls.ev[2 * tok], st);
    return tok;
}
void prof_cls_end(hipStream_t st, int tok, double work) {
    if (tok < 0) return;
    (void)hipEventRecord(g_cls.ev[2 * tok + 1], st);
    g_cls.work[tok] = work;
}

extern "C" int mlsp_profile_begin(void) {
    if (g_cls.ev.empty()) {
        g_cls.ev.resize(2 * PROF_MAX_PAIRS); g_cls.cls.resize(PROF_MAX_PAIRS); g_cls.work.resize(PROF_MAX_PAIRS);
        for (auto& e : g_cls.ev)
            if (hipEventCreate(&e) != hipSuccess) { g_cls.ev.clear(); return MLSP_ERR_UNSUPPORTED; }
    }
    g_cls.used = 0;
    if (g_prof.ev.empty()) {
        g_prof.ev.resize(2 * PROF_MAX_PAIRS);
        g_prof.rec.resize(PROF_MAX_PAIRS);
        for (auto& e : g_prof.ev)
            if (hipEventCreate(&e) != hipSuccess) { g_prof.ev.clear(); return MLSP_ERR_UNSUPPORTED; }
    }
    g_prof.used = 0; g_prof.flop = 0.0; g_prof.bytes = 0.0; g_prof.on = true;
    return MLSP_OK;
}

// out[0] = total milliseconds inside gemm_f32_kernel, out[1] = launches, out[2] = sum of 2*M*N*K, out[3] = sum of the A + B + C bytes
extern "C" int mlsp_profile_end(double* out) {
    g_prof.on = false;
    double ms = 0.0;
    const bool dump = getenv("MLSP_PROF_DUMP") != nullptr;    // debug listing: one line per profiled launch
    for (size_t i = 0; i < g_prof.used; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return MLSP_ERR_UNSUPPORTED;
        if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return MLSP_ERR_UNSUPPORTED;
        ms += t;
        if (dump) {
            const auto& r = g_prof.rec[i];
            fprintf(stderr, "gemm %c%c M=%d N=%d K=%d split=%d bm=%d epi=%d %s  %.1f us  %.1f TF\n", r.ta ? 'T' : 'N', r.tb ? 'T' : 'N',
                    r.M, r.N, r.K, r.ns, r.bm, r.fused, r.split ? "bf16x6" : "f32", t * 1e3, 2.0 * r.M * r.N * r.K / (t * 1e-3) / 1e12);
        }
    }
    if (out) { out[0] = ms; out[1] = (double)g_prof.used; out[2] = g_prof.flop; out[3] = g_prof.bytes; }
    return MLSP_OK;
}

// out [MLSP_PROF_NCLS][3] = {milliseconds, launches, algorithmic work} per kernel class of the last begin/end bracket (class 0 = the
// GEMM family, same figures as mlsp_profile_end).  Call after mlsp_profile_end.
extern "C" int mlsp_profile_classes(double* out, int ncls) {
    if (!out || ncls < MLSP_PROF_NCLS || g_prof.on) return MLSP_ERR_ARG;
    for (int i = 0; i < 3 * ncls; ++i) out[i] = 0.0;
    for (size_t i = 0; i < g_prof.used; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return MLSP_ERR_UNSUPPORTED;
        out[0] += t; out[1] += 1.0;
        if (g_prof.rec[i].split) {
            const auto& r = g_prof.rec[i];
            out[3 * MLSP_PROF_GEMM_SPLIT] += t; out[3 * MLSP_PROF_GEMM_SPLIT + 1] += 1.0;
            out[3 * MLSP_PROF_GEMM_SPLIT + 2] += 2.0 * r.M * (double)r.N * r.K;
        }
    }
    out[2] = g_prof.flop;
    for (size_t i = 0; i < g_cls.used; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(g_cls.ev[2 * i + 1]) != hipSuccess) return MLSP_ERR_UNSUPPORTED;
        if (hipEventElapsedTime(&t, g_cls.ev[2 * i], g_cls.ev[2 * i + 1]) != hipSuccess) return MLSP_ERR_UNSUPPORTED;
        const int c = g_cls.cls[i];
        out[3 * c] += t; out[3 * c + 1] += 1.0; out[3 * c + 2] += g_cls.work[i];
    }
    return MLSP_OK;
}

// out [4][4] = {milliseconds, launches, algorithmic FLOP, algorithmic A + B + C bytes (split-K slabs: written and read once more)} of the
// launches of the last bracket that ran on gemm_split_kernel, by kind: 0 forward (A row-major, B = W as stored), 1 dgrad, 2 wgrad
// (k-major A); row 3: the subset of all kinds that ran on the two-piece f16 products (mode 3).  bench.py prices each kind's HBM-side bytes
// (rocprofv3 --pmc, per template instantiation) against these.
extern "C" int mlsp_profile_split_kinds(double* out) {
    if (!out || g_prof.on) return MLSP_ERR_ARG;
    for (int i = 0; i < 16; ++i) out[i] = 0.0;
    for (size_t i = 0; i < g_prof.used; ++i) {
        const auto& r = g_prof.rec[i];
        if (!r.split) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return MLSP_ERR_UNSUPPORTED;
        const int kind = r.ta ? 2 : (r.tb ? 0 : 1);
        const double fl = 2.0 * r.M * (double)r.N * r.K;
        // A + B + C as this formulation reads / writes them: a (d', y) pair is two A operands, a beta = 1 launch reads C as well
        const double by = 4.0 * ((double)r.M * r.K * ((r.extra & 1) ? 2.0 : 1.0) + (double)r.K * r.N + (double)r.M * r.N * ((r.ns > 1 ? 2.0 * r.ns + 1.0 : 1.0) + ((r.extra & 2) ? 1.0 : 0.0)));
        double* o = out + 4 * kind;
        o[0] += t; o[1] += 1.0; o[2] += fl; o[3] += by;
        if (r.split == 2) { out[12] += t; out[13] += 1.0; out[14] += fl; out[15] += by; }     // (row 3: those on the two-piece f16 products)
    }
    return MLSP_OK;
}

int launch_skinny_gemm(hipStream_t st, bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                       float* C, int ldc, const float* bias);
int launch_thin_gemm(hipStream_t st, bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                     int ldc, const float* bias, float* slab, size_t slab_floats, const GemmXf* xf, const GemmBs* bs = nullptr);
int thin_bs_parts(int M, int N, int K);
bool thin_xf_supported(bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, int which);

// Can this contraction stage `which` (1: A [M][K] row-major, 2: B [K][N] k-major) through the operand transform?  Interior tiles,
// 16-byte loads, fp32 operands, the MFMA tile kernels (not the thin / skinny / N = 64 ones).
// (mode 2: gemm_split_kernel<.., XF, XD> where the split kernel pays, the fp32 transform kernels on the short-K launches: exact fp32
// products either way.  On the split kernel the A-side transform keeps the scale / shift of a workgroup's K range in LDS: <= SX_XF_KMAX.)
bool gemm_xf_supported(bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, int which) {
    if (tl_call_precision == 1) return false;
    if (thin_xf_supported(ta, tb, M, N, K, A, lda, B, ldb, which)) return true;       // the streaming kernels of thin.hip transform too
    if (which == 1 ? ta : !(ta && !tb)) return false;
    if (M <= 32 || N < 32 || K < 32 || (ta && !tb && K <= 32)) return false;
    const bool vec = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0) && (ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);
    int ns = gemm_pick_split(M, N, K);
    const int bm = (ns == 1) ? gemm_pick_bm(M, N, K) : 128;
    const int ktiles = (K + BK - 1) / BK, kts = (ktiles + ns - 1) / ns;
    if (which == 1 && tl_call_precision == 2 && gemm_split_pays(M, N, kts) && kts * BK > SX_XF_KMAX) return false;
    return vec && (M % bm == 0) && (N % BN == 0) && (K % BK == 0) && (long)(ta ? M : K) * lda * 4 < (1L << 30) && (long)K * ldb * 4 < (1L << 30);
}

// Row panels (of 128 rows) a dgrad dX [M][N] = dY [M][K] W with the fused statistics pass (GemmBs) writes partial sums for; 0: this
// shape does not take the fused pass (the caller runs the streaming reduction instead).  Split kernel, 128-row interior tiles, one K pass.
int gemm_bs_parts(int M, int N, int K, int lda, int ldb, int ldc) {
    if (tl_call_precision != 2 || M % 128 || N % BN || K % BK || lda % 4 || ldb % 4 || ldc % 4) return 0;
    if (gemm_pick_split(M, N, K) != 1 || gemm_pick_bm(M, N, K) != 128 || !gemm_split_pays(M, N, K / BK)) return 0;
    return M / 128;
}

// Can the layer's dgrad (ta = tb = false: dX [M][N] = dY [M][K] W) / weight gradient (ta, !tb: dW [M][N] = dY^T [K][M] X [K][N]) form dY from
// (d', y) in its A operand loads (GemmDy)?  gemm_split_kernel only: interior tiles, 16-byte loads, the dgrad's K range within SX_XF_KMAX.
bool gemm_dy_supported(bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb) {
    if (tl_call_precision != 2 || tb) return false;
    if (M <= 32 || N < 32 || K <= 32) return false;
    const bool vec = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0) && (ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);
    const int ns = gemm_pick_split(M, N, K);
    const int bm = (ns == 1) ? gemm_pick_bm(M, N, K) : 128;
    const int ktiles = (K + BK - 1) / BK, kts = (ktiles + ns - 1) / ns;
    if (!gemm_split_pays(M, N, kts) || (!ta && kts * BK > SX_XF_KMAX)) return false;
    return vec && (M % bm == 0) && (N % BN == 0) && (K % BK == 0) && (long)(ta ? K : M) * lda * 4 < (1L << 30) && (long)K * ldb * 4 < (1L << 30);
}

// will a transform launch of this shape run on gemm_split_kernel (the only kernel that transforms inside a block-diagonal launch)?
bool gemm_xf_on_split(bool ta, bool tb, int M, int N, int K, int which) {
    if (tl_call_precision != 2) return false;
    const int ns = gemm_pick_split(M, N, K), ktiles = (K + BK - 1) / BK, kts = (ktiles + ns - 1) / ns;
    return gemm_split_pays(M, N, kts) && (which != 1 || kts * BK <= SX_XF_KMAX);
}

int launch_gemm(hipStream_t st, bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B,
                int ldb, float* C, int ldc, const float* bias, const float* gbias, int rows_per_group, float* slab,
                size_t slab_floats, double* stat_part = nullptr, const float* sel_gamma = nullptr, float* sel_val = nullptr,
                int* sel_row = nullptr, bool accumulate = false, const GemmXf* xf = nullptr, int stat_ld = 0, const GemmGroups* grp = nullptr,
                const GemmBs* bs = nullptr, const GemmDy* dy = nullptr) {
    // grp (nullable): block-diagonal product in one launch (GemmArgs groups; M / N are the LAUNCH's dimensions, K one group's).
    // Only on the interior-tile fp32 kernel: MLSP_ERR_UNSUPPORTED otherwise (nothing launched; the caller launches group by group).
    // stat_ld (0: N): C is a column slice of a [M][stat_ld] matrix whose BatchNorm statistics are taken as ONE vector (multi.hip):
    // stat_part points at this slice's first column of the [panels][2][stat_ld] partial rows
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || (!C && !sel_gamma)) return MLSP_ERR_ARG;
    // one tiny dimension (3 coordinates, 3 / 16 outputs): streaming VALU kernels priced against HBM, not MFMA tiles (thin.hip)
    if (grp && (grp->G < 2 || grp->G > 4 || sel_gamma || gbias || tl_call_precision == 1 || (ta && tb))) return MLSP_ERR_UNSUPPORTED;
    // a transform in a block-diagonal launch: split kernel only; the dropout stream is indexed in 32 bits per aligned quad
    if (xf && ((xf->ld & 3) || (xf->col & 3) || (((uintptr_t)xf->scale | (uintptr_t)xf->shift) & 15) || (double)(xf->which == 1 ? M : K) * xf->ld >= 17179869184.0))
        return MLSP_ERR_UNSUPPORTED;
    if (dy && (!dy->y || !dy->coef || (dy->cld & 3) || (((uintptr_t)dy->y | (uintptr_t)dy->coef) & 15) || gbias || sel_gamma || stat_part ||
               (xf && xf->which != 2) || !gemm_dy_supported(ta, tb, M, N, K, A, lda, B, ldb)))
        return MLSP_ERR_UNSUPPORTED;                                         // nothing launched: the caller runs the streaming apply pass
    if (!dy && !grp && !gbias && !stat_part && !sel_gamma && !accumulate && (!xf || tl_call_precision != 1)) {
        const int rc = launch_thin_gemm(st, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, slab, slab_floats, xf, bs);
        if (rc != MLSP_ERR_UNSUPPORTED) return rc;
    }
    if (xf && !gemm_xf_supported(ta, tb, M, N, K, A, lda, B, ldb, xf->which)) return MLSP_ERR_UNSUPPORTED;   // nothing launched: caller materialises
    if (!dy && !grp && !gbias && !stat_part && !sel_gamma && !accumulate && !xf && !bs && !bias && C && gemm_fold64(ta, tb, M, N, K, lda, ldb) &&
        (((uintptr_t)A | (uintptr_t)B) & 15) == 0) {
        const size_t inner = gemm_slab_floats(128, 128, K / 2);
        if (slab && slab_floats >= inner + 128 * 128) {
            float* C2 = slab + inner;
            const size_t used0 = g_prof.used;
            const int rc = launch_gemm(st, true, false, 128, 128, K / 2, A, 128, B, 128, C2, 128, nullptr, nullptr, 0, slab, inner);
            if (rc != MLSP_OK) return rc;
            if (g_prof.used == used0 + 1) {              // the profiler prices the ALGORITHMIC contraction (64 x 64 x K), not the folded one
                auto& r = g_prof.rec[used0];
                g_prof.flop -= 2.0 * 128 * 128.0 * (K / 2) - 2.0 * 64 * 64.0 * K;
                r.M = 64; r.N = 64; r.K = K;
            }
            hipLaunchKernelGGL(gemm_fold64_sum_kernel, dim3(16), dim3(256), 0, st, C2, C, ldc);
            return mlsp_launch_status();
        }
    }
    // per-cloud layers (<= 32 rows, or a 32-deep wgrad): one-pass skinny kernels, no split-K slab (skinny.hip)
    if (!dy && !grp && !gbias && !stat_part && !sel_gamma && !accumulate && !xf && !bs && ((!ta && M <= 32) || (ta && !tb && K <= 32))) {
        const int rc = launch_skinny_gemm(st, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias);
        if (rc != MLSP_ERR_UNSUPPORTED) return rc;
    }
    if (sel_gamma && (!sel_val || !sel_row || gemm_pick_split(M, N, K) != 1)) return MLSP_ERR_ARG;
    if (gbias && rows_per_group <= 0) return MLSP_ERR_ARG;
    GemmArgs p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.gbias = gbias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.rows_per_group = rows_per_group;
    int ns = gemm_pick_split(M, N, K);
    if (ns > 1 && (!slab || slab_floats < (size_t)ns * M * N)) ns = 1;   // no slab: fall back to one pass
    if (accumulate) ns = 1;                                              // beta = 1 lives in the one-pass epilogue
    p.accumulate = accumulate ? 1 : 0;
    p.fast_out = 0;
    if (stat_part && gemm_pick_split(M, N, K) != 1) return MLSP_ERR_ARG;  // caller must check gemm_stat_parts()
    p.stat_part = stat_part; p.stat_ld = stat_ld > 0 ? stat_ld : N;
    p.sel_gamma = sel_gamma; p.sel_val = sel_val; p.sel_row = sel_row;
    p.bs_amax = nullptr; p.dy_amax = 0;
    p.bs_y = nullptr; p.bs_ldy = 0; p.bs_bn = nullptr; p.bs_bnld = 0; p.bs_slope = 1.f; p.bs_thresh = 0; p.bs_ik = 1.f; p.bs_xH = 0; p.bs_ld4 = 0; p.bs_col = 0;
    if (bs) {
        if (ta || tb || xf || bias || gbias || accumulate || stat_part || sel_gamma || !bs->y || !bs->bn || !bs->part || (bs->ld & 3) || (bs->col & 3) ||
            gemm_bs_parts(M, N, K, lda, ldb, ldc) == 0 || (double)M * bs->ld >= 17179869184.0)
            return MLSP_ERR_UNSUPPORTED;
        p.bs_y = bs->y; p.bs_ldy = bs->ldy; p.bs_bn = bs->bn; p.bs_bnld = bs->bnld;
        p.bs_slope = bs->act == 0 ? 1.f : bs->act == 1 ? 0.f : bs->slope; p.bs_thresh = bs->thresh; p.bs_ik = bs->inv_keep;
        p.bs_xH = mix32_host((uint32_t)bs->seed) ^ (uint32_t)(bs->seed >> 32) * 0x9e3779b9U; p.bs_ld4 = bs->ld / 4; p.bs_col = bs->col;
        p.stat_part = bs->part; p.stat_ld = bs->stat_ld; p.bs_amax = bs->amax;
    }
    p.dy_y = dy ? dy->y : nullptr; p.dy_coef = dy ? dy->coef : nullptr; p.dy_cld = dy ? dy->cld : 0; p.dy_amax = (dy && dy->amax) ? 1 : 0;
    p.a_amax = nullptr; p.a_amax_n = 0; p.b_amax[0] = p.b_amax[1] = p.b_amax[2] = p.b_amax[3] = nullptr; p.b_amax_n[0] = p.b_amax_n[1] = p.b_amax_n[2] = p.b_amax_n[3] = 0; p.x_mean = p.x_invstd = nullptr; p.stat_sqrt_rows = 0.f;
    p.x_scale = p.x_shift = nullptr; p.x_act = 0; p.x_slope = 0.f; p.x_thresh = 0; p.x_inv_keep = 1.f; p.x_seed = 0; p.x_ld = 0; p.x_col = 0;
    if (xf) {
        p.x_scale = xf->scale; p.x_shift = xf->shift; p.x_act = xf->act; p.x_thresh = xf->thresh;
        p.x_slope = xf->act == 0 ? 1.f : xf->act == 1 ? 0.f : xf->slope;          // effective negative-side factor (xf_quad)
        p.x_inv_keep = xf->inv_keep; p.x_seed = xf->seed; p.x_ld = xf->ld; p.x_col = xf->col;
    }
    int ktiles = (K + BK - 1) / BK;
    int kts = (ktiles + ns - 1) / ns;
    ns = (ktiles + kts - 1) / kts;
    const int bm = (ns == 1) ? gemm_pick_bm(M, N, K) : 128;
    p.ntm = (M + bm - 1) / bm; p.ntn = (N + BN - 1) / BN;
    p.nsplit = ns; p.ksplit = kts * BK;
    p.a_vec = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
    p.b_vec = (ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);
    if (ns > 1) { p.C = slab; p.ldc = N; }
    p.xcd_map = p.ntm >= 16 && p.ntn > 1;
    dim3 grid(p.xcd_map ? ((p.ntm + 7) / 8) * 8 * p.ntn : p.ntm * p.ntn, ns);
    static const bool no_xcd2 = getenv("MLSP_GEMM_NO_XCD2") != nullptr;       // read-once A/B switch (tools/ab)
    bool xcd2 = false;                                                        // (decided below, once the kernel is known: split kernel only)
    const bool prof = g_prof.on && g_prof.used < PROF_MAX_PAIRS;
    // FAST: every tile interior (M, N, K-range multiples of the tile), 16-byte loads legal on both operands
    // (byte offsets inside an operand tile's K range are 32-bit buffer offsets)
    const long a_span = ta ? (long)p.ksplit * lda * 4 : 128L * lda * 4 + (long)p.ksplit * 4;
    const long b_span = !tb ? (long)p.ksplit * ldb * 4 : 128L * ldb * 4 + (long)p.ksplit * 4;
    const bool fast = p.a_vec && p.b_vec && (M % bm == 0) && (N % BN == 0) && (K % BK == 0) && a_span < (1L << 31) - 4096 && b_span < (1L << 31) - 4096;
    static const bool old_epilogue = getenv("MLSP_GEMM_OLD_EPILOGUE") != nullptr;       // read-once A/B switch (tools/ab)
    // lean output pass: every row of a tile takes the same per-cloud bias row, byte offsets inside a wave's region fit 31 bits
    p.fast_out = (fast && (!gbias || rows_per_group % bm == 0) && (long)p.ldc * 4 * 64 < (1L << 30) && !old_epilogue) ? 1 : 0;
    p.gmode = 0; p.gtiles = 1; p.a_gs = p.b_gs = 0; p.Bg[0] = p.Bg[1] = p.Bg[2] = p.Bg[3] = B;
    if (grp) {
        const int per = (grp->mode == 1 ? N : M) / grp->G;                  // columns (mode 1) / rows (mode 2) of one group
        const int tile = grp->mode == 1 ? BN : bm;
        bool ok = fast && tl_call_precision != 1 && (grp->mode == 1 || grp->mode == 2) && per * grp->G == (grp->mode == 1 ? N : M) && per % tile == 0;
        for (int g = 0; g < grp->G && ok && grp->mode == 1; ++g) ok = grp->Bg[g] && (((uintptr_t)grp->Bg[g] & 15) == 0);
        if (!ok) return MLSP_ERR_UNSUPPORTED;
        p.gmode = grp->mode; p.gtiles = per / tile; p.a_gs = grp->a_gs; p.b_gs = grp->b_gs;
        for (int g = 0; g < 4; ++g) p.Bg[g] = grp->mode == 1 ? grp->Bg[g < grp->G ? g : 0] : B;
    }
    const bool xf_split = xf && fast && tl_call_precision == 2 && gemm_split_pays(M, N, kts) && (xf->which != 1 || kts * BK <= SX_XF_KMAX);
    if (grp && xf && !xf_split) return MLSP_ERR_UNSUPPORTED;             // (the fp32 transform kernels take no groups: nothing launched)
    // split-K launches of gemm_split_kernel with few row panels (weight gradients): XCD-grouped (split, panel) order, see the kernel
    xcd2 = !no_xcd2 && !p.xcd_map && ns > 1 && p.ntn > 1 && (p.ntm * ns) % 8 == 0 && fast && tl_call_precision == 2 && gemm_split_pays(M, N, kts) &&
           (!xf || xf_split);
    static const bool no_xcd3 = getenv("MLSP_GEMM_NO_XCD3") != nullptr;       // read-once A/B switch (tools/ab)
    const bool xcd3 = !no_xcd3 && !p.xcd_map && ns > 1 && ns % 8 == 0 && p.ntm > 1 && fast && tl_call_precision == 2 && gemm_split_pays(M, N, kts) && (!xf || xf_split);
    if (xcd3) { p.xcd_map = 3; grid = dim3(p.ntm * p.ntn * ns, 1); }       // (all tiles of a split on one XCD: see the kernel)
    else if (xcd2) { p.xcd_map = 2; grid = dim3(p.ntm * p.ntn * ns, 1); }
    // (mode 1, N = 64: the bf16 kernel's tiles are 128 columns wide, so these launches used to fall to the bounds-checked f32 kernel at half-empty
    // tiles -- 33 us where the 64-column f32 kernel takes 12; exact products are within what mode 1 promises.  Read-once A/B switch.)
    static const bool no_n64_bf16 = getenv("MLSP_GEMM_NO_N64_BF16") != nullptr;
    const bool n64 = !dy && !grp && !xf && N == 64 && p.a_vec && p.b_vec && M % 128 == 0 && K % BK == 0 && !(ta && tb) && !gbias && (!stat_part || (bm == 128 && ns == 1)) && !sel_gamma &&
                     (!bias || (ns == 1 && (((uintptr_t)bias) & 3) == 0)) && (tl_call_precision != 1 || !no_n64_bf16) && (ns == 1 || p.ldc == N);
    // two-piece f16 products (mode 3) on this launch?  The split kernel, and a bound for both operands: partial maxima of the operands as
    // they lie in memory (measured by ONE streaming launch here, or earlier in this API call), the analytic bound for a transformed one
    // (batch statistics at hand).  Anything missing: the three-piece bf16 products (same kernel family, same accuracy class).
    bool half = false;
    if (tl_split_half && tl_amax.base && fast && tl_call_precision == 2 && gemm_split_pays(M, N, kts) && !n64 && (!xf || xf_split)) {
        AmaxBatch batch;
        bool ok = true;
        // Does MEASURING what is not at hand pay?  The three-product kernel saves about a third of the six-product launch (~150 TF on
        // these shapes); a measuring pass streams the operand at ~3 TB/s plus a launch.  Long K loops win easily (conv5: 76 us against
        // 25 for its 67 MB input); the set-abstraction layers' K = 128 contractions over 262144 rows do not (tools/x6/lib_bench: 90 us with
        // the pass against 70 on six products; configs[3] 3.58 -> 3.46 ms) and stay on the six products unless their bounds are free.
        {
            static const bool always = getenv("MLSP_AMAX_ALWAYS") != nullptr;          // read-once A/B switch
            double mbytes = 0.0;
            auto need = [&](const float* X, long rows, int cols, int ld) { if (!amax_at_hand(X, rows, cols, ld)) mbytes += (double)rows * cols * 4.0; };
            if (!(xf && xf->which == 1) && !(dy && dy->amax))
                need(A, ta ? K : M, (ta ? M : K) + ((grp && grp->mode == 1) ? (int)((grp->G - 1) * grp->a_gs) : 0), lda);
            if (xf && xf->which == 2) { }
            else if (grp && grp->mode == 1) { for (int g = 0; g < grp->G; ++g) need(grp->Bg[g], tb ? N / grp->G : K, tb ? K : N / grp->G, ldb); }
            else need(B, tb ? N : K, (tb ? K : N) + ((grp && grp->mode == 2) ? (int)((grp->G - 1) * grp->b_gs) : 0), ldb);
            const double gain_us = 0.33 * (2.0 * M * N * K) / 150e6, cost_us = 3.0 + mbytes / 3e6;
            if (mbytes > 0.0 && gain_us < cost_us && !always) ok = false;
        }
        if (!ok) { }
        else if (xf && xf->which == 1) ok = xf->mean && xf->invstd;
        else if (dy && dy->amax) { /* bound from the coefficient rows (max |d'| per channel, left by the consumers' dgrads): nothing to measure */ }
        else {
            const int acols = (ta ? M : K) + ((grp && grp->mode == 1) ? (int)((grp->G - 1) * grp->a_gs) : 0);
            p.a_amax = amax_get(batch, A, ta ? K : M, acols, lda, &p.a_amax_n);
            ok = p.a_amax != nullptr;
        }
        if (!ok) { }
        else if (xf && xf->which == 2) ok = ok && xf->mean && xf->invstd;
        else if (grp && grp->mode == 1) {
            for (int g = 0; g < grp->G && ok; ++g) { p.b_amax[g] = amax_get(batch, grp->Bg[g], tb ? N / grp->G : K, tb ? K : N / grp->G, ldb, &p.b_amax_n[g]); ok = p.b_amax[g] != nullptr; }
        } else {
            const int bcols = (tb ? K : N) + ((grp && grp->mode == 2) ? (int)((grp->G - 1) * grp->b_gs) : 0);
            p.b_amax[0] = amax_get(batch, B, tb ? N : K, bcols, ldb, &p.b_amax_n[0]);
            ok = ok && p.b_amax[0] != nullptr;
        }
        amax_flush(st, batch);                 // (whatever was queued is in the cache: measure it even if this launch ends up on the bf16 pieces)
        if (xf) { p.x_mean = xf->mean; p.x_invstd = xf->invstd; }
        p.stat_sqrt_rows = sqrtf((float)(ta ? K : M));
        half = ok;
    }
    if (prof) (void)hipEventRecord(g_prof.ev[2 * g_prof.used], st);
    if (n64) {
        dim3 g64(M / 128, ns);
        if (!ta && tb) hipLaunchKernelGGL((gemm_f32_n64_kernel<false, true>), g64, dim3(256), 0, st, p);
        else if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_n64_kernel<false, false>), g64, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_f32_n64_kernel<true, false>), g64, dim3(256), 0, st, p);
    } else
#define GEMM_GO(TA_, TB_, WM_) do { if (fast && tl_call_precision == 1) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, TB_, WM_, false, false, false>), grid, dim3(256), 0, st, p); \
                                     else if (half) hipLaunchKernelGGL((gemm_split_kernel<TA_, TB_, WM_, 0, false, false, 2>), grid, dim3(256), 0, st, p); \
                                     else if (fast && tl_call_precision == 2 && gemm_split_pays(M, N, kts)) hipLaunchKernelGGL((gemm_split_kernel<TA_, TB_, WM_>), grid, dim3(256), 0, st, p); \
                                     else if (fast) hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_, WM_, true>), grid, dim3(256), 0, st, p); \
                                     else hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_, WM_, false>), grid, dim3(256), 0, st, p); } while (0)
    if (dy) {                                                            // dual A operand (validated above: split kernel, interior tiles)
        if (!fast || (xf && !xf_split) || (grp && grp->mode != (ta ? 2 : 1))) return MLSP_ERR_UNSUPPORTED;
#define SPLIT_DY_GO(TA_, XF_, XD_) do { if (half) { if (bm == 128) hipLaunchKernelGGL((gemm_split_kernel<TA_, false, 2, XF_, XD_, true, 2>), grid, dim3(256), 0, st, p); \
                                                    else hipLaunchKernelGGL((gemm_split_kernel<TA_, false, 1, XF_, XD_, true, 2>), grid, dim3(256), 0, st, p); } \
                                        else if (bm == 128) hipLaunchKernelGGL((gemm_split_kernel<TA_, false, 2, XF_, XD_, true>), grid, dim3(256), 0, st, p); \
                                        else hipLaunchKernelGGL((gemm_split_kernel<TA_, false, 1, XF_, XD_, true>), grid, dim3(256), 0, st, p); } while (0)
        if (!ta) { if (xf) return MLSP_ERR_UNSUPPORTED; SPLIT_DY_GO(false, 0, false); }
        else if (!xf) SPLIT_DY_GO(true, 0, false);
        else if (xf->thresh) SPLIT_DY_GO(true, 2, true);
        else SPLIT_DY_GO(true, 2, false);
#undef SPLIT_DY_GO
    } else if (xf_split) {                                               // operand transform on the split kernel, plain or block-diagonal
#define SPLIT_XF_GO(TA_, TB_, XF_) do { if (half) { if (bm == 128) { if (xf->thresh) hipLaunchKernelGGL((gemm_split_kernel<TA_, TB_, 2, XF_, true, false, 2>), grid, dim3(256), 0, st, p); \
                                                                      else hipLaunchKernelGGL((gemm_split_kernel<TA_, TB_, 2, XF_, false, false, 2>), grid, dim3(256), 0, st, p); } \
                                                     else { if (xf->thresh) hipLaunchKernelGGL((gemm_split_kernel<TA_, TB_, 1, XF_, true, false, 2>), grid, dim3(256), 0, st, p); \
                                                            else hipLaunchKernelGGL((gemm_split_kernel<TA_, TB_, 1, XF_, false, false, 2>), grid, dim3(256), 0, st, p); } } \
                                         else if (bm == 128) { if (xf->thresh) hipLaunchKernelGGL((gemm_split_kernel<TA_, TB_, 2, XF_, true>), grid, dim3(256), 0, st, p); \
                                                          else hipLaunchKernelGGL((gemm_split_kernel<TA_, TB_, 2, XF_, false>), grid, dim3(256), 0, st, p); } \
                                         else { if (xf->thresh) hipLaunchKernelGGL((gemm_split_kernel<TA_, TB_, 1, XF_, true>), grid, dim3(256), 0, st, p); \
                                                else hipLaunchKernelGGL((gemm_split_kernel<TA_, TB_, 1, XF_, false>), grid, dim3(256), 0, st, p); } } while (0)
        if (xf->which == 1 && !ta && tb && (!grp || grp->mode == 1)) SPLIT_XF_GO(false, true, 1);
        else if (xf->which == 2 && ta && !tb && (!grp || grp->mode == 2)) SPLIT_XF_GO(true, false, 2);
        else return MLSP_ERR_UNSUPPORTED;
#undef SPLIT_XF_GO
    } else if (grp && tl_call_precision == 2 && gemm_split_pays(M, N, kts)) {   // block-diagonal launch on the split kernel (groups are a run-time argument there)
        if (grp->mode == 1 && !ta && tb) { if (bm == 128) GEMM_GO(false, true, 2); else GEMM_GO(false, true, 1); }
        else if (grp->mode == 1 && !ta && !tb) { if (bm == 128) GEMM_GO(false, false, 2); else GEMM_GO(false, false, 1); }
        else if (grp->mode == 2 && ta && !tb) { if (bm == 128) GEMM_GO(true, false, 2); else GEMM_GO(true, false, 1); }
        else return MLSP_ERR_UNSUPPORTED;
    } else if (grp) {                                     // block-diagonal launch (validated above: fast, fp32)
        if (grp->mode == 1 && !ta && tb) { if (bm == 128) hipLaunchKernelGGL((gemm_f32_kernel<false, true, 2, true, 0, 1>), grid, dim3(256), 0, st, p);
                                           else hipLaunchKernelGGL((gemm_f32_kernel<false, true, 1, true, 0, 1>), grid, dim3(256), 0, st, p); }
        else if (grp->mode == 1 && !ta && !tb) { if (bm == 128) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 2, true, 0, 1>), grid, dim3(256), 0, st, p);
                                                 else hipLaunchKernelGGL((gemm_f32_kernel<false, false, 1, true, 0, 1>), grid, dim3(256), 0, st, p); }
        else if (grp->mode == 2 && ta && !tb) { if (bm == 128) hipLaunchKernelGGL((gemm_f32_kernel<true, false, 2, true, 0, 2>), grid, dim3(256), 0, st, p);
                                                else hipLaunchKernelGGL((gemm_f32_kernel<true, false, 1, true, 0, 2>), grid, dim3(256), 0, st, p); }
        else return MLSP_ERR_UNSUPPORTED;
    } else if (xf) {                                      // operand transform: FAST fp32 instantiations only (gemm_xf_supported)
        if (xf->which == 1 && tb) {
            if (bm == 128) hipLaunchKernelGGL((gemm_f32_kernel<false, true, 2, true, 1>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_f32_kernel<false, true, 1, true, 1>), grid, dim3(256), 0, st, p);
        } else if (xf->which == 1) {
            if (bm == 128) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 2, true, 1>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_f32_kernel<false, false, 1, true, 1>), grid, dim3(256), 0, st, p);
        } else {
            if (bm == 128) hipLaunchKernelGGL((gemm_f32_kernel<true, false, 2, true, 2>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_f32_kernel<true, false, 1, true, 2>), grid, dim3(256), 0, st, p);
        }
    } else if (bm == 128) {
        if (!ta && tb) GEMM_GO(false, true, 2);
        else if (!ta && !tb) GEMM_GO(false, false, 2);
        else if (ta && !tb) GEMM_GO(true, false, 2);
        else GEMM_GO(true, true, 2);
    } else {
        if (!ta && tb) GEMM_GO(false, true, 1);
        else if (!ta && !tb) GEMM_GO(false, false, 1);
        else if (ta && !tb) GEMM_GO(true, false, 1);
        else GEMM_GO(true, true, 1);
    }
#undef GEMM_GO
    if (prof) {
        (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], st);
        const bool on_split = fast && tl_call_precision == 2 && gemm_split_pays(M, N, kts) && !n64 && (!xf || xf_split);
        g_prof.rec[g_prof.used] = {M, N, K, ta, tb, ns, bm, (stat_part ? 1 : 0) + (sel_gamma ? 2 : 0), on_split ? (half ? 2 : 1) : 0, (dy ? 1 : 0) + (accumulate ? 2 : 0)};
        g_prof.used++;
        g_prof.flop += 2.0 * M * (double)N * K;
        g_prof.bytes += 4.0 * ((double)M * K + (double)K * N + (C ? (double)M * N : 0.0));
    }
    if (ns > 1) launch_splitk_reduce_any(st, slab, C, M, N, ldc, ns, bias, gbias, rows_per_group);
    return mlsp_launch_status();
}


// ---- bf16 activation storage (BASELINE.json configs[4]) ---------------------------------------------------------------------
// The same contraction with operands and/or the output held as bf16 in HBM: activations X / Y / Z and their gradients are bf16,
// weights, biases, BatchNorm statistics and weight gradients stay fp32.  Runs on gemm_bf16_kernel (v_mfma_f32_32x32x16_bf16, fp32
// accumulation); a bf16 operand is copied to LDS as it is (half the HBM bytes, no conversion), an fp32 one is rounded on the way.
// Only interior-tile shapes (M % tile, N % 128, K % 32 == 0, 16-byte aligned rows): MLSP_ERR_UNSUPPORTED otherwise, and the caller keeps
// that layer in fp32.  Same BN-statistics / bias / per-cloud-bias / beta = 1 epilogues; the statistics come from the fp32 accumulators.
int launch_gemm_mx(hipStream_t st, bool ta, bool tb, int M, int N, int K, const void* A, int a_bf16, int lda, const void* B, int b_bf16,
                   int ldb, void* C, int c_bf16, int ldc, const float* bias, const float* gbias, int rows_per_group, float* slab,
                   size_t slab_floats, double* stat_part, bool accumulate) {
    if (!a_bf16 && !b_bf16 && !c_bf16)
        return launch_gemm(st, ta, tb, M, N, K, (const float*)A, lda, (const float*)B, ldb, (float*)C, ldc, bias, gbias, rows_per_group,
                           slab, slab_floats, stat_part, nullptr, nullptr, nullptr, accumulate);
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return MLSP_ERR_ARG;
    if (gbias && rows_per_group <= 0) return MLSP_ERR_ARG;
    GemmArgs p;
    p.A = (const float*)A; p.B = (const float*)B; p.C = (float*)C; p.bias = bias; p.gbias = gbias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.rows_per_group = rows_per_group;
    int ns = gemm_pick_split(M, N, K);
    if (ns > 1 && (!slab || slab_floats < (size_t)ns * M * N)) ns = 1;
    if (accumulate) ns = 1;
    if (ns > 1 && c_bf16) return MLSP_ERR_UNSUPPORTED;                   // a split result is reduced in fp32
    if (stat_part && ns != 1) return MLSP_ERR_ARG;
    p.accumulate = accumulate ? 1 : 0; p.c_bf16 = c_bf16; p.fast_out = 0;
    p.stat_part = stat_part; p.stat_ld = N; p.sel_gamma = nullptr; p.sel_val = nullptr; p.sel_row = nullptr;
    const int ktiles = (K + BK - 1) / BK;
    const int kts = (ktiles + ns - 1) / ns;
    ns = (ktiles + kts - 1) / kts;
    const int bm = (ns == 1) ? gemm_pick_bm(M, N, K) : 128;
    p.ntm = (M + bm - 1) / bm; p.ntn = (N + BN - 1) / BN;
    p.nsplit = ns; p.ksplit = kts * BK;
    const bool a_ok = a_bf16 ? (lda % 8 == 0) : (lda % 4 == 0), b_ok = b_bf16 ? (ldb % 8 == 0) : (ldb % 4 == 0);
    p.a_vec = a_ok && (((uintptr_t)A & 15) == 0);
    p.b_vec = b_ok && (((uintptr_t)B & 15) == 0);
    const bool nedge = (N % BN != 0);
    if (nedge && !(!tb && !b_bf16 && N % 4 == 0 && !stat_part && !(c_bf16 && ns > 1))) return MLSP_ERR_UNSUPPORTED;   // k-major fp32 B only
    if (!(p.a_vec && p.b_vec && M % bm == 0 && K % BK == 0)) return MLSP_ERR_UNSUPPORTED;
    if (ns > 1) { p.C = slab; p.ldc = N; }
    p.xcd_map = p.ntm >= 16 && p.ntn > 1;
    dim3 grid(p.xcd_map ? ((p.ntm + 7) / 8) * 8 * p.ntn : p.ntm * p.ntn, ns);
    const bool prof = g_prof.on && g_prof.used < PROF_MAX_PAIRS;
    if (prof) (void)hipEventRecord(g_prof.ev[2 * g_prof.used], st);
    const int code = (a_bf16 ? 4 : 0) | (b_bf16 ? 2 : 0) | ((c_bf16 && ns == 1) ? 1 : 0);
#define MX_GO(TA_, TB_, WM_, A_, B_, C_) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, TB_, WM_, A_, B_, C_>), grid, dim3(256), 0, st, p)
#define MX_ACT_W(TA_, TB_, WM_) do { /* activation x weight: A in {f32, bf16}, B = fp32 weights, C in {f32, bf16} */ \
        if (code == 0) MX_GO(TA_, TB_, WM_, false, false, false); else if (code == 1) MX_GO(TA_, TB_, WM_, false, false, true); \
        else if (code == 4) MX_GO(TA_, TB_, WM_, true, false, false); else if (code == 5) MX_GO(TA_, TB_, WM_, true, false, true); \
        else return MLSP_ERR_UNSUPPORTED; } while (0)
#define MX_ACT_ACT(WM_) do { /* wgrad: A = dY, B = X in {f32, bf16}, C = fp32 */ \
        if (code == 0) MX_GO(true, false, WM_, false, false, false); else if (code == 4) MX_GO(true, false, WM_, true, false, false); \
        else if (code == 2) MX_GO(true, false, WM_, false, true, false); else if (code == 6) MX_GO(true, false, WM_, true, true, false); \
        else return MLSP_ERR_UNSUPPORTED; } while (0)
    if (ta && tb) return MLSP_ERR_UNSUPPORTED;
    if (nedge) {            // ragged channel count on the N side (192-channel head input): dgrad dX = dY * W and wgrad dW = dY^T * X, B in fp32
#define MX_NE(TA_, WM_) do { if (code == 0) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, false, WM_, false, false, false, true>), grid, dim3(256), 0, st, p); \
        else if (code == 1) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, false, WM_, false, false, true, true>), grid, dim3(256), 0, st, p); \
        else if (code == 4) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, false, WM_, true, false, false, true>), grid, dim3(256), 0, st, p); \
        else if (code == 5) hipLaunchKernelGGL((gemm_bf16_kernel<TA_, false, WM_, true, false, true, true>), grid, dim3(256), 0, st, p); \
        else return MLSP_ERR_UNSUPPORTED; } while (0)
        if (ta) { if (bm == 128) MX_NE(true, 2); else MX_NE(true, 1); }
        else { if (bm == 128) MX_NE(false, 2); else MX_NE(false, 1); }
#undef MX_NE
    } else if (bm == 128) {
        if (!ta && tb) MX_ACT_W(false, true, 2); else if (!ta && !tb) MX_ACT_W(false, false, 2); else MX_ACT_ACT(2);
    } else {
        if (!ta && tb) MX_ACT_W(false, true, 1); else if (!ta && !tb) MX_ACT_W(false, false, 1); else MX_ACT_ACT(1);
    }
#undef MX_ACT_ACT
#undef MX_ACT_W
#undef MX_GO
    if (prof) {
        (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], st);
        g_prof.rec[g_prof.used] = {M, N, K, ta, tb, ns, bm, (stat_part ? 1 : 0) + 4, 0, 0};
        g_prof.used++;
        g_prof.flop += 2.0 * M * (double)N * K;
        g_prof.bytes += (a_bf16 ? 2.0 : 4.0) * M * K + (b_bf16 ? 2.0 : 4.0) * K * N + (c_bf16 ? 2.0 : 4.0) * M * N;
    }
    if (ns > 1) launch_splitk_reduce_any(st, slab, (float*)C, M, N, ldc, ns, bias, gbias, rows_per_group);
    return mlsp_launch_status();
}
