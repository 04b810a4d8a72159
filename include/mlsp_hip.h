/* libmlsp_hip.so -- C ABI of the MI355X (gfx950) hot path of MLSP.
 *
 * The reference (VITA-Group/MLSP) has no FFI/operator interface of its own: its boundary is the
 * Python nn.Module / free-function surface of PointDA/Models.py, PointDA/model_utils.py and
 * MLSP/mlsp.py.  This header is the native boundary underneath the Python
 * mirror of that surface (mlsp_amd/Models.py, model_utils.py, mlsp.py): each entry point names
 * the reference code it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors); the library never
 *     allocates or frees caller-visible memory.  Scratch is passed in (`ws`, `ws_bytes`; size it
 *     with mlsp_workspace_bytes).
 *   - every compute call is asynchronous on `stream`, re-entrant, never throws, and returns 0 on
 *     success, a negative MLSP_ERR_* code for bad arguments, or a positive hipError_t.  The compute
 *     entry points keep no mutable state: what a call does is a function of its arguments alone, from any thread, on any stream
 *     (ABI v8: the products of the GEMM family are the per-call `precision` argument below, not a library switch; ABI v13:
 *     mlsp_operand_bounds_next hands ONE following call an optional table through a thread-local slot that the call empties -- an
 *     out-of-band argument of that call, not state).  The only process-wide object is the measurement hook mlsp_profile_begin/_end (HIP events around launches while armed; bench.py only,
 *     never armed in a production step).  Environment variables read once per process, all A/B measurement switches:
 *     MLSP_TNET_BWD_OLD=1 (round-1 T-Net backward kernel), MLSP_GEMM_SPLIT_ALWAYS, MLSP_GEMM_OLD_EPILOGUE.
 *   - `precision` (every entry point that reaches a matrix-core contraction takes it, just before its workspace):
 *       MLSP_PREC_F32    0  f32 MFMA, exact fp32 products (v_mfma_f32_32x32x2_f32);
 *       MLSP_PREC_BF16   1  operands ROUNDED to bf16, fp32 accumulation (BASELINE.json configs[4]; reduced precision);
 *       MLSP_PREC_BF16X6 2  fp32-ACCURATE products on the bf16 matrix cores: every operand value split exactly into three bf16 pieces
 *                           (8 + 8 + 8 significand bits), six piece products per multiply, fp32 accumulation.  Against float64 its error
 *                           is below the f32-MFMA chain's (tests/test_gpu_kernels.py::test_gemm_split_bf16_accuracy), 1.55-1.65x faster
 *                           on the 32768-row layers.  The mode of every number bench.py reported in rounds 3-5 (its `bf16x6_split`
 *                           leg since).  An infinite operand becomes NaN (the f32 MFMA would give +-inf).
 *       MLSP_PREC_F16X3  3  (ABI v13) the same accuracy class with HALF the matrix work: two f16 pieces per operand value under a
 *                           per-workgroup power-of-two scale, three piece products (see the definition below).  THE MODE OF EVERY NUMBER
 *                           bench.py REPORTS since round 6 and the default of the Python mirror (MLSP_GEMM_PRECISION overrides it there).
 *     Launches off the interior-tile path, short K loops, operand-transform and N = 64 launches run on the f32 kernels in every mode
 *     (exact fp32 either way); kNN distances, BatchNorm statistics, reductions and losses are fp32 in every mode.  A backward call
 *     normally passes the precision of its forward (the Python mirror keeps it in the autograd context); nothing breaks if it differs.
 *   - activations are POINT-major fp32 row matrices: [rows][C] with rows = B*N points (or
 *     B*N*k edges), channels contiguous.  The reference itself moves to this layout before its
 *     gather (model_utils.py:35).  Indices are int32, local to their cloud (0..N-1).
 *   - act: 0 none, 1 ReLU, 2 LeakyReLU(slope).
 *   - bn_save is [4][C] floats: scale (= gamma*invstd), shift (= beta - mean*scale), mean, invstd.
 */
#ifndef MLSP_HIP_H
#define MLSP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mlsp_stream_t; /* == hipStream_t */

#define MLSP_ABI_VERSION 13
#define MLSP_OK 0
#define MLSP_ERR_ARG (-1)
#define MLSP_ERR_WORKSPACE (-2)
#define MLSP_ERR_UNSUPPORTED (-3)
#define MLSP_PREC_F32 0
#define MLSP_PREC_BF16 1
#define MLSP_PREC_BF16X6 2
/* ABI v13.  fp32-accurate products on the f16 matrix cores with HALF the matrix work of MLSP_PREC_BF16X6: every fp32 operand value, times a
 * per-workgroup power of two, is split into two f16 pieces (11 + 11 significand bits) and a product is three piece products with fp32
 * accumulation.  The scale comes from a bound of the operand's magnitude: measured (one streaming launch per operand and call, partial
 * maxima in the last 64 KiB of the call's workspace) or analytic (an operand that is transformed by batch-statistics BatchNorm on the fly);
 * a launch without a bound (no workspace tail, eval-mode transform) runs the six-product bf16 split instead.  Same accuracy class as
 * MLSP_PREC_BF16X6 (tests/test_gpu_kernels.py::test_gemm_split_bf16_accuracy holds both to the same bar against float64). */
#define MLSP_PREC_F16X3 3
/* Caller-owned operand bounds for MLSP_PREC_F16X3 (optional; ABI v13).  partials: 256 floats on the device -- partial maxima of |X| over X
 * [rows][cols] (row pitch ld, fp32, 16-byte aligned, cols % 4 == 0).  mlsp_operand_bounds_next(tab, n) hands `n` entries to the NEXT entry
 * point this thread calls that takes a `precision` (and to that call only; the table must stay alive until it returns): a GEMM operand of
 * that call with the same pointer and shape uses the entry's partials instead of measuring into the workspace -- as they are when
 * valid != 0; when valid == 0 the call measures INTO them (its one measuring launch) and sets valid = 1 in the caller's table.  The caller
 * keeps such a buffer for as long as the tensor's contents do not change: an activation that several layers read, a weight that its
 * layer's backward reads again.  A VALID entry may hold any number n <= 4096 of partials from any source whose maximum bounds |X| -- e.g. the
 * per-tile maxima mlsp_adam_flat_f32 leaves of the parameters it just updated.  Purely an optimisation: without it every call measures
 * what it needs. */
typedef struct { const float* ptr; long rows; int cols, ld; float* partials; int valid; int n; } mlsp_bound_t;   /* n: partials held (0 = 256); an entry to be measured (valid == 0) holds 256 */
int mlsp_operand_bounds_next(mlsp_bound_t* tab, int n);

int mlsp_abi_version(void);
const char* mlsp_strerror(int code);

/* Upper bound of the scratch bytes any entry point needs for a problem with `rows` activation
 * rows, `cin`/`cout` channels (pass the largest of each used with one workspace). */
size_t mlsp_workspace_bytes(int rows, int cin, int cout);

/* knn(x,k): PointDA/model_utils.py:9-16 (twin PointSegDA/Models.py:8-15).
 * x [B][N] rows of C floats with row pitch ldx; idx [B][N][k], nearest first (canonical arithmetic,
 * see oracle/knn_canon.c).  If rev_off != NULL also builds the reverse neighbour index used by the
 * backward passes: rev_off [B*N+1], rev_ent [B*N*k] packed (i_local << 8 | slot), sorted. */
int mlsp_knn_f32(const float* x, int ldx, int B, int N, int C, int k, int32_t* idx, int32_t* rev_off, int32_t* rev_ent,
                 void* ws, size_t ws_bytes, mlsp_stream_t stream);

/* reverse neighbour index alone, for caller-provided indices (get_graph_feature(..., idx=...)) */
int mlsp_knn_reverse(const int32_t* idx, int B, int N, int k, int32_t* rev_off, int32_t* rev_ent, mlsp_stream_t stream);

/* get_graph_feature(x,args,k,idx): PointDA/model_utils.py:18-42, materialised edge-major:
 * F [B*N*k][2C] = [x_j - x_i ; x_i].  (The reference's [B,2C,N,k] result is a permuted view of this.) */
int mlsp_graph_feature_fwd_f32(const float* x, const int32_t* idx, int B, int N, int C, int k, float* F, mlsp_stream_t stream);
int mlsp_graph_feature_bwd_f32(const float* dF, const int32_t* rev_off, const int32_t* rev_ent, int B, int N, int C, int k,
                               float* dx, mlsp_stream_t stream);

/* Fused EdgeConv block = get_graph_feature + conv_2d (1x1 Conv2d bias=False + BatchNorm2d + act) + max over k:
 * PointDA/model_utils.py:18-63 with PointDA/Models.py:115-129.  Algebraically folded (edge.hip).
 * W [Cout][2C] in the reference's Conv2d layout.  Saved for backward: uv [P][2Cout], msel [P][Cout],
 * argsel [P][Cout] u8, s1 [P][Cout], bn_save [4][Cout].  out (row pitch ldo >= Cout) and dOut (row pitch lddo) may be column
 * slices of a wider matrix: the four EdgeConv outputs are written straight into the [P][512] concatenation of
 * PointDA/Models.py:131 (no torch.cat pass) and its gradient is read in place.
 * Wd [2Cout][C] (nullable): the folded weight [Wa ; Wb - Wa] the forward builds anyway; a caller that keeps it and hands it to the
 * backward saves the rebuild.  dx (row pitch lddx >= C) may be a column slice as well, and with dx_accumulate != 0 the input
 * gradient is ADDED to it: layer l+1 adds its input gradient into layer l's slice of the concatenation's gradient, which is the
 * sum autograd would otherwise form with a separate pass. */
int mlsp_edgeconv_fwd_f32(const float* x, int ldx, const int32_t* idx, const float* W, const float* gamma, const float* beta,
                          float* run_mean, float* run_var, float momentum, float eps, int act, float slope, int training,
                          int B, int N, int C, int Cout, int k, float* out, int ldo, float* uv, float* msel, uint8_t* argsel,
                          float* s1, float* bn_save, float* Wd, int precision, void* ws, size_t ws_bytes, mlsp_stream_t stream);
int mlsp_edgeconv_bwd_f32(const float* dOut, int lddo, const float* x, int ldx, const int32_t* rev_off, const int32_t* rev_ent,
                          const float* W, const float* out, int ldo, const float* uv, const float* msel, const uint8_t* argsel,
                          const float* s1, const float* bn_save, const float* Wd, int act, float slope, int training, int B, int N,
                          int C, int Cout, int k, float* dx, int lddx, int dx_accumulate, float* dW, float* dgamma, float* dbeta,
                          int precision, void* ws, size_t ws_bytes, mlsp_stream_t stream);

/* Fused per-edge stage of the T-Net (PointDA/model_utils.py:111-115: conv2d1 6->64, conv2d2 64->128 per EDGE, max over k),
 * dgcnn branch (bias-free convs, LeakyReLU).  x [P][C] point-major (C = 3), W1 [C1][2C], W2 [C2][C1]; requires C1 = 64, C2 = 128
 * (returns MLSP_ERR_UNSUPPORTED otherwise: use graph_feature + pointmlp + segmax).  out [P][C2].
 * Saved for backward: uv [P][2*C1], s1 [P][C1], bn1_save [4][C1], zsel [P][C2], argsel [P][C2] u8, bn2_save [4][C2]. */
int mlsp_tnet_edge_fwd_f32(const float* x, int ldx, const int32_t* idx, const float* W1, const float* gamma1, const float* beta1,
                           float* run_mean1, float* run_var1, const float* W2, const float* gamma2, const float* beta2,
                           float* run_mean2, float* run_var2, float momentum, float eps, float slope, int training, int B, int N,
                           int C, int C1, int C2, int k, float* out, float* uv, float* s1, float* bn1_save, float* zsel,
                           uint8_t* argsel, float* bn2_save, int precision, void* ws, size_t ws_bytes, mlsp_stream_t stream);
int mlsp_tnet_edge_bwd_f32(const float* dOut, const float* x, int ldx, const int32_t* idx, const int32_t* rev_off,
                           const int32_t* rev_ent, const float* W1, const float* W2, const float* out, const float* uv,
                           const float* s1, const float* bn1_save, const float* zsel, const uint8_t* argsel, const float* bn2_save,
                           float slope, int training, int B, int N, int C, int C1, int C2, int k, float* dx, float* dW1,
                           float* dgamma1, float* dbeta1, float* dW2, float* dgamma2, float* dbeta2, int precision, void* ws, size_t ws_bytes,
                           mlsp_stream_t stream);

/* Per-row MLP layer = Linear/1x1 conv (+bias) + BatchNorm + act + dropout:
 * conv_2d / fc_layer (model_utils.py:45-87), conv5+bn5 (Models.py:132), head layers (Models.py:192-196,
 * 226-230, 272-279).  W [Cout][Cin] with row pitch ldw (a column slice of a wider weight is legal).  gbias [G][Cout] (nullable) is a per-row-group bias (row r uses group r / rows_per_group):
 * the x5-repeat half of the heads' 1536-channel input enters as a per-cloud bias (Models.py:156-160).
 * Y = pre-BN (saved), Z = output.  gamma == NULL: no BN (Y is not written, Z = act(linear)).
 * Backward: dx_accumulate != 0 adds the input gradient to what dX already holds (beta = 1 in the dgrad epilogue): the four consumers
 * of the concatenated encoder features (conv5 and the three heads, Models.py:132,156-160) sum their gradients in one buffer instead of
 * three 67 MB element-wise adds.
 * Chained layers (the output of a Linear+BN+act layer feeds only other such layers: conv1 -> conv2 -> conv3 -> conv4 of the heads,
 * Models.py:192-197): the producer is called with Z == NULL (only Y, the pre-BN output, and bn_save are written: the streaming
 * BN+act pass is skipped), the consumer through mlsp_pointmlp_fwd_chain_f32 / mlsp_pointmlp_bwd_chain_f32 with Xpre pointing at its
 * input columns inside the producer's Y and an mlsp_defer_t that describes the producer: act(Xpre * scale + shift) and the dropout mask
 * are applied while the GEMM stages the operand (forward: A rows; wgrad: the k-major B operand) -- on the split-product kernel, the f32
 * MFMA kernel and the streaming kernels of the 3 / 16-channel output layers alike, bit-identical to the product over the materialised
 * tensor -- or by one streaming pass into the workspace for shapes outside those kernels.  The gradients the consumer returns in dX are
 * with respect to the ACTIVATED input, i.e. exactly the dZ the producer's backward expects. */
typedef struct mlsp_defer {
    const float* bn_save;  /* the producer's [4][ld]: scale | shift | mean | invstd (device) */
    int ld;                /* the producer's width: row pitch of its Y and of bn_save; its dropout stream is indexed row * ld + column */
    int col;               /* first column of THIS consumer's input inside the producer's Y (Xpre == Y + col) */
    int act;               /* the producer's activation on these columns: 0 none, 1 ReLU, 2 LeakyReLU(slope) */
    float slope;
    float p_drop;          /* its dropout rate on these columns (0: none / eval mode) */
    uint64_t seed;         /* ... and the seed of that dropout stream */
} mlsp_defer_t;
int mlsp_pointmlp_fwd_f32(const float* X, int ldx, int M, int Cin, const float* W, int ldw, int Cout, const float* bias,
                          const float* gbias, int rows_per_group, const float* gamma, const float* beta, float* run_mean,
                          float* run_var, float momentum, float eps, int training, int act, float slope, float p_drop,
                          uint64_t seed, float* Y, float* Z, float* bn_save, float* group_ysum, int precision, void* ws, size_t ws_bytes,
                          mlsp_stream_t stream);
int mlsp_pointmlp_fwd_chain_f32(const float* Xpre, int ldx, const mlsp_defer_t* in, int M, int Cin, const float* W, int ldw, int Cout, const float* bias,
                                const float* gbias, int rows_per_group, const float* gamma, const float* beta, float* run_mean,
                                float* run_var, float momentum, float eps, int training, int act, float slope, float p_drop,
                                uint64_t seed, float* Y, float* Z, float* bn_save, float* group_ysum, int precision, void* ws, size_t ws_bytes,
                                mlsp_stream_t stream);
/* BatchNorm-backward reduction fused into the consumer's dgrad (round 5).  The gradient w.r.t. a chained layer's activated output is
 * produced by its CONSUMER's dgrad; that launch (gemm_split_kernel's output pass, the small-K streaming kernel) can multiply it by the
 * producer's activation derivative and dropout mask right there, store the masked gradient d' and leave the producer's column sums of
 * d' and d' * yhat per 128-row panel: the producer's streaming reduction pass (a full read of dZ and Y) disappears.
 *   consumer: in_stats != NULL ([M / 128][2][in->ld] doubles, indexed by the producer's column, FOLLOWED (ABI v13) by [M / 128][in->ld]
 *             floats: the panels' column maxima of |d'|, a bound the producer's two-piece f16 products need) -- legal when
 *             mlsp_*_bwd_stats_parts() > 0 for the layer; dX then holds d', not dZ;
 *   producer: pre_stats != NULL ([pre_parts][2][Cout] doubles + [pre_parts][Cout] floats, as the consumers left them) -- dZ is d', its sums
 *             are given: finalisation + the BatchNorm part only;
 *             pre_stats == NULL with pre_parts < 0 (ABI v13) -- dZ is d' but its sums are INCOMPLETE: only some consumers took part in this
 *             backward pass (a loss on a subset of the heads, PointDA/trainer.py:551-565), the columns of the others are zero.  The
 *             call reduces the sums itself and does not apply the activation derivative / dropout mask a second time.
 * All consumers of a producer must be ABLE to do it or none does (they write columns of the same dX and of the same partial rows); the
 * ones that run in a given backward pass then all do it.
 * A producer in that role (training mode) does not form its output gradient dY either when its dgrad and weight-gradient launches run on
 * gemm_split_kernel: they read d' and Y and apply dY = (d' + y * nk2[c] + c0[c]) * sc[c] in their operand loads (coefficients from the
 * finalised sums), so the streaming "apply" pass and the dY tensor are gone; the per-cloud bias gradient (gbias) then comes from the panel
 * sums of d' and the clouds' column sums of Y -- group_ysum [n_groups][Cout], which the forward leaves when asked (nullable on both
 * sides: without it the backward takes one pass over Y).  Other shapes keep the apply pass (same results to rounding). */
int mlsp_pointmlp_bwd_stats_parts(int M, int Cin, int Cout, int ldw, int lddx, int precision);
int mlsp_pointmlp_bwd_chain_f32(const float* dZ, const float* Xpre, int ldx, const mlsp_defer_t* in, int M, int Cin, const float* W, int ldw, int Cout, const float* Y,
                                const float* bn_save, int has_bn, int training, int act, float slope, float p_drop, uint64_t seed,
                                int n_groups, int rows_per_group, float* dX, int lddx, int dx_accumulate, float* dW, float* dbias,
                                float* dgbias, float* dgamma, float* dbeta, double* in_stats, const double* pre_stats, int pre_parts,
                                const float* group_ysum, int precision, void* ws, size_t ws_bytes, mlsp_stream_t stream);
int mlsp_pointmlp_bwd_f32(const float* dZ, const float* X, int ldx, int M, int Cin, const float* W, int ldw, int Cout, const float* Y,
                          const float* bn_save, int has_bn, int training, int act, float slope, float p_drop, uint64_t seed,
                          int n_groups, int rows_per_group, float* dX, int lddx, int dx_accumulate, float* dW, float* dbias, float* dgbias,
                          float* dgamma, float* dbeta, const double* pre_stats, int pre_parts, const float* group_ysum, int precision, void* ws,
                          size_t ws_bytes, mlsp_stream_t stream);

/* The same layer with its ACTIVATIONS stored as bf16 in HBM (BASELINE.json configs[4]: PointSegDA N=2048 k=40 "bf16 with MFMA edge-MLP";
 * PointSegDA/Models.py:245-385 head stacks).  x_bf16: X and dX are bf16 (else fp32); out_bf16: Y, Z, dZ and the internal dY are bf16.
 * W, bias, gbias, BatchNorm parameters / statistics, dW and every reduction stay fp32; products are bf16 x bf16 with fp32 accumulation.
 * BN layers on interior GEMM tiles only: mlsp_pointmlp_mx_supported() != 0, otherwise both entry points return MLSP_ERR_UNSUPPORTED and
 * the caller runs that layer through mlsp_pointmlp_*_f32. */
int mlsp_pointmlp_mx_supported(int M, int Cin, int Cout, int ldx, int x_bf16, int training, int precision);
int mlsp_pointmlp_fwd_mx(const void* X, int x_bf16, int ldx, int M, int Cin, const float* W, int ldw, int Cout, const float* bias,
                         const float* gbias, int rows_per_group, const float* gamma, const float* beta, float* run_mean,
                         float* run_var, float momentum, float eps, int training, int act, float slope, float p_drop, uint64_t seed,
                         void* Y, void* Z, int out_bf16, float* bn_save, int precision, void* ws, size_t ws_bytes, mlsp_stream_t stream);
int mlsp_pointmlp_bwd_mx(const void* dZ, const void* X, int x_bf16, int ldx, int M, int Cin, const float* W, int ldw, int Cout,
                         const void* Y, int out_bf16, const float* bn_save, int training, int act, float slope, float p_drop,
                         uint64_t seed, int n_groups, int rows_per_group, void* dX, int lddx, int dx_accumulate, float* dW, float* dbias,
                         float* dgbias, float* dgamma, float* dbeta, int precision, void* ws, size_t ws_bytes, mlsp_stream_t stream);

/* Fused per-point conv (bias-free) + BatchNorm + act + max over the N points of each cloud:
 * conv5/bn5/LeakyReLU/adaptive_max_pool1d (PointDA/Models.py:132-136) and the T-Net's conv2d3 + torch.max(dim=2)
 * (PointDA/model_utils.py:116-117).  out [B][Cout].  Backward is closed-form through the Gram matrix X^T X (colmax.hip):
 * the dense [P][Cout] gradient is never formed.  Saved: ysel [B][Cout], arg [B][Cout], bn_save [4][Cout].  X must be
 * contiguous (ldx == Cin) for the backward. */
int mlsp_pointmlp_colmax_fwd_f32(const float* X, int ldx, int B, int N, int Cin, const float* W, int ldw, int Cout,
                                 const float* gamma, const float* beta, float* run_mean, float* run_var, float momentum, float eps,
                                 int training, int act, float slope, float* out, float* ysel, int32_t* arg, float* bn_save, int precision, void* ws,
                                 size_t ws_bytes, mlsp_stream_t stream);
int mlsp_pointmlp_colmax_bwd_f32(const float* dOut, const float* X, int ldx, int B, int N, int Cin, const float* W, int ldw,
                                 int Cout, const float* out, const float* ysel, const int32_t* arg, const float* bn_save,
                                 int training, int act, float slope, float* dX, int dx_accumulate, float* dW, float* dgamma, float* dbeta,
                                 int precision, void* ws, size_t ws_bytes, mlsp_stream_t stream);

/* Several Linear (+bias) layers side by side under ONE BatchNorm + activation (+dropout) pass: layer d of the three MLSP heads
 * (PointDA/Models.py:193-196 position, :227-230 normal, :274-279 cardinality) in one [M][sum Cout] matrix.  Segment s:
 *   Y[:, y_s : y_s + Cout_s] = X[:, x_col_s : x_col_s + Cin_s] W_s^T (+ bias_s),   y_s = Cout_0 + ... + Cout_{s-1}
 * gamma / beta / running statistics / dgamma / dbeta / dbias are [C] vectors over all the channels (C = sum Cout <= 1024), bn_save
 * [4][C].  chan [2][C]: per channel the negative-side factor of the activation max(a, f a) (0 ReLU, 0.2 LeakyReLU, 1 none) and the
 * dropout switch (0 / 1; rate p_drop, counter hash of element row * C + channel).  `segs` and `dW` (one [Cout_s][Cin_s] gradient per
 * segment) are HOST arrays.  Backward: dX [M][..] (row pitch lddx; nullable) gets segment s's input gradient in columns x_col_s..;
 * segments reading the same columns add up.  MLSP_ERR_UNSUPPORTED (mlsp_multimlp_supported() == 0) when a column slice is not
 * 16-byte aligned or C > 1024: run one mlsp_pointmlp_* per segment instead.
 * Chaining (ABI v9): `in` (HOST array of nseg descriptors, or NULL) says that X is the PRE-BatchNorm output of the previous merged layer
 * (in[s].col == segs[s].x_col): every segment's GEMM -- single or block-diagonal -- transforms its operand while staging it (forward:
 * A; backward: the X side of the weight gradient), see mlsp_defer_t above.  Z == NULL in the forward: this layer's own BatchNorm +
 * activation + dropout pass is left to ITS consumers (Y and bn_save are written). */
typedef struct mlsp_seg {
    const float* W;        /* [Cout][Cin], row pitch ldw */
    const float* bias;     /* [Cout] or NULL */
    int ldw, x_col, Cin, Cout;
} mlsp_seg_t;
int mlsp_multimlp_supported(int M, const mlsp_seg_t* segs, int nseg, int precision);
int mlsp_multimlp_fwd_f32(const float* X, int ldx, int M, const mlsp_seg_t* segs, int nseg, const mlsp_defer_t* in, const float* gamma, const float* beta,
                          float* run_mean, float* run_var, float momentum, float eps, int training, const float* chan, float p_drop,
                          uint64_t seed, float* Y, float* Z, float* bn_save, int precision, void* ws, size_t ws_bytes, mlsp_stream_t stream);
int mlsp_multimlp_bwd_stats_parts(int M, const mlsp_seg_t* segs, int nseg, int ldx, int lddx, int precision);
int mlsp_multimlp_bwd_f32(const float* dZ, const float* X, int ldx, int M, const mlsp_seg_t* segs, int nseg, const mlsp_defer_t* in, const float* Y,
                          const float* bn_save, int training, const float* chan, float p_drop, uint64_t seed, float* dX, int lddx,
                          float* const* dW, float* dbias, float* dgamma, float* dbeta, double* in_stats, const double* pre_stats, int pre_parts,
                          int precision, void* ws, size_t ws_bytes, mlsp_stream_t stream);

/* max over the k edges of every point (`.max(dim=-1)`, model_utils.py:114) on an edge-major matrix */
int mlsp_segmax_fwd_f32(const float* Z, int P, int k, int C, float* out, uint8_t* argk, mlsp_stream_t stream);
int mlsp_segmax_bwd_f32(const float* dOut, const uint8_t* argk, int P, int k, int C, float* dZ, mlsp_stream_t stream);

/* max over the N points of every cloud (model_utils.py:117 torch.max(dim=2); Models.py:136 adaptive_max_pool1d) */
int mlsp_colmax_fwd_f32(const float* Z, int B, int N, int C, float* out, int32_t* arg, mlsp_stream_t stream);
int mlsp_colmax_bwd_f32(const float* dOut, const int32_t* arg, int B, int N, int C, float* dZ, mlsp_stream_t stream);

/* masked symmetric Chamfer (MLSP/mlsp.py:115-182, scaled as calc_loss :222-229):
 * loss = scale * sum_b (A_b + B_b)/cnt_b,  scale = DefRec_weight * DefRec_SCALER / B.
 * pred [B][N][3], gold [B][3][N], mask [B][3][N]; per_cloud [B][3], argA/argB [B][N] saved. */
int mlsp_chamfer_masked_fwd_f32(const float* pred, const float* gold, const float* mask, int B, int N, float scale,
                                float* per_cloud, int32_t* argA, int32_t* argB, float* loss, mlsp_stream_t stream);
int mlsp_chamfer_masked_bwd_f32(const float* pred, const float* gold, const float* mask, int B, int N, float scale,
                                const float* per_cloud, const int32_t* argA, const int32_t* argB, const float* grad_loss,
                                float* dpred, mlsp_stream_t stream);

/* ONE direction of the masked Chamfer distance = the reference's chamfer_distance(p1, p2, mask) (MLSP/mlsp.py:115-153):
 * loss = sum_b (sum over the masked rows i of p1 of min_j (|p1_i - p2_j|^2 + 100 [mask_j == 0])) / cnt_b.
 * p1, p2 [B][N][3]; mask_cord [B][N] = mask[:, :, 0]; per_cloud [B][2] = {sum, cnt} and arg [B][N] (arg-min column of every
 * masked row) are saved for the backward, which writes dp1 and / or dp2 [B][N][3] (either may be NULL). */
int mlsp_chamfer_dir_fwd_f32(const float* p1, const float* p2, const float* mask_cord, int B, int N, float* per_cloud, int32_t* arg,
                             float* loss, mlsp_stream_t stream);
int mlsp_chamfer_dir_bwd_f32(const float* p1, const float* p2, const float* mask_cord, int B, int N, const float* per_cloud,
                             const int32_t* arg, const float* grad_loss, float* dp1, float* dp2, mlsp_stream_t stream);

/* normal loss (MLSP/mlsp.py:275-287; weighted form PointDA/trainer.py:551-556):
 * out[0] = -weight * sum_i w_i |cos(pred_i, gt_i)| / sum_i w_i ; out[1] = sum w.  w nullable (=1). */
int mlsp_normal_loss_fwd_f32(const float* pred, const float* gt, const float* w, int P, float weight, float* out, void* ws,
                             size_t ws_bytes, mlsp_stream_t stream);
int mlsp_normal_loss_bwd_f32(const float* pred, const float* gt, const float* w, int P, float weight, const float* fwd_out,
                             const float* grad_loss, float* dpred, mlsp_stream_t stream);

/* cardinality-head tail (PointDA/Models.py:281-285): p = softmax(logits), density = p . fc2w */
int mlsp_density_tail_fwd_f32(const float* logits, const float* fc2w, int P, int nc, float* pvec, float* dens,
                              mlsp_stream_t stream);
int mlsp_density_tail_bwd_f32(const float* pvec, const float* fc2w, const float* dpvec, const float* ddens, int P, int nc,
                              float* dlogits, mlsp_stream_t stream);

/* densityloss (MLSP/mlsp.py:430-454): out = {kl, mae, sum mask}; mask nullable */
int mlsp_density_loss_fwd_f32(const float* pvec, const float* dens, const float* target_vec, const float* target,
                              const float* mask, int P, int nc, float density_weight, float* out, void* ws, size_t ws_bytes,
                              mlsp_stream_t stream);
int mlsp_density_loss_bwd_f32(const float* pvec, const float* dens, const float* target_vec, const float* target,
                              const float* mask, int P, int nc, float density_weight, const float* fwd_out,
                              const float* grad_kl, const float* grad_mae, float* dpvec, float* ddens, mlsp_stream_t stream);

/* plain fp32 GEMM on the matrix cores (exposed for tests and the 3x3 input transform):
 * C[M][N] = opA(A) opB(B) + bias;  ta/tb as in gemm.hip */
int mlsp_gemm_f32(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                  const float* bias, int precision, void* ws, size_t ws_bytes, mlsp_stream_t stream);

/* Label generators of the target branch (python-pcl in the reference, PARITY UNPINNED -- see labels.hip):
 * radius neighbour count behind cal_density (MLSP/mlsp.py:240-272; count excludes cloud index 0, capped at max_nn) and
 * k-neighbourhood PCA normals behind kSearchNormalEstimation (PointDA/trainer.py:173-188; idx from mlsp_knn_f32, self
 * included; oriented towards the origin).  x [B][N] rows of >= 3 floats, row pitch ldx. */
int mlsp_radius_count_f32(const float* x, int ldx, int B, int N, float radius, int max_nn, int32_t* count, mlsp_stream_t stream);
int mlsp_knn_normals_f32(const float* x, int ldx, const int32_t* idx, int B, int N, int k, float* normals, mlsp_stream_t stream);

/* PointNet++ set-abstraction front end (BASELINE.json configs[3]).  The reference states the semantics in
 * PointDA/hengshuang_transformer/pointnet_util.py only: farthest_point_sample :53-73 (start index supplied by the caller --
 * the reference draws it with torch.randint :65), query_ball_point :76-96 (first nsample indices with !(d2 > r2) in index
 * order, padded with the first), sample_and_group :99-136 (rows [xyz_j - new_xyz_i | feat_j], edge-major (b, i, s)).
 * xyz [B][N] rows of >= 3 floats (pitch ldx); new_xyz [B][S] rows (pitch ldq); feat [B*N][D] or NULL when D == 0;
 * idx / fps_idx are int32 local to their cloud.  mlsp_group_reverse builds the reverse index of idx [B][S][ns] over the
 * N source points (same format as mlsp_knn_reverse) for the deterministic backward of the grouping. */
/* kNN of query points among DIFFERENT reference points: `square_distance` + argsort()[:, :, :k] (pointnet_util.py:26-38,116-118 knn=True
 * grouping, :237-239 Msg, :287-289 the 3-NN of feature propagation; the KNN calls of PointDA/model_utils.py:175,188 have this shape).
 * d = (-2 q.r + |q|^2) + |r|^2 in fp32 (dot = fmaf chain over the C <= 8 coordinates), ascending, ties -> lower reference index.
 * ref [B][Nr] rows (pitch ldr), qry [B][Nq] rows (pitch ldq) -> idx [B][Nq][k] (local to the cloud), dist [B][Nq][k] (nullable). k <= 64. */
int mlsp_knn_query_f32(const float* ref, int ldr, int Nr, const float* qry, int ldq, int Nq, int B, int C, int k, int32_t* idx, float* dist,
                       mlsp_stream_t stream);
/* PointNetFeaturePropagation interpolation (pointnet_util.py:287-294): out [B][N][D] = inverse-distance weighted mean of the features
 * feat [B][S][D] of the three nearest sampled points (idx, dist [B][N][3] from mlsp_knn_query_f32); backward over the reverse index of idx
 * (mlsp_group_reverse(idx, B, N, S, 3, ...)): dfeat [B][S][D], deterministic order. */
int mlsp_interp3_fwd_f32(const float* feat, const int32_t* idx, const float* dist, int B, int N, int S, int D, float* out, mlsp_stream_t stream);
int mlsp_interp3_bwd_f32(const float* dout, const float* dist, const int32_t* rev_off, const int32_t* rev_ent, int B, int N, int S, int D,
                         float* dfeat, mlsp_stream_t stream);
int mlsp_fps_f32(const float* xyz, int ldx, int B, int N, int S, const int32_t* start, int32_t* fps_idx, mlsp_stream_t stream);
int mlsp_ball_query_f32(const float* xyz, int ldx, const float* new_xyz, int ldq, int B, int N, int S, float radius_sq, int nsample,
                        int32_t* idx, mlsp_stream_t stream);
int mlsp_group_reverse(const int32_t* idx, int B, int S, int N, int ns, int32_t* rev_off, int32_t* rev_ent, mlsp_stream_t stream);
int mlsp_sa_group_fwd_f32(const float* xyz, int ldx, const float* feat, int D, const float* new_xyz, int ldq, const int32_t* idx, int B,
                          int N, int S, int ns, float* G, mlsp_stream_t stream);
/* dfeat [B][N][D] = the sum, over the groups a point sits in, of columns [col, col + D) of the grouped-row gradient dG (row pitch ldg):
 * col = 3 for the feature columns, col = 0 / D = 3 for the coordinate columns (d xyz of `grouped_xyz - new_xyz`, pointnet_util.py:122-124). */
int mlsp_sa_group_bwd_f32(const float* dG, int ldg, int col, int D, const int32_t* rev_off, const int32_t* rev_ent, int B, int N, int S, int ns,
                          float* dfeat, mlsp_stream_t stream);

/* Linear + BatchNorm + activation + max over the k consecutive rows of every group, fused (pointnet_util.py:188-195: the last conv of a
 * set-abstraction MLP and `torch.max(new_points, 2)[0]`; also sample_and_group_all with k = N <= 255).  X [M][Cin] edge rows, M = G*k.
 * fwd: Y [M][Cout] pre-BN output (kept for the backward), out [G][Cout] = max_s act(BN(y)), ysel [G][Cout] the selected pre-BN value
 * (max for scale >= 0, min otherwise), argk [G][Cout] uint8 its first slot, bn_save [4][Cout].  The activated tensor is never written.
 * bwd: dOut [G][Cout] -> dX (nullable) [M][Cin], dW [Cout][Cin], dbias (nullable), dgamma, dbeta; the BatchNorm sums come from the
 * G*Cout selected entries, the gradient of the activated tensor is never materialised.  k <= 255, Cout % 4 == 0, 256 % (Cout/4) == 0. */
int mlsp_pointmlp_segmax_fwd_f32(const float* X, int ldx, int M, int Cin, const float* W, int ldw, int Cout, const float* bias,
                                 const float* gamma, const float* beta, float* run_mean, float* run_var, float momentum, float eps,
                                 int training, int act, float slope, int k, float* Y, float* out, float* ysel, uint8_t* argk, float* bn_save,
                                 int precision, void* ws, size_t ws_bytes, mlsp_stream_t stream);
int mlsp_pointmlp_segmax_bwd_f32(const float* dOut, const float* X, int ldx, int M, int Cin, const float* W, int ldw, int Cout, const float* Y,
                                 const float* ysel, const uint8_t* argk, const float* bn_save, int training, int act, float slope, int k,
                                 float* dX, int lddx, float* dW, float* dbias, float* dgamma, float* dbeta, int precision, void* ws, size_t ws_bytes,
                                 mlsp_stream_t stream);

/* Folded first layer of a set-abstraction MLP (pointnet_util.py:120-129 grouping + :185-190 first Conv2d + BatchNorm2d + ReLU): the
 * conv over the edge rows [x_j - c_i | f_j] equals u_j - w_i with u [B*N][C] = [x | f] W^T + b per source point and w [B*S][C] = c Wx^T
 * per centre (two per-point GEMMs the caller runs with mlsp_pointmlp_fwd_f32 / mlsp_gemm_f32); neither the grouped tensor nor the
 * pre-BN output is materialised.  idx [B][S][ns] int32 (indices local to the cloud, as mlsp_ball_query_f32 / mlsp_knn_query_f32
 * return them).  fwd: BatchNorm over the E = B*S*ns edges (training: batch statistics, running buffers updated; eval: running
 * buffers), Z [E][C] = relu(scale * (u_j - w_i) + shift), bn_save [4][C] = scale | shift | mean | invstd.
 * bwd: dZ [E][C] -> du [B*N][C], dw [B*S][C] (closed-form BN backward; du over the reverse index of idx from
 * mlsp_group_reverse(idx, B, S, N, ns, ...): fixed summation order), dgamma, dbeta [C].
 * C in {16, 32, 64, 128, 256}, ns <= 256; MLSP_ERR_UNSUPPORTED otherwise (the caller keeps the grouped path). */
int mlsp_sa_fold_fwd_f32(const float* u, const float* w, const int32_t* idx, int B, int N, int S, int ns, int C, const float* gamma,
                         const float* beta, float* run_mean, float* run_var, float momentum, float eps, int training, float* Z,
                         float* bn_save, void* ws, size_t ws_bytes, mlsp_stream_t stream);
int mlsp_sa_fold_bwd_f32(const float* dZ, const float* u, const float* w, const int32_t* idx, const int32_t* rev_off, const int32_t* rev_ent,
                         const int32_t* rev_cnt, const int32_t* pad_cnt, int B, int N, int S, int ns, int C, const float* bn_save, int training,
                         float* du, float* dw, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, mlsp_stream_t stream);
/* Reverse index of padded groups WITHOUT their padding slots (ball-query groups repeat their first hit: slot s > 0 with idx[i][s] ==
 * idx[i][0]): rev_off [B*N+1] list starts, rev_cnt [B*N] list lengths (the lists of a cloud no longer fill its S*ns entries), rev_ent as
 * mlsp_group_reverse, pad_cnt [B*S] padding slots per group.  mlsp_sa_fold_bwd_f32 takes rev_cnt / pad_cnt (both or neither, NULL = the
 * full index of mlsp_group_reverse) and adds a group's identical padding rows as a multiple of one row. */
int mlsp_group_reverse_compact(const int32_t* idx, int B, int S, int N, int ns, int32_t* rev_off, int32_t* rev_cnt, int32_t* rev_ent,
                               int32_t* pad_cnt, mlsp_stream_t stream);

/* Input corruption.  mlsp_region_assign_f32: utils/pc_utils.py:33-73 assign_region_to_point on X [B][C][N]
 * (channel-major as the trainer holds it); thr[n+1] = fp32 voxel edges, clip = fp32(0.99999999); regions int32 [B][N].
 * mlsp_deform_regions_f32: MLSP/mlsp.py:10-51 deform_input, DefRec_dist == 'volume_based_voxels': per cloud the first
 * `groups` regions of `order` (np.random.permutation(n^3) in the reference, :27) holding >= min_pts points are replaced by
 * lookup[region] + noise (noise [B][3][N], already scaled by sqrt(0.001): pc_utils.draw_from_gaussian :114-122);
 * X is updated in place, mask [B][C][N] receives 1 on the first three channels of the replaced points, 0 elsewhere. */
int mlsp_region_assign_f32(const float* X, int B, int C, int N, const float* thr, int n, float clip, int32_t* regions, mlsp_stream_t stream);
int mlsp_deform_regions_f32(float* X, int B, int C, int N, const int32_t* regions, const int32_t* order, int nreg, const float* lookup,
                            const float* noise, int min_pts, int groups, float* mask, mlsp_stream_t stream);
/* Input transform (PointDA/Models.py:113 `x = torch.matmul(transformd_x0, x)`, PointSegDA/Models.py:218): out [B*N][3], out[p] = T[b] x[p] for the
 * points of cloud b; x [B*N][3] point-major, T [B][3][3].  Backward: dx (nullable) [B*N][3] and dT [B][3][3] (fixed reduction order). */
int mlsp_transform3_fwd_f32(const float* x, const float* T, int B, int N, float* out, mlsp_stream_t stream);
int mlsp_transform3_bwd_f32(const float* x, const float* T, const float* dout, int B, int N, float* dx, float* dT, mlsp_stream_t stream);

/* Two 1x1 convolutions with nothing in between (PointSegDA/Models.py:176-178, :180-182 `x = self.conv2(self.conv1(x))`) are ONE linear map:
 * W [Co][Ci] = Wb Wa, b [Co] = Wb ba + bb with Wa [Cm][Ci], ba [Cm], Wb [Co][Cm], bb [Co] (row-major, fmaf chains over m ascending).
 * Backward: dWa [Cm][Ci] = Wb^T dW, dba [Cm] = Wb^T db, dWb [Co][Cm] = dW Wa^T + db ba^T (the gradient of bb is db itself).
 * Ci, Cm, Co <= 1024. */
int mlsp_compose_linear_fwd_f32(const float* Wa, const float* ba, const float* Wb, const float* bb, int Cm, int Ci, int Co, float* W, float* b,
                                mlsp_stream_t stream);
int mlsp_compose_linear_bwd_f32(const float* dW, const float* db, const float* Wa, const float* ba, const float* Wb, int Cm, int Ci, int Co,
                                float* dWa, float* dba, float* dWb, mlsp_stream_t stream);

/* deform_input(..., 'volume_based_radius') = pc_utils.collapse_to_point (MLSP/mlsp.py:33-36, utils/pc_utils.py:76-111): per cloud one
 * point with >= min_pts points within sqrt(radius2) is picked (uniformly by u[b] in [0,1) among the candidates in index order, or
 * choice[b] >= 0 pins it) and all points within that radius of it become centre + noise (noise [B][3][N], already scaled by the
 * Gaussian's std); mask [B][3][N] marks them; chosen[b] = the picked index or -1 when the cloud has no candidate (left untouched).
 * X [B][3][N] is updated in place. */
int mlsp_collapse_to_point_f32(float* X, int B, int N, const int32_t* choice, const float* u, const float* noise, float radius2,
                               int min_pts, float* mask, int32_t* chosen, mlsp_stream_t stream);

/* MLSP/mlsp.py:54-89 scan_input / p_scan: X [B][N][C] point-major; R [B][9] float64 row-major rotation of every cloud
 * (rotate_point_cloud_3d :91-112, drawn on the host); pixel = int(2 / pixel_size).  Xs [B][N][C] keeps only the visible points,
 * mask [B][N][C] is 0 on their first three channels and 1 elsewhere. */
int mlsp_scan_select_f32(const float* X, int B, int N, int C, const double* R, int pixel, float* Xs, float* mask, mlsp_stream_t stream);

/* (ABI v7's process-wide mlsp_set_gemm_precision is gone: see `precision` in the conventions above.) */

/* Measurement aid (bench.py `roofline`): while armed, every gemm_f32_kernel launch is bracketed by two HIP
 * events on its launch stream.  mlsp_profile_end synchronises those events and fills
 * out[4] = {total ms in the kernel, launches, sum of algorithmic 2*M*N*K, sum of the A + B + C bytes}.  Not for production steps. */
int mlsp_profile_begin(void);
int mlsp_profile_end(double* out);
/* Per kernel class of the same bracket (call after mlsp_profile_end): out [MLSP_PROF_CLASSES][3] = {ms, launches, algorithmic work}.
 * Classes: 0 GEMM family (FLOP) | 1 kNN C <= 4 incl. its row norms (compulsory bytes: (C + k) * 4 per point) | 2 kNN C = 64 |
 * 3 kNN C = 128 (FLOP: 2 N C per point) | 4 EdgeConv neighbour gather-reduce (compulsory bytes) | 5 T-Net per-edge stage forward |
 * 6 its backward (FLOP of the 64 -> 128 per-edge contraction: 2 resp. 4 * E * 64 * 128) | 7 the launches of class 0 that ran on the
 * bf16-split kernel (mode 2; algorithmic FLOP, each executed as six bf16 MFMA products). */
#define MLSP_PROF_CLASSES 8
int mlsp_profile_classes(double* out, int ncls);
/* out [4][4]: {ms, launches, algorithmic FLOP, algorithmic bytes} of the bracket's gemm_split_kernel launches by kind (forward, dgrad, wgrad);
 * row 3 (ABI v13): those of all kinds that ran on the two-piece f16 products (MLSP_PREC_F16X3). */
int mlsp_profile_split_kinds(double* out);

/* Adam step (PointDA/trainer.py:258-259, stepped at :571) over flat parameter / exp_avg / exp_avg_sq buffers P / M / V in ONE launch:
 * segment s covers elements [off[s], off[s] + numel[s]) of the three buffers (buffers 16-byte aligned; off % 4 == 0 takes the 16-byte path) and reads its
 * gradient where autograd left it (grads[s], contiguous fp32, device).  off / numel / grads are HOST arrays.  step >= 1 numbers this
 * update (bias corrections 1 - beta^step); step_out (nullable, device float) receives it.  The element-wise arithmetic restates
 * torch's fused Adam (ATen/native/cuda/fused_adam_utils.cuh, ADAM_MODE::ORIGINAL: weight decay added to the gradient) type by type.
 * tile_amax (nullable; ABI v13): one float per 2048-element tile -- tiles numbered segment by segment in the order given, ceil(numel[s] /
 * 2048) per segment -- receives the largest magnitude of the tile's UPDATED parameters: a free bound for the GEMMs that read the
 * parameters next (mlsp_bound_t: partials = tile_amax + first tile of the weight, n = its tiles). */
int mlsp_adam_flat_f32(float* P, float* M, float* V, const uint32_t* off, const uint32_t* numel, const float* const* grads, int nseg, double lr,
                       double beta1, double beta2, double weight_decay, double eps, int64_t step, float* step_out, float* tile_amax,
                       mlsp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MLSP_HIP_H */
