// Shared device/host helpers for libmlsp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define MLSP_OK 0
#define MLSP_ERR_ARG (-1)        // bad shape / null pointer / unsupported size
#define MLSP_ERR_WORKSPACE (-2)  // workspace too small
#define MLSP_ERR_UNSUPPORTED (-3)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

static inline int mlsp_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLSP_OK : (int)e;
}

// HIP-event profiling hook (gemm.hip; armed by mlsp_profile_begin, off otherwise): kernel classes of mlsp_profile_classes()
enum { MLSP_PROF_GEMM = 0, MLSP_PROF_KNN_C3, MLSP_PROF_KNN_C64, MLSP_PROF_KNN_C128, MLSP_PROF_EDGE_REDUCE, MLSP_PROF_TNET_FWD,
       MLSP_PROF_TNET_BWD, MLSP_PROF_GEMM_SPLIT /* the subset of class 0 that ran on gemm_split_kernel */, MLSP_PROF_NCLS };
int prof_cls_begin(hipStream_t st, int cls);              // -> token (< 0: not armed)
void prof_cls_end(hipStream_t st, int token, double work);

// Products of the GEMM family are a PER-CALL argument of every entry point that reaches it (include/mlsp_hip.h `precision`): 0 f32 MFMA,
// 1 bf16-rounded operands, 2 fp32-accurate six-product bf16 split.  An entry point opens a GemmPrecisionScope from its argument; the
// launch helpers below it read gemm_precision_mode().  The value lives in a thread-local for the duration of that one call (restored on
// exit, so nested / concurrent calls on any threads never see each other's mode): nothing outlives a call, the library keeps no switch.
#define MLSP_AMAX_TAIL_BYTES 65536
struct GemmPrecisionScope {
    int prev;
    void* prev_tail; void* prev_offered; int prev_noffered;
    // ws / ws_bytes: the call's workspace -- its last MLSP_AMAX_TAIL_BYTES hold the operand-magnitude partials of the two-piece f16 products
    // (mode 3; gemm.hip amax_partials); Workspace below never hands that tail out
    explicit GemmPrecisionScope(int mode, void* ws = nullptr, size_t ws_bytes = 0);
    ~GemmPrecisionScope();
    GemmPrecisionScope(const GemmPrecisionScope&) = delete;
    GemmPrecisionScope& operator=(const GemmPrecisionScope&) = delete;
};
int gemm_precision_mode();
#define PREC_SCOPE(mode_) if ((mode_) < 0 || (mode_) > 3) return MLSP_ERR_ARG; GemmPrecisionScope prec_scope_(mode_, ws, ws_bytes)

// Raise a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) -- at most once per (device, kernel, size): the host call
// costs tens of microseconds, a dozen of them per step put the enqueue thread behind the GPU on slower hosts.  (api.hip; a cache of
// what was already asked of the runtime, not dispatch state: the same launches happen with or without it.)
hipError_t mlsp_lds_limit(const void* fn, size_t lds);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// simple bump allocator over the caller-provided workspace
struct Workspace {
    char* base;
    size_t size, off;
    Workspace(void* p, size_t n) : base((char*)p), size(n > MLSP_AMAX_TAIL_BYTES ? n - MLSP_AMAX_TAIL_BYTES : 0), off(0) {}     // (the tail: see PREC_SCOPE)
    template <typename T>
    T* take(size_t count) {
        size_t bytes = align_up(count * sizeof(T), 256);
        if (!base || off + bytes > size) { off = size + 1; return nullptr; }
        T* r = (T*)(base + off);
        off += bytes;
        return r;
    }
    bool ok() const { return off <= size; }
};

// Linear block id -> (cloud, chunk) such that every chunk of a cloud has the same blockIdx % 8, i.e. lands on one XCD (and one
// 4 MiB L2) under the observed round-robin dispatch: per-cloud kernels (kNN candidates, neighbour-row gathers) then re-read
// their cloud from L2 instead of the fabric.  Placement only affects speed; B % 8 != 0 falls back to cloud-major order.
__device__ __forceinline__ void xcd_cloud_map(int bid, int bpc, int B, int& cloud, int& chunk) {
    if ((B & 7) == 0) {
        const int g = bid / (8 * bpc), r = bid - g * 8 * bpc;
        cloud = 8 * g + (r & 7); chunk = r >> 3;
    } else {
        cloud = bid / bpc; chunk = bid - cloud * bpc;
    }
}

// Operand transform of a GEMM (gemm.hip): the operand holds the PRE-BatchNorm output of the previous layer; act(x * scale[c] + shift[c])
// and that layer's dropout are applied while the tile is staged.  which: 1 = A ([M][K] row-major, c = k), 2 = B ([K][N] k-major, c = n).
// scale / shift point at the operand's FIRST channel; ld / col: row pitch of the matrix the previous layer wrote and the operand's first
// column in it (the dropout stream is indexed by the element's place in that matrix: a column slice of a merged layer keeps its mask).
// mean / invstd (nullable): the previous layer's BATCH statistics rows (training mode), at the operand's first channel -- with them the
// transformed values are bounded analytically (|yhat| <= sqrt(rows)), which the two-piece f16 products need (gemm.hip GemmArgs a_amax).
struct GemmXf {
    const float* scale; const float* shift; int act; float slope; uint32_t thresh; float inv_keep; uint64_t seed; int ld; int which; int col;
    const float* mean = nullptr; const float* invstd = nullptr;
};

// The previous layer's BatchNorm-backward reduction fused into a dgrad's output pass (gemm.hip GemmArgs bs_*; thin.hip): y / bn point at the
// column of the dgrad's C column 0 (y: that layer's pre-BN output, row pitch ldy; bn: its scale | shift | mean | invstd rows, pitch bnld);
// act / slope / thresh / inv_keep / seed: its activation and dropout; ld / col: row pitch of its matrix and C's column 0 in it; part:
// [row panels of 128][2][stat_ld] fp64 sums of d' and d' * yhat, at C's column 0.
struct GemmBs {
    const float* y; int ldy; const float* bn; int bnld; int act; float slope; uint32_t thresh; float inv_keep; uint64_t seed; int ld; int col;
    double* part; int stat_ld;
    float* amax = nullptr;          // nullable: [row panels][stat_ld] floats at C's column 0: column maxima of |d'| per panel (a by-product for the producer)
};

// A layer's BatchNorm backward applied to a GEMM's A operand while it is staged (gemm.hip GemmArgs dy_*, gemm_split_kernel<.., DY>): A holds the
// masked gradient d' that the consumer's dgrad left (GemmBs), y the layer's pre-BN output at the same coordinates and pitch, coef the rows
// c0 | nk2 | sc (pitch cld; bn.hip bn_bwd_finalize_kernel) at A's channel 0:  dY = (d' + y * nk2 + c0) * sc.
struct GemmDy { const float* y; const float* coef; int cld; int amax = 0; };       // amax: coef has a FOURTH row, max |d'| per channel

// Block-diagonal product in one GEMM launch (gemm.hip GemmArgs groups).  mode 1: output COLUMNS are grouped (forward: A = X + g * a_gs,
// B = Bg[g] = W_g; dgrad alike); mode 2: output ROWS are grouped (wgrad: B = X + g * b_gs; A and C take the launch-wide row index).
struct GemmGroups { int G, mode; long a_gs, b_gs; const float* Bg[4]; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float lrelu_or_relu(float x, int act, float slope) {
    // act: 0 none, 1 relu, 2 leaky relu
    if (act == 0) return x;
    return x > 0.f ? x : (act == 1 ? 0.f : x * slope);
}

// Counter-based dropout: ONE 32-bit hash (lowbias32 mix) per aligned quad of elements, one byte of it per element: element i is kept
// iff byte (i & 3) of hash(seed, i >> 2) >= thresh, thresh = round(256 p) in [0, 255] (p = 0.5 is exact; other rates are quantised to
// 1/256 and inv_keep = 256 / (256 - thresh) keeps the expectation exact).  The two 32-bit multiplies of the mix are quarter-rate
// instructions: per quad instead of per element they are cheap enough to regenerate the mask inside a GEMM operand load (gemm.hip XF).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t dropout_hash4(uint64_t seed, uint64_t quad) {
    return mix32((uint32_t)quad ^ mix32((uint32_t)(quad >> 32) + (uint32_t)seed) ^ (uint32_t)(seed >> 32) * 0x9e3779b9U);
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t i, uint32_t thresh) {
    return ((dropout_hash4(seed, i >> 2) >> (8 * ((uint32_t)i & 3))) & 255u) >= thresh;
}
// The operand transform on one aligned quad (the arithmetic of bn_act_fwd_vec_kernel / multi_act_fwd_kernel, element by element):
// act(v * sc + sh) with the activation as ONE max (slope = effective negative-side factor in [0, 1]), then dropout by byte e of the
// quad's hash.  XfDev: what a kernel needs of a GemmXf; xH = the launch-uniform half of dropout_hash4 (quad index < 2^32).
struct XfDev { const float* scale; const float* shift; float slope, inv_keep; uint32_t thresh, xH; int ld, col; };
__device__ __forceinline__ f32x4 xf_apply_quad(f32x4 v, const f32x4& sc, const f32x4& sh, float slope, uint32_t thresh, float inv_keep, uint32_t hq) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a = fmaf(v[e], sc[e], sh[e]);
        a = fmaxf(a, a * slope);
        if (thresh) a = ((hq >> (8 * e)) & 255u) >= thresh ? a * inv_keep : 0.f;
        v[e] = a;
    }
    return v;
}
// host side: the byte threshold of a rate and the matching rescale
static inline uint32_t dropout_thresh8(float p) {
    if (!(p > 0.f)) return 0u;
    const int t = (int)(p * 256.0f + 0.5f);
    return (uint32_t)(t > 255 ? 255 : t);
}
static inline float dropout_inv_keep8(float p) {
    const uint32_t t = dropout_thresh8(p);
    return t ? 256.0f / (float)(256u - t) : 1.0f;
}

// host: device-side view of a GemmXf (effective slope: 1 no activation, 0 ReLU, the LeakyReLU slope otherwise)
static inline uint32_t mix32_host(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
static inline XfDev xf_dev(const GemmXf& x) {
    XfDev d;
    d.scale = x.scale; d.shift = x.shift; d.slope = x.act == 0 ? 1.f : x.act == 1 ? 0.f : x.slope; d.inv_keep = x.inv_keep; d.thresh = x.thresh;
    d.xH = mix32_host((uint32_t)x.seed) ^ (uint32_t)(x.seed >> 32) * 0x9e3779b9U; d.ld = x.ld; d.col = x.col;
    return d;
}
