// Adam step of the hot path (PointDA/trainer.py:258-259: optim.Adam(model.parameters(), lr, weight_decay), stepped at :571) over FLAT
// parameter / moment buffers in one launch.
//
// torch's fused Adam hands every workgroup one 64 Ki-element chunk of one tensor: 4.55 M parameters are 70-odd mostly full chunks plus 77
// tails, i.e. a quarter of the chip moving 127 MB -- 110 us in five launches (profiles/r5_*).  Here the parameters, exp_avg and exp_avg_sq
// live in three flat fp32 buffers (mlsp_amd/optim.py), the gradients are read WHERE AUTOGRAD LEFT THEM through a pointer table in the
// kernel arguments (no packing copy), and a workgroup owns a 2048-element tile: ~2300 workgroups, one pass at the HBM roof.
//
// Arithmetic: the element-wise update of torch's fused kernel (ATen/native/cuda/fused_adam_utils.cuh `adam_math`, ADAM_MODE::ORIGINAL, no
// amsgrad / maximize / grad scaling), restated with its types: lr, beta1, beta2, weight_decay, eps are doubles there, so the products with
// them are formed in double and rounded to float on assignment; the two bias corrections are rounded to float before use.
//   grad   += param * weight_decay                          (double, float result)
//   exp_avg = beta1 * exp_avg + (1 - beta1) * grad          (double)
//   exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * grad * grad   (double, left to right)
//   step_size = lr / bias_correction1 ;  denom = sqrt(exp_avg_sq) / bias_correction2_sqrt + eps ;  param -= step_size * exp_avg / denom
// The one freedom the source leaves is the lowering: torch's build contracts each double a * b + c above into an fma and keeps the float
// division and square root correctly rounded -- established by tools/r5/adam_probe (32 lowerings against torch._fused_adam_ on 4 Mi
// elements: this one differs in 0 elements of param / exp_avg / exp_avg_sq, the uncontracted form in 16,285 exp_avg values;
// profiles/r5_adam_lowering_probe.txt).  This file is compiled with -ffp-contract=off, so the fmas are written out.
// tests/test_gpu_optim.py asserts bit-identity with torch.optim.Adam(fused=True) step by step.
#include "common.h"
#include "../../include/mlsp_hip.h"
#include <math.h>

#define ADAM_MAX_SEGS 96
#define ADAM_TILE 2048            // elements per workgroup (256 threads x 2 quads)

struct AdamSegs {
    int n;
    int tile_begin[ADAM_MAX_SEGS + 1];      // first tile of every segment; [n] = total
    unsigned off[ADAM_MAX_SEGS];            // first element of the segment in the flat buffers (a multiple of 4 takes the 16-byte path)
    unsigned numel[ADAM_MAX_SEGS];
    const float* grad[ADAM_MAX_SEGS];       // the segment's gradient, contiguous
};

__device__ __forceinline__ void adam_one(float& param, float grad, float& ea, float& es, double lr, double b1, double b2, double wd, double eps,
                                         float bc1, float bc2s) {
    if (wd != 0.0) grad = (float)fma((double)param, wd, (double)grad);
    ea = (float)fma(b1, (double)ea, (1.0 - b1) * (double)grad);
    es = (float)fma(b2, (double)es, ((1.0 - b2) * (double)grad) * (double)grad);
    const float step_size = (float)(lr / (double)bc1);
    const float denom = (float)((double)(sqrtf(es) / bc2s) + eps);
    param -= step_size * ea / denom;
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ P, float* __restrict__ M, float* __restrict__ V, AdamSegs s, double lr,
                                                        double b1, double b2, double wd, double eps, float bc1, float bc2s, float step,
                                                        float* __restrict__ step_out, float* __restrict__ tile_amax) {
    // tile_amax (nullable, [gridDim.x]): max |updated parameter| of this workgroup's tile -- a by-product for the GEMMs that read the
    // parameters next (the two-piece f16 products scale every operand by a bound of its magnitude: include/mlsp_hip.h mlsp_bound_t)
    float pmx = 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && step_out) *step_out = step;     // the optimizer's device-side step counter (state_dict)
    // which segment owns this tile: binary search over <= 96 tile offsets in the kernel arguments
    int lo = 0, hi = s.n - 1;
    const int t = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s.tile_begin[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const unsigned n = s.numel[lo];
    const unsigned e0 = (unsigned)(t - s.tile_begin[lo]) * ADAM_TILE;
    const float* __restrict__ g = s.grad[lo];
    float* p = P + s.off[lo];
    float* m = M + s.off[lo];
    float* v = V + s.off[lo];
    const bool gvec = (((uintptr_t)g) & 15) == 0 && (s.off[lo] & 3) == 0;      // 16-byte accesses on all four streams
#pragma unroll
    for (int u = 0; u < ADAM_TILE / 1024; ++u) {
        const unsigned i = e0 + u * 1024 + threadIdx.x * 4;
        if (i >= n) break;
        if (i + 4 <= n && gvec) {
            f32x4 pp = *(const f32x4*)(p + i), mm = *(const f32x4*)(m + i), vv = *(const f32x4*)(v + i);
            const f32x4 gg = *(const f32x4*)(g + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float p1 = pp[e], m1 = mm[e], v1 = vv[e];
                adam_one(p1, gg[e], m1, v1, lr, b1, b2, wd, eps, bc1, bc2s);
                pp[e] = p1; mm[e] = m1; vv[e] = v1;
                pmx = fmaxf(pmx, fabsf(p1));
            }
            *(f32x4*)(p + i) = pp; *(f32x4*)(m + i) = mm; *(f32x4*)(v + i) = vv;
        } else {
            for (unsigned j = i; j < n && j < i + 4; ++j) {
                float pp = p[j], mm = m[j], vv = v[j];
                adam_one(pp, g[j], mm, vv, lr, b1, b2, wd, eps, bc1, bc2s);
                p[j] = pp; m[j] = mm; v[j] = vv;
                pmx = fmaxf(pmx, fabsf(pp));
            }
        }
    }
    if (tile_amax) {                    // (uniform: a kernel argument)
        __shared__ float smx[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pmx = fmaxf(pmx, __shfl_xor(pmx, o, 64));
        if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = pmx;
        __syncthreads();
        if (threadIdx.x == 0) tile_amax[t] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    }
}

extern "C" {

// One Adam step over nseg parameter segments of the flat buffers P / M / V (exp_avg / exp_avg_sq): segment s covers elements
// [off[s], off[s] + numel[s]) (the buffers 16-byte aligned; a segment whose offset and gradient pointer are 16-byte aligned moves 16 bytes per
// lane, any other one element by element) and reads its gradient from grads[s] (any fp32 device pointer, contiguous).  step >= 1 is this update's number (bias corrections 1 - beta^step); step_out (nullable, device float) receives it.
// Host arrays; any nseg (launched in groups of 96).
// tile_amax (nullable; ABI v13): one float per 2048-element tile, tiles numbered segment by segment in the order given
// (ceil(numel[s] / 2048) tiles per segment): the largest magnitude of the UPDATED parameters of the tile.
int mlsp_adam_flat_f32(float* P, float* M, float* V, const uint32_t* off, const uint32_t* numel, const float* const* grads, int nseg, double lr,
                       double beta1, double beta2, double weight_decay, double eps, int64_t step, float* step_out, float* tile_amax,
                       mlsp_stream_t st) {
    if (!P || !M || !V || !off || !numel || !grads || nseg <= 0 || step < 1) return MLSP_ERR_ARG;
    if ((((uintptr_t)P | (uintptr_t)M | (uintptr_t)V) & 15) != 0) return MLSP_ERR_ARG;
    // (as the reference kernel: pow in double, the corrections handed on as floats)
    const float bc1 = (float)(1.0 - pow(beta1, (double)(float)step));
    const float bc2s = (float)sqrt(1.0 - pow(beta2, (double)(float)step));
    size_t tile_base = 0;
    for (int s0 = 0; s0 < nseg; s0 += ADAM_MAX_SEGS) {
        AdamSegs a;
        a.n = nseg - s0 < ADAM_MAX_SEGS ? nseg - s0 : ADAM_MAX_SEGS;
        int tiles = 0;
        for (int i = 0; i < a.n; ++i) {
            if (!grads[s0 + i] || numel[s0 + i] == 0) return MLSP_ERR_ARG;
            a.tile_begin[i] = tiles;
            a.off[i] = off[s0 + i]; a.numel[i] = numel[s0 + i]; a.grad[i] = grads[s0 + i];
            tiles += (int)((numel[s0 + i] + ADAM_TILE - 1) / ADAM_TILE);
        }
        a.tile_begin[a.n] = tiles;
        hipLaunchKernelGGL(adam_flat_kernel, dim3(tiles), dim3(256), 0, st, P, M, V, a, lr, beta1, beta2, weight_decay, eps, bc1, bc2s, (float)step,
                           s0 == 0 ? step_out : (float*)nullptr, tile_amax ? tile_amax + tile_base : (float*)nullptr);
        tile_base += tiles;
    }
    return mlsp_launch_status();
}

}  // extern "C"
