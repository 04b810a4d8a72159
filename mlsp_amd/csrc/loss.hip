// The three masked-local-structure losses and the cardinality head's tail.
//   masked Chamfer          MLSP/mlsp.py:115-153 (chamfer_distance), :156-182 (reconstruction_loss)
//   |cos| normal loss       MLSP/mlsp.py:275-283 and the weighted form at PointDA/trainer.py:551-556
//   soft-label CE + L1      MLSP/mlsp.py:430-454 (densityloss)
//   softmax -> E[count]     PointDA/Models.py:281-285 (Density_prediction tail, frozen fc2)
// The reference materialises two [B,N,N,3] tensors per Chamfer direction; here one workgroup per
// cloud keeps the cloud in LDS and only the MASKED rows (40-80 per cloud) are scanned, one wave per
// row, lanes across the N columns.
#include "common.h"
#include <math.h>

// ---------------------------------------------------------------------------------------------
// pred [B][N][3] (head output), gold [B][3][N], mask [B][3][N] (row 0 is used, as mlsp.py:141).
// per_cloud[b] = {sumA, sumB, cnt}: A = rows from gold vs columns of pred, B = the converse.
// argA/argB [B][N]: arg-min column for every masked row (undefined elsewhere).
__global__ __launch_bounds__(1024) void chamfer_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gold,
                                                           const float* __restrict__ mask, int N,
                                                           float* __restrict__ per_cloud, int* __restrict__ argA,
                                                           int* __restrict__ argB) {
    extern __shared__ float csm[];
    float* px = csm;              // pred  [3][N]
    float* gx = csm + 3 * N;      // gold  [3][N]
    float* pen = csm + 6 * N;     // penalty per column: 0 if masked else 100
    int* rows = (int*)(csm + 7 * N);          // compacted masked rows [N]
    float* rowA = csm + 8 * N;                // per-row min, direction A  [N]
    float* rowB = csm + 9 * N;                // direction B [N]
    __shared__ int nrows;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const float* pb = pred + (size_t)b * N * 3;
    const float* gb = gold + (size_t)b * 3 * N;
    const float* mb = mask + (size_t)b * 3 * N;
    for (int n = tid; n < N; n += nt) {
        px[n] = pb[n * 3 + 0]; px[N + n] = pb[n * 3 + 1]; px[2 * N + n] = pb[n * 3 + 2];
        gx[n] = gb[n]; gx[N + n] = gb[N + n]; gx[2 * N + n] = gb[2 * N + n];
        pen[n] = mb[n] == 0.f ? 100.f : 0.f;
    }
    if (tid == 0) nrows = 0;
    __syncthreads();
    // deterministic compaction of masked rows (ascending) by one wave
    if (wave == 0) {
        int base = 0;
        for (int n0 = 0; n0 < N; n0 += 64) {
            int n = n0 + lane;
            bool m = n < N && mb[n] != 0.f;
            unsigned long long bal = __ballot(m);
            int pos = __popcll(bal & ((1ull << lane) - 1ull));
            if (m) rows[base + pos] = n;
            base += __popcll(bal);
        }
        if (lane == 0) nrows = base;
    }
    __syncthreads();
    const int nr = nrows;
    // one wave per (row, direction)
    for (int job = wave; job < 2 * nr; job += nw) {
        const int dir = job & 1, i = rows[job >> 1];
        const float* p1 = dir == 0 ? gx : px;     // rows
        const float* p2 = dir == 0 ? px : gx;     // columns
        const float ax = p1[i], ay = p1[N + i], az = p1[2 * N + i];
        float best = INFINITY;
        int bj = 0x7fffffff;
        for (int j = lane; j < N; j += 64) {
            float dx = ax - p2[j], dy = ay - p2[N + j], dz = az - p2[2 * N + j];
            float nrm = sqrtf(dx * dx + dy * dy + dz * dz);     // reference: norm(...)**2  (mlsp.py:138)
            float d = nrm * nrm + pen[j];
            if (d < best) { best = d; bj = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(best, o, 64);
            int oj = __shfl_xor(bj, o, 64);
            if (ov < best || (ov == best && oj < bj)) { best = ov; bj = oj; }
        }
        if (lane == 0) {
            if (dir == 0) { rowA[job >> 1] = best; argA[(size_t)b * N + i] = bj; }
            else { rowB[job >> 1] = best; argB[(size_t)b * N + i] = bj; }
        }
    }
    __syncthreads();
    if (tid == 0) {   // fixed-order sums -> reproducible
        float sa = 0.f, sb = 0.f;
        for (int r = 0; r < nr; ++r) { sa += rowA[r]; sb += rowB[r]; }
        per_cloud[b * 3 + 0] = sa; per_cloud[b * 3 + 1] = sb; per_cloud[b * 3 + 2] = (float)nr;
    }
}

// loss = scale * sum_b (A_b + B_b) / cnt_b       (scale = weight * DefRec_SCALER / B)
__global__ void chamfer_finalize_kernel(const float* __restrict__ per_cloud, int B, float scale, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += per_cloud[b * 3 + 0] / per_cloud[b * 3 + 2] + per_cloud[b * 3 + 1] / per_cloud[b * 3 + 2];
        loss[0] = scale * s;
    }
}

// dpred [B][N][3]; gscale_ptr[0] = upstream grad, multiplied by `scale`
__global__ __launch_bounds__(1024) void chamfer_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gold,
                                                           const float* __restrict__ mask, int N,
                                                           const float* __restrict__ per_cloud, const int* __restrict__ argA,
                                                           const int* __restrict__ argB, const float* __restrict__ gout,
                                                           float scale, float* __restrict__ dpred) {
    extern __shared__ int bsm[];
    int* rows = bsm;            // [N]
    int* tgt = bsm + N;         // argA of each masked row [N]
    __shared__ int nrows;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const float* pb = pred + (size_t)b * N * 3;
    const float* gb = gold + (size_t)b * 3 * N;
    const float* mb = mask + (size_t)b * 3 * N;
    if (wave == 0) {
        int base = 0;
        for (int n0 = 0; n0 < N; n0 += 64) {
            int n = n0 + lane;
            bool m = n < N && mb[n] != 0.f;
            unsigned long long bal = __ballot(m);
            int pos = __popcll(bal & ((1ull << lane) - 1ull));
            if (m) { rows[base + pos] = n; tgt[base + pos] = argA[(size_t)b * N + n]; }
            base += __popcll(bal);
        }
        if (lane == 0) nrows = base;
    }
    __syncthreads();
    const int nr = nrows;
    const float coef = gout[0] * scale * 2.0f / per_cloud[b * 3 + 2];
    for (int j = tid; j < N; j += nt) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        const float pxj = pb[j * 3], pyj = pb[j * 3 + 1], pzj = pb[j * 3 + 2];
        // direction A: masked gold rows whose nearest pred column is j   (fixed order)
        for (int r = 0; r < nr; ++r)
            if (tgt[r] == j) {
                int i = rows[r];
                gx += pxj - gb[i]; gy += pyj - gb[N + i]; gz += pzj - gb[2 * N + i];
            }
        // direction B: j itself is a masked pred row
        if (mb[j] != 0.f) {
            int t = argB[(size_t)b * N + j];
            gx += pxj - gb[t]; gy += pyj - gb[N + t]; gz += pzj - gb[2 * N + t];
        }
        float* o = dpred + ((size_t)b * N + j) * 3;
        o[0] = coef * gx; o[1] = coef * gy; o[2] = coef * gz;
    }
}

// ---------------------------------------------------------------------------------------------
// ONE direction of the masked Chamfer distance, as the reference's chamfer_distance(p1, p2, mask) (MLSP/mlsp.py:115-153):
// p1, p2 [B][N][3] point-major, mc [B][N] = mask[:, :, 0].  Rows = the masked points of p1, columns = all points of p2 with a
// +100 penalty on the unmasked ones.  per_cloud[b] = {sum of the row minima, number of masked rows}; arg [B][N] the arg-min column
// of every masked row.
__global__ __launch_bounds__(1024) void chamfer_dir_fwd_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                               const float* __restrict__ mc, int N, float* __restrict__ per_cloud,
                                                               int* __restrict__ arg) {
    extern __shared__ float dsm[];
    float* cx = dsm;                  // p2 (columns) [3][N]
    float* pen = dsm + 3 * N;         // [N]
    int* rows = (int*)(dsm + 4 * N);  // [N]
    float* rowv = dsm + 5 * N;        // [N]
    __shared__ int nrows;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const float* a = p1 + (size_t)b * N * 3;
    const float* c = p2 + (size_t)b * N * 3;
    const float* mb = mc + (size_t)b * N;
    for (int n = tid; n < N; n += nt) {
        cx[n] = c[n * 3]; cx[N + n] = c[n * 3 + 1]; cx[2 * N + n] = c[n * 3 + 2];
        pen[n] = mb[n] == 0.f ? 100.f : 0.f;
    }
    if (wave == 0) {
        int base = 0;
        for (int n0 = 0; n0 < N; n0 += 64) {
            int n = n0 + lane;
            bool m = n < N && mb[n] != 0.f;
            unsigned long long bal = __ballot(m);
            int pos = __popcll(bal & ((1ull << lane) - 1ull));
            if (m) rows[base + pos] = n;
            base += __popcll(bal);
        }
        if (lane == 0) nrows = base;
    }
    __syncthreads();
    const int nr = nrows;
    for (int r = wave; r < nr; r += nw) {
        const int i = rows[r];
        const float ax = a[i * 3], ay = a[i * 3 + 1], az = a[i * 3 + 2];
        float best = INFINITY;
        int bj = 0x7fffffff;
        for (int j = lane; j < N; j += 64) {
            float dx = ax - cx[j], dy = ay - cx[N + j], dz = az - cx[2 * N + j];
            float nrm = sqrtf(dx * dx + dy * dy + dz * dz);     // norm(...)**2 (mlsp.py:138)
            float d = nrm * nrm + pen[j];
            if (d < best) { best = d; bj = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(best, o, 64);
            int oj = __shfl_xor(bj, o, 64);
            if (ov < best || (ov == best && oj < bj)) { best = ov; bj = oj; }
        }
        if (lane == 0) { rowv[r] = best; arg[(size_t)b * N + i] = bj; }
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s += rowv[r];
        per_cloud[b * 2] = s; per_cloud[b * 2 + 1] = (float)nr;
    }
}

__global__ void chamfer_dir_finalize_kernel(const float* __restrict__ per_cloud, int B, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += per_cloud[b * 2] / per_cloud[b * 2 + 1];
        loss[0] = s;
    }
}

// dp1_i = coef (p1_i - p2_t(i)) on masked rows (0 elsewhere); dp2_j = -coef sum_{i masked, t(i) = j} (p1_i - p2_j); coef = 2 g / cnt_b
__global__ __launch_bounds__(1024) void chamfer_dir_bwd_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                               const float* __restrict__ mc, int N, const float* __restrict__ per_cloud,
                                                               const int* __restrict__ arg, const float* __restrict__ gout,
                                                               float* __restrict__ dp1, float* __restrict__ dp2) {
    extern __shared__ int esm[];
    int* rows = esm;
    int* tgt = esm + N;
    __shared__ int nrows;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const float* a = p1 + (size_t)b * N * 3;
    const float* c = p2 + (size_t)b * N * 3;
    const float* mb = mc + (size_t)b * N;
    if (wave == 0) {
        int base = 0;
        for (int n0 = 0; n0 < N; n0 += 64) {
            int n = n0 + lane;
            bool m = n < N && mb[n] != 0.f;
            unsigned long long bal = __ballot(m);
            int pos = __popcll(bal & ((1ull << lane) - 1ull));
            if (m) { rows[base + pos] = n; tgt[base + pos] = arg[(size_t)b * N + n]; }
            base += __popcll(bal);
        }
        if (lane == 0) nrows = base;
    }
    __syncthreads();
    const int nr = nrows;
    const float coef = gout[0] * 2.0f / per_cloud[b * 2 + 1];
    for (int j = tid; j < N; j += nt) {
        if (dp1) {
            float gx = 0.f, gy = 0.f, gz = 0.f;
            if (mb[j] != 0.f) {
                const int t = arg[(size_t)b * N + j];
                gx = a[j * 3] - c[t * 3]; gy = a[j * 3 + 1] - c[t * 3 + 1]; gz = a[j * 3 + 2] - c[t * 3 + 2];
            }
            float* o = dp1 + ((size_t)b * N + j) * 3;
            o[0] = coef * gx; o[1] = coef * gy; o[2] = coef * gz;
        }
        if (dp2) {
            float gx = 0.f, gy = 0.f, gz = 0.f;
            const float cxj = c[j * 3], cyj = c[j * 3 + 1], czj = c[j * 3 + 2];
            for (int r = 0; r < nr; ++r)          // fixed order
                if (tgt[r] == j) {
                    const int i = rows[r];
                    gx += cxj - a[i * 3]; gy += cyj - a[i * 3 + 1]; gz += czj - a[i * 3 + 2];
                }
            float* o = dp2 + ((size_t)b * N + j) * 3;
            o[0] = coef * gx; o[1] = coef * gy; o[2] = coef * gz;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// normal loss:  loss = -weight * sum_i w_i |cos_i| / sum_i w_i     (w == null: w_i = 1)
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if (lane == 0) { sh[w * 2] = a; sh[w * 2 + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double x = 0, y = 0;
        for (int u = 0; u < (int)(blockDim.x >> 6); ++u) { x += sh[u * 2]; y += sh[u * 2 + 1]; }
        a = x; b = y;
    }
}

__device__ __forceinline__ float cos_terms(const float* p, const float* g, float& inp, float* ph, float* gh) {
    float np = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    float ng = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    inp = fmaxf(np, 1e-12f);                 // F.normalize eps
    float ing = fmaxf(ng, 1e-12f);
    ph[0] = p[0] / inp; ph[1] = p[1] / inp; ph[2] = p[2] / inp;
    gh[0] = g[0] / ing; gh[1] = g[1] / ing; gh[2] = g[2] / ing;
    return ph[0] * gh[0] + ph[1] * gh[1] + ph[2] * gh[2];
}

__global__ __launch_bounds__(256) void normal_loss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                                  const float* __restrict__ w, int P,
                                                                  double* __restrict__ part) {
    __shared__ double sh[8];
    double s = 0.0, sw = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        float inp, ph[3], gh[3];
        float c = cos_terms(pred + (size_t)i * 3, gt + (size_t)i * 3, inp, ph, gh);
        float wi = w ? w[i] : 1.f;
        s += (double)fabsf(c) * wi; sw += wi;
    }
    block_sum2(s, sw, sh);
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = s; part[blockIdx.x * 2 + 1] = sw; }
}

// out[0] = loss, out[1] = sum w  (kept for backward)
// (one wave: lanes stride the partials, then a fixed fp64 shuffle tree -- reproducible)
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(64) void normal_loss_finalize_kernel(const double* __restrict__ part, int nparts, float weight,
                                                                  float* __restrict__ out) {
    double s = 0, sw = 0;
    for (int i = threadIdx.x; i < nparts; i += 64) { s += part[i * 2]; sw += part[i * 2 + 1]; }
    s = wave_sum_f64(s); sw = wave_sum_f64(sw);
    if (threadIdx.x == 0) {
        out[0] = (float)(-(double)weight * s / sw);
        out[1] = (float)sw;
    }
}

__global__ __launch_bounds__(256) void normal_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                              const float* __restrict__ w, int P,
                                                              const float* __restrict__ fwd_out, const float* __restrict__ gout,
                                                              float weight, float* __restrict__ dpred) {
    const float coef = -gout[0] * weight / fwd_out[1];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        float inp, ph[3], gh[3];
        float c = cos_terms(pred + (size_t)i * 3, gt + (size_t)i * 3, inp, ph, gh);
        float sg = (c > 0.f) - (c < 0.f);
        float f = coef * (w ? w[i] : 1.f) * sg / inp;
        float* o = dpred + (size_t)i * 3;
        o[0] = f * (gh[0] - c * ph[0]); o[1] = f * (gh[1] - c * ph[1]); o[2] = f * (gh[2] - c * ph[2]);
    }
}

// ---------------------------------------------------------------------------------------------
// cardinality head tail:  p = softmax(logits) ; density = sum_c p_c * w_c      (nc <= 64)
__global__ __launch_bounds__(256) void density_tail_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ w,
                                                               int P, int nc, float* __restrict__ pvec,
                                                               float* __restrict__ dens) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        const float* l = logits + (size_t)i * nc;
        float m = l[0];
        for (int c = 1; c < nc; ++c) m = fmaxf(m, l[c]);
        float s = 0.f;
        for (int c = 0; c < nc; ++c) s += expf(l[c] - m);
        float inv = 1.f / s, d = 0.f;
        for (int c = 0; c < nc; ++c) {
            float p = expf(l[c] - m) * inv;
            pvec[(size_t)i * nc + c] = p;
            d = fmaf(p, w[c], d);
        }
        dens[i] = d;
    }
}

// dlogits_c = p_c * (g_c - sum_c' p_c' g_c'),  g_c = dp_c + dd * w_c
__global__ __launch_bounds__(256) void density_tail_bwd_kernel(const float* __restrict__ pvec, const float* __restrict__ w,
                                                               const float* __restrict__ dp, const float* __restrict__ dd,
                                                               int P, int nc, float* __restrict__ dlogits) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        const float* p = pvec + (size_t)i * nc;
        float ddi = dd ? dd[i] : 0.f;
        float dot = 0.f;
        for (int c = 0; c < nc; ++c) {
            float g = (dp ? dp[(size_t)i * nc + c] : 0.f) + ddi * w[c];
            dot = fmaf(p[c], g, dot);
        }
        for (int c = 0; c < nc; ++c) {
            float g = (dp ? dp[(size_t)i * nc + c] : 0.f) + ddi * w[c];
            dlogits[(size_t)i * nc + c] = p[c] * (g - dot);
        }
    }
}

// densityloss partials: {sum m*sum_c t log(p+1e-10), sum m*|d-target|} and sum m  (m == null: 1)
__global__ __launch_bounds__(256) void density_loss_partial_kernel(const float* __restrict__ pvec, const float* __restrict__ dens,
                                                                   const float* __restrict__ tvec, const float* __restrict__ target,
                                                                   const float* __restrict__ m, int P, int nc,
                                                                   double* __restrict__ part) {
    __shared__ double sh[8];
    __shared__ double sh2[8];
    double a = 0.0, b = 0.0, sm = 0.0, dummy = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        float ce = 0.f;
        for (int c = 0; c < nc; ++c) ce = fmaf(tvec[(size_t)i * nc + c], logf(pvec[(size_t)i * nc + c] + 1e-10f), ce);
        float mi = m ? m[i] : 1.f;
        a += (double)ce * mi; b += (double)fabsf(dens[i] - target[i]) * mi; sm += mi;
    }
    block_sum2(a, b, sh);
    __syncthreads();
    block_sum2(sm, dummy, sh2);
    if (threadIdx.x == 0) { part[blockIdx.x * 3] = a; part[blockIdx.x * 3 + 1] = b; part[blockIdx.x * 3 + 2] = sm; }
}

// out = {kl, mae, sum m}:  kl = -Dw * a / sm ;  mae = Dw * 0.05 * b / sm
__global__ __launch_bounds__(64) void density_loss_finalize_kernel(const double* __restrict__ part, int nparts, float dweight,
                                                                   float* __restrict__ out) {
    double a = 0, b = 0, sm = 0;
    for (int i = threadIdx.x; i < nparts; i += 64) { a += part[i * 3]; b += part[i * 3 + 1]; sm += part[i * 3 + 2]; }
    a = wave_sum_f64(a); b = wave_sum_f64(b); sm = wave_sum_f64(sm);
    if (threadIdx.x == 0) {
        out[0] = (float)(-(double)dweight * a / sm);
        out[1] = (float)((double)dweight * 0.05 * b / sm);
        out[2] = (float)sm;
    }
}

__global__ __launch_bounds__(256) void density_loss_bwd_kernel(const float* __restrict__ pvec, const float* __restrict__ dens,
                                                               const float* __restrict__ tvec, const float* __restrict__ target,
                                                               const float* __restrict__ m, int P, int nc,
                                                               const float* __restrict__ fwd_out, const float* __restrict__ gkl,
                                                               const float* __restrict__ gmae, float dweight,
                                                               float* __restrict__ dp, float* __restrict__ dd) {
    const float sm = fwd_out[2];
    const float ck = gkl ? -gkl[0] * dweight / sm : 0.f;
    const float cm = gmae ? gmae[0] * dweight * 0.05f / sm : 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        float mi = m ? m[i] : 1.f;
        for (int c = 0; c < nc; ++c)
            dp[(size_t)i * nc + c] = ck * mi * tvec[(size_t)i * nc + c] / (pvec[(size_t)i * nc + c] + 1e-10f);
        float df = dens[i] - target[i];
        dd[i] = cm * mi * (float)((df > 0.f) - (df < 0.f));
    }
}

// ---------------------------------------------------------------------------------------------
#define LOSS_PARTS 256

int launch_chamfer_fwd(hipStream_t st, const float* pred, const float* gold, const float* mask, int B, int N, float scale,
                       float* per_cloud, int* argA, int* argB, float* loss) {
    size_t lds = (size_t)10 * N * sizeof(float);
    if (lds > 150 * 1024) return MLSP_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = mlsp_lds_limit((const void*)chamfer_fwd_kernel, lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(chamfer_fwd_kernel, dim3(B), dim3(1024), lds, st, pred, gold, mask, N, per_cloud, argA, argB);
    hipLaunchKernelGGL(chamfer_finalize_kernel, dim3(1), dim3(64), 0, st, per_cloud, B, scale, loss);
    return mlsp_launch_status();
}
int launch_chamfer_bwd(hipStream_t st, const float* pred, const float* gold, const float* mask, int B, int N, float scale,
                       const float* per_cloud, const int* argA, const int* argB, const float* gout, float* dpred) {
    size_t lds = (size_t)2 * N * sizeof(int);
    hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(B), dim3(1024), lds, st, pred, gold, mask, N, per_cloud, argA, argB, gout, scale,
                       dpred);
    return mlsp_launch_status();
}
int launch_chamfer_dir_fwd(hipStream_t st, const float* p1, const float* p2, const float* mc, int B, int N, float* per_cloud, int* arg,
                           float* loss) {
    size_t lds = (size_t)6 * N * sizeof(float);
    if (lds > 150 * 1024) return MLSP_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = mlsp_lds_limit((const void*)chamfer_dir_fwd_kernel, lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(chamfer_dir_fwd_kernel, dim3(B), dim3(1024), lds, st, p1, p2, mc, N, per_cloud, arg);
    hipLaunchKernelGGL(chamfer_dir_finalize_kernel, dim3(1), dim3(64), 0, st, per_cloud, B, loss);
    return mlsp_launch_status();
}
int launch_chamfer_dir_bwd(hipStream_t st, const float* p1, const float* p2, const float* mc, int B, int N, const float* per_cloud,
                           const int* arg, const float* gout, float* dp1, float* dp2) {
    size_t lds = (size_t)2 * N * sizeof(int);
    if (lds > 64 * 1024) return MLSP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(chamfer_dir_bwd_kernel, dim3(B), dim3(1024), lds, st, p1, p2, mc, N, per_cloud, arg, gout, dp1, dp2);
    return mlsp_launch_status();
}
int launch_normal_loss_fwd(hipStream_t st, const float* pred, const float* gt, const float* w, int P, float weight,
                           double* part, float* out) {
    hipLaunchKernelGGL(normal_loss_partial_kernel, dim3(LOSS_PARTS), dim3(256), 0, st, pred, gt, w, P, part);
    hipLaunchKernelGGL(normal_loss_finalize_kernel, dim3(1), dim3(64), 0, st, part, LOSS_PARTS, weight, out);
    return mlsp_launch_status();
}
int launch_normal_loss_bwd(hipStream_t st, const float* pred, const float* gt, const float* w, int P, float weight,
                           const float* fwd_out, const float* gout, float* dpred) {
    hipLaunchKernelGGL(normal_loss_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, st, pred, gt, w, P, fwd_out, gout, weight,
                       dpred);
    return mlsp_launch_status();
}
int launch_density_tail_fwd(hipStream_t st, const float* logits, const float* w, int P, int nc, float* pvec, float* dens) {
    hipLaunchKernelGGL(density_tail_fwd_kernel, dim3((P + 255) / 256), dim3(256), 0, st, logits, w, P, nc, pvec, dens);
    return mlsp_launch_status();
}
int launch_density_tail_bwd(hipStream_t st, const float* pvec, const float* w, const float* dp, const float* dd, int P, int nc,
                            float* dlogits) {
    hipLaunchKernelGGL(density_tail_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, st, pvec, w, dp, dd, P, nc, dlogits);
    return mlsp_launch_status();
}
int launch_density_loss_fwd(hipStream_t st, const float* pvec, const float* dens, const float* tvec, const float* target,
                            const float* m, int P, int nc, float dweight, double* part, float* out) {
    hipLaunchKernelGGL(density_loss_partial_kernel, dim3(LOSS_PARTS), dim3(256), 0, st, pvec, dens, tvec, target, m, P, nc, part);
    hipLaunchKernelGGL(density_loss_finalize_kernel, dim3(1), dim3(64), 0, st, part, LOSS_PARTS, dweight, out);
    return mlsp_launch_status();
}
int launch_density_loss_bwd(hipStream_t st, const float* pvec, const float* dens, const float* tvec, const float* target,
                            const float* m, int P, int nc, float dweight, const float* fwd_out, const float* gkl,
                            const float* gmae, float* dp, float* dd) {
    hipLaunchKernelGGL(density_loss_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, st, pvec, dens, tvec, target, m, P, nc,
                       fwd_out, gkl, gmae, dweight, dp, dd);
    return mlsp_launch_status();
}
