// PointNet++ set-abstraction front end (BASELINE.json configs[3]): farthest point sampling, ball query and
// grouping.  The reference states these semantics only in PointDA/hengshuang_transformer/pointnet_util.py (:53-73 FPS,
// :76-96 query_ball_point, :99-136 sample_and_group); the pointnet2_ops CUDA library it otherwise leans on is not used
// for them.  Index results are bit-exact against that torch code: same fp32 expression ((dx*dx + dy*dy) + dz*dz, no
// contraction), same tie rule (first index), same "first nsample in index order, padded with the first" selection.
// The SA-MLP behind the grouping is the existing Linear+BN+act kernel family over the B*S*nsample edge rows.
#include "common.h"

// ---- FPS: one workgroup per cloud, the cloud in LDS, running distances in registers.
// out[b][i] = index of the i-th sample; sample 0 is start[b].  N <= 8 * 1024.
#define FPS_T 1024
#define FPS_PPT 8
__global__ __launch_bounds__(FPS_T) void fps_kernel(const float* __restrict__ xyz, int ldx, int N, int S, const int* __restrict__ start,
                                                    int* __restrict__ out) {
    extern __shared__ float fsm[];
    float* cx = fsm;                 // [N][3]
    float* wv = fsm + 3 * N;         // [16] per-wave best value
    int* wi = (int*)(wv + 16);       // [16] per-wave best index
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = xyz + (size_t)b * N * ldx;
    for (int j = tid; j < N; j += FPS_T) {
        cx[3 * j + 0] = xb[(size_t)j * ldx + 0]; cx[3 * j + 1] = xb[(size_t)j * ldx + 1]; cx[3 * j + 2] = xb[(size_t)j * ldx + 2];
    }
    float dmin[FPS_PPT];
#pragma unroll
    for (int u = 0; u < FPS_PPT; ++u) dmin[u] = 1e10f;
    __syncthreads();
    int far = start[b];
    for (int i = 0; i < S; ++i) {
        if (tid == 0) out[(size_t)b * S + i] = far;
        const float fx = cx[3 * far], fy = cx[3 * far + 1], fz = cx[3 * far + 2];
        float best = -1.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < FPS_PPT; ++u) {
            const int j = tid + FPS_T * u;
            if (j < N) {
                const float dx = cx[3 * j] - fx, dy = cx[3 * j + 1] - fy, dz = cx[3 * j + 2] - fz;
                const float d = (dx * dx + dy * dy) + dz * dz;
                dmin[u] = fminf(dmin[u], d);
                if (dmin[u] > best) { best = dmin[u]; bi = j; }       // j ascending within a thread: first index wins ties
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { wv[wave] = best; wi[wave] = bi; }
        __syncthreads();
        if (wave == 0) {
            float v = lane < FPS_T / 64 ? wv[lane] : -1.f;
            int ix = lane < FPS_T / 64 ? wi[lane] : 0x7fffffff;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                const float ov = __shfl_xor(v, o, 64);
                const int oi = __shfl_xor(ix, o, 64);
                if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
            }
            if (lane == 0) wi[0] = ix;
        }
        __syncthreads();
        far = wi[0];
        __syncthreads();                       // wi[0] is rewritten by the next iteration's wave results
    }
}

// ---- FPS, round 2: the per-sample critical path is one arg-max over the cloud.  The kernel above spends it in ~20 dependent
// ds_bpermute shuffles and two workgroup barriers (1.6 us per sample).  Here: 4 waves, <= 8 points per lane in registers, the arg-max
// as a max over ONE 64-bit key (distance bits << 32 | ~index: distances are >= 0 so their bit patterns order like the floats, and the
// larger ~index is the smaller index = the reference's first maximum) reduced inside a wave by DPP row / bank shifts (no LDS
// crossbar), across the 4 waves through a double-buffered LDS slot and ONE barrier.  Same distance expression, same running
// minimum, same tie rule: identical indices.  N <= 2048.
__device__ __forceinline__ unsigned long long fps_key_max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
template <int CTRL>
__device__ __forceinline__ unsigned long long fps_dpp64(unsigned long long v) {        // lanes without a source keep their own value
    const unsigned lo = __builtin_amdgcn_update_dpp((int)(unsigned)v, (int)(unsigned)v, CTRL, 0xf, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp((int)(unsigned)(v >> 32), (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, false);
    return ((unsigned long long)hi << 32) | lo;
}
__global__ __launch_bounds__(256) void fps_kernel2(const float* __restrict__ xyz, int ldx, int N, int S, const int* __restrict__ start,
                                                   int* __restrict__ out) {
    extern __shared__ float fsm[];
    float* cx = fsm;                                             // [N][3]
    unsigned long long* slot = (unsigned long long*)(fsm + 3 * ((N + 1) & ~1));   // [2][4] wave maxima, double-buffered
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = xyz + (size_t)b * N * ldx;
    float px[8], py[8], pz[8], dmin[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int j = tid + 256 * u;
        px[u] = py[u] = pz[u] = 0.f; dmin[u] = 1e10f;
        if (j < N) {
            px[u] = xb[(size_t)j * ldx]; py[u] = xb[(size_t)j * ldx + 1]; pz[u] = xb[(size_t)j * ldx + 2];
            cx[3 * j] = px[u]; cx[3 * j + 1] = py[u]; cx[3 * j + 2] = pz[u];
        }
    }
    __syncthreads();
    int far = start[b];
    for (int i = 0; i < S; ++i) {
        if (tid == 0) out[(size_t)b * S + i] = far;
        const float fx = cx[3 * far], fy = cx[3 * far + 1], fz = cx[3 * far + 2];
        unsigned long long key = 0ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = tid + 256 * u;
            if (j < N) {
                const float dx = px[u] - fx, dy = py[u] - fy, dz = pz[u] - fz;
                const float d = (dx * dx + dy * dy) + dz * dz;
                dmin[u] = fminf(dmin[u], d);
                key = fps_key_max(key, ((unsigned long long)__float_as_uint(dmin[u]) << 32) | (unsigned)(~j));
            }
        }
        // wave max by DPP: within rows of 16 (shr 1, 2, 3 + combine), then row broadcasts 15 and 31 -> lane 63 holds the wave's maximum
        key = fps_key_max(key, fps_dpp64<0x111>(key));           // row_shr:1
        key = fps_key_max(key, fps_dpp64<0x112>(key));           // row_shr:2
        key = fps_key_max(key, fps_dpp64<0x114>(key));           // row_shr:4
        key = fps_key_max(key, fps_dpp64<0x118>(key));           // row_shr:8   -> lane 15 of every row: row maximum
        key = fps_key_max(key, fps_dpp64<0x142>(key));           // row_bcast:15 -> rows 1..3 see the previous row's maximum
        key = fps_key_max(key, fps_dpp64<0x143>(key));           // row_bcast:31 -> lane 63: maximum of all four rows
        if (lane == 63) slot[(i & 1) * 4 + wave] = key;
        __syncthreads();
        const unsigned long long* sl = slot + (i & 1) * 4;
        const unsigned long long best = fps_key_max(fps_key_max(sl[0], sl[1]), fps_key_max(sl[2], sl[3]));
        far = (int)(~(unsigned)best);
    }
}

// ---- ball query: one wave per query; candidates scanned in index order, 64 at a time.
// idx[b][i][0..nsample): the first nsample j with !(|q_i - x_j|^2 > r2), padded with the first hit.
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ xyz, int ldx, const float* __restrict__ q, int ldq, int N,
                                                         int S, float r2, int nsample, int* __restrict__ idx) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= S) return;
    const float* xb = xyz + (size_t)b * N * ldx;
    const float* qi = q + ((size_t)b * S + i) * ldq;
    const float qx = qi[0], qy = qi[1], qz = qi[2];
    int* o = idx + ((size_t)b * S + i) * nsample;
    int cnt = 0, first = 0;
    for (int j0 = 0; j0 < N && cnt < nsample; j0 += 64) {
        const int j = j0 + lane;
        bool in = false;
        if (j < N) {
            const float dx = qx - xb[(size_t)j * ldx], dy = qy - xb[(size_t)j * ldx + 1], dz = qz - xb[(size_t)j * ldx + 2];
            const float d = (dx * dx + dy * dy) + dz * dz;
            in = !(d > r2);
        }
        const unsigned long long m = __ballot(in);
        if (m) {
            if (cnt == 0) first = j0 + (int)__builtin_ctzll(m);
            const int pos = cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (in && pos < nsample) o[pos] = j;
            cnt += __builtin_popcountll(m);
        }
    }
    if (cnt > nsample) cnt = nsample;
    for (int s = cnt + lane; s < nsample; s += 64) o[s] = first;
}

// ---- grouping: G[(b,i,s)][0:3] = xyz[j] - new_xyz[i],  G[.][3:3+D] = feat[j],  j = idx[b][i][s]   (pointnet_util.py:120-129)
__global__ __launch_bounds__(256) void sa_group_fwd_kernel(const float* __restrict__ xyz, int ldx, const float* __restrict__ feat, int D,
                                                           const float* __restrict__ q, int ldq, const int* __restrict__ idx, int N, int S,
                                                           int ns, size_t total, float* __restrict__ G) {
    const int C = 3 + D;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t e = t / C;
        const int c = (int)(t % C);
        const size_t bi = e / ns;                    // b*S + i
        const int b = (int)(bi / S);
        const int j = idx[e];
        const size_t src = (size_t)b * N + j;
        G[t] = c < 3 ? xyz[src * ldx + c] - q[bi * ldq + c] : feat[src * D + (c - 3)];
    }
}

// dfeat[b][j][:] = sum over the (i, s) with idx[b][i][s] == j of dG[(b,i,s)][col : col + D]  (rows of pitch ldg) -- walks the reverse
// index (fixed order).  col = 3: the feature columns of the grouped rows; col = 0, D = 3: their coordinate columns (d xyz).
__global__ __launch_bounds__(256) void sa_group_bwd_kernel(const float* __restrict__ dG, int ldg, int col, int D, const int* __restrict__ rev_off,
                                                           const int* __restrict__ rev_ent, int N, int S, int ns, int P,
                                                           float* __restrict__ dfeat) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= P) return;
    const int b = j / N;
    const int e0 = rev_off[j], e1 = rev_off[j + 1];
    for (int c0 = 0; c0 < D; c0 += 64) {
        const int c = c0 + lane;
        float acc = 0.f;
        int e = e0;
        for (; e + 4 <= e1; e += 4) {                    // four rows in flight (padded groups make some lists very long)
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ent = rev_ent[e + u];
                const size_t row = ((size_t)b * S + (ent >> 8)) * ns + (ent & 255);
                t[u] = c < D ? dG[row * ldg + col + c] : 0.f;
            }
            acc += (t[0] + t[1]) + (t[2] + t[3]);
        }
        for (; e < e1; ++e) {
            const int ent = rev_ent[e];
            const size_t row = ((size_t)b * S + (ent >> 8)) * ns + (ent & 255);
            if (c < D) acc += dG[row * ldg + col + c];
        }
        if (c < D) dfeat[(size_t)j * D + c] = acc;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
int launch_fps(hipStream_t st, const float* xyz, int ldx, int B, int N, int S, const int* start, int* out) {
    if (!xyz || !start || !out || B <= 0 || N <= 0 || S <= 0 || S > N || ldx < 3 || N > FPS_T * FPS_PPT) return MLSP_ERR_ARG;
    size_t lds = ((size_t)3 * N + 32) * sizeof(float);
    if (lds > 150 * 1024) return MLSP_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = mlsp_lds_limit((const void*)fps_kernel, lds);
        if (e != hipSuccess) return (int)e;
    }
    if (N <= 2048 && N >= 64) {
        const size_t lds2 = ((size_t)3 * ((N + 1) & ~1)) * sizeof(float) + 8 * sizeof(unsigned long long);
        hipLaunchKernelGGL(fps_kernel2, dim3(B), dim3(256), lds2, st, xyz, ldx, N, S, start, out);
        return mlsp_launch_status();
    }
    hipLaunchKernelGGL(fps_kernel, dim3(B), dim3(FPS_T), lds, st, xyz, ldx, N, S, start, out);
    return mlsp_launch_status();
}

int launch_ball_query(hipStream_t st, const float* xyz, int ldx, const float* q, int ldq, int B, int N, int S, float r2, int nsample,
                      int* idx) {
    if (!xyz || !q || !idx || B <= 0 || N <= 0 || S <= 0 || nsample <= 0 || nsample > 255 || ldx < 3 || ldq < 3) return MLSP_ERR_ARG;
    hipLaunchKernelGGL(ball_query_kernel, dim3((S + 3) / 4, B), dim3(256), 0, st, xyz, ldx, q, ldq, N, S, r2, nsample, idx);
    return mlsp_launch_status();
}

int launch_sa_group_fwd(hipStream_t st, const float* xyz, int ldx, const float* feat, int D, const float* q, int ldq, const int* idx,
                        int B, int N, int S, int ns, float* G) {
    if (!xyz || !q || !idx || !G || (D > 0 && !feat) || D < 0 || B <= 0 || N <= 0 || S <= 0 || ns <= 0) return MLSP_ERR_ARG;
    const size_t total = (size_t)B * S * ns * (3 + D);
    const size_t nb = (total + 255) / 256;
    hipLaunchKernelGGL(sa_group_fwd_kernel, dim3((unsigned)(nb < 65536 ? nb : 65536)), dim3(256), 0, st, xyz, ldx, feat, D, q, ldq, idx, N, S,
                       ns, total, G);
    return mlsp_launch_status();
}

int launch_sa_group_bwd(hipStream_t st, const float* dG, int ldg, int col, int D, const int* rev_off, const int* rev_ent, int B, int N, int S,
                        int ns, float* dfeat) {
    if (!dG || !rev_off || !rev_ent || !dfeat || D <= 0 || col < 0 || ldg < col + D || B <= 0 || N <= 0 || S <= 0 || ns <= 0 || ns > 255)
        return MLSP_ERR_ARG;
    const int P = B * N;
    hipLaunchKernelGGL(sa_group_bwd_kernel, dim3((P + 3) / 4), dim3(256), 0, st, dG, ldg, col, D, rev_off, rev_ent, N, S, ns, P, dfeat);
    return mlsp_launch_status();
}

// ---- kNN of QUERY points among REFERENCE points (ref != query): `square_distance` + argsort()[:, :, :k] of
// pointnet_util.py:26-38,116-118 (knn=True grouping, :237-239 Msg) and the 3-NN of PointNetFeaturePropagation (:287-289);
// also the shape of knn_cuda.KNN / the KNN calls of PointDA/model_utils.py:175,188.
//   d(q, r) = fl( fl(-2 * dot(q, r) + |q|^2) + |r|^2 ),  dot an fmaf chain over the C <= 8 coordinates, norms (x0^2 + x1^2) + ...
//   order: d ascending, ties -> lower reference index; the k best, nearest first.
// One thread per query with a sorted k-list in registers; the references stream through LDS in tiles of 256.
template <int KMAX>
__global__ __launch_bounds__(256) void knn_query_kernel(const float* __restrict__ ref, int ldr, int Nr, const float* __restrict__ qry,
                                                        int ldq, int Nq, int C, int k, int* __restrict__ idx, float* __restrict__ dist) {
    __shared__ float rs[256 * 8];
    __shared__ float rn[256];
    const int b = blockIdx.y, tid = threadIdx.x, q = blockIdx.x * 256 + tid;
    const float* rb = ref + (size_t)b * Nr * ldr;
    float qv[8], qn = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) qv[c] = 0.f;
    if (q < Nq) {
        const float* qp = qry + ((size_t)b * Nq + q) * ldq;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) { qv[c] = qp[c]; qn = c == 0 ? qv[0] * qv[0] : qn + qv[c] * qv[c]; }
    }
    float bv[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) { bv[s] = INFINITY; bi[s] = 0x7fffffff; }
    for (int r0 = 0; r0 < Nr; r0 += 256) {
        __syncthreads();
        const int r = r0 + tid;
        float nn = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float v = (r < Nr && c < C) ? rb[(size_t)r * ldr + c] : 0.f;
            rs[tid * 8 + c] = v;
            if (c < C) nn = c == 0 ? v * v : nn + v * v;
        }
        rn[tid] = nn;
        __syncthreads();
        const int lim = min(256, Nr - r0);
        for (int j = 0; j < lim; ++j) {
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < C) dot = fmaf(qv[c], rs[j * 8 + c], dot);
            const float d = (-2.f * dot + qn) + rn[j];
            if (d < bv[KMAX - 1]) {                                     // strict: an equal distance with a higher index stays out
                const int jj = r0 + j;
#pragma unroll
                for (int s = KMAX - 1; s >= 1; --s) {
                    const bool up = d < bv[s - 1];
                    const bool here = d < bv[s];
                    const float nv = up ? bv[s - 1] : (here ? d : bv[s]);
                    const int ni = up ? bi[s - 1] : (here ? jj : bi[s]);
                    bv[s] = nv; bi[s] = ni;
                }
                if (d < bv[0]) { bv[0] = d; bi[0] = jj; }
            }
        }
    }
    if (q < Nq) {
#pragma unroll
        for (int s = 0; s < KMAX; ++s)
            if (s < k) {
                idx[((size_t)b * Nq + q) * k + s] = bi[s];
                if (dist) dist[((size_t)b * Nq + q) * k + s] = bv[s];
            }
    }
}

int launch_knn_query(hipStream_t st, const float* ref, int ldr, int Nr, const float* qry, int ldq, int Nq, int B, int C, int k, int* idx,
                     float* dist) {
    if (!ref || !qry || !idx || B <= 0 || Nr <= 0 || Nq <= 0 || C <= 0 || C > 8 || ldr < C || ldq < C || k <= 0 || k > Nr) return MLSP_ERR_ARG;
    if (k > 64) return MLSP_ERR_UNSUPPORTED;
    const dim3 grid((Nq + 255) / 256, B);
    if (k <= 4) hipLaunchKernelGGL((knn_query_kernel<4>), grid, dim3(256), 0, st, ref, ldr, Nr, qry, ldq, Nq, C, k, idx, dist);
    else if (k <= 16) hipLaunchKernelGGL((knn_query_kernel<16>), grid, dim3(256), 0, st, ref, ldr, Nr, qry, ldq, Nq, C, k, idx, dist);
    else if (k <= 32) hipLaunchKernelGGL((knn_query_kernel<32>), grid, dim3(256), 0, st, ref, ldr, Nr, qry, ldq, Nq, C, k, idx, dist);
    else hipLaunchKernelGGL((knn_query_kernel<64>), grid, dim3(256), 0, st, ref, ldr, Nr, qry, ldq, Nq, C, k, idx, dist);
    return mlsp_launch_status();
}

// ---- PointNetFeaturePropagation interpolation (pointnet_util.py:287-294): out[b][n] = sum_t w_t * feat[b][idx[n][t]],
// w_t = (1 / (d_t + 1e-8)) / sum_t' (1 / (d_t' + 1e-8)) over the three nearest sampled points.  Thread = (point, channel quad).
__global__ __launch_bounds__(256) void interp3_fwd_kernel(const float* __restrict__ feat, const int* __restrict__ idx,
                                                          const float* __restrict__ dist, int N, int S, int D, size_t total,
                                                          float* __restrict__ out) {
    const int dq = (D + 3) >> 2;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t pn = t / dq; const int c = (int)(t % dq) * 4;
        const int b = (int)(pn / N);
        float w[3], ws = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u) { w[u] = 1.0f / (dist[pn * 3 + u] + 1e-8f); ws += w[u]; }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const float wu = w[u] / ws;
            const float* f = feat + ((size_t)b * S + idx[pn * 3 + u]) * D + c;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < D) acc[e] += f[e] * wu;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < D) out[pn * D + c + e] = acc[e];
    }
}
// backward over the reverse index of idx [B][N][3] (mlsp_group_reverse with the roles S <- N, N <- S): wave per source point
__global__ __launch_bounds__(256) void interp3_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ dist,
                                                          const int* __restrict__ rev_off, const int* __restrict__ rev_ent, int N, int S,
                                                          int D, int total_src, float* __restrict__ dfeat) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= total_src) return;
    const int b = s / S;
    const int e0 = rev_off[s], e1 = rev_off[s + 1];
    for (int c = lane; c < D; c += 64) {
        float acc = 0.f;
        for (int e = e0; e < e1; ++e) {
            const int ent = rev_ent[e];
            const size_t pn = (size_t)b * N + (ent >> 8);
            const int u = ent & 255;
            float ws = 0.f, wu = 0.f;
#pragma unroll
            for (int v = 0; v < 3; ++v) { const float w = 1.0f / (dist[pn * 3 + v] + 1e-8f); ws += w; wu = v == u ? w : wu; }
            acc += dout[pn * D + c] * (wu / ws);
        }
        dfeat[(size_t)s * D + c] = acc;
    }
}
int launch_interp3_fwd(hipStream_t st, const float* feat, const int* idx, const float* dist, int B, int N, int S, int D, float* out) {
    if (!feat || !idx || !dist || !out || B <= 0 || N <= 0 || S < 3 || D <= 0) return MLSP_ERR_ARG;
    const size_t total = (size_t)B * N * ((D + 3) / 4);
    const size_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(interp3_fwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, feat, idx, dist, N, S, D, total, out);
    return mlsp_launch_status();
}
int launch_interp3_bwd(hipStream_t st, const float* dout, const float* dist, const int* rev_off, const int* rev_ent, int B, int N, int S,
                       int D, float* dfeat) {
    if (!dout || !dist || !rev_off || !rev_ent || !dfeat || B <= 0 || N <= 0 || S <= 0 || D <= 0) return MLSP_ERR_ARG;
    hipLaunchKernelGGL(interp3_bwd_kernel, dim3((B * S + 3) / 4), dim3(256), 0, st, dout, dist, rev_off, rev_ent, N, S, D, B * S, dfeat);
    return mlsp_launch_status();
}

// ---- folded first layer of a set-abstraction MLP (pointnet_util.py:120-129,185-190) ---------------------------------
// The first 1x1 conv acts on the edge rows [x_j - c_i | f_j]:  W [x_j - c_i ; f_j] + b = u_j - w_i  with  u = [x | f] W^T + b  per
// SOURCE point (B*N rows) and  w = c Wx^T  per CENTRE (B*S rows): two small per-point GEMMs instead of one over the B*S*ns edge rows,
// and neither the grouped tensor [E, 3+D] nor the pre-BN output [E, C] is written.  BatchNorm batch statistics run over the E edges
// (pass 1: gather u_j - w_i and reduce), pass 2 gathers again (the u rows stay in L2) and writes z = relu(scale*y + shift) [E, C], the
// operand of the second conv.  Backward recomputes y the same way: dz' = dz*[scale*y + shift > 0], closed-form BN backward
// dy = scale*(dz' - mean(dz') - yhat*mean(dz'*yhat)), then  dw_i = -sum_s dy_(i,s)  (a centre's edges are consecutive rows) and
// du_j = sum over the edges that point at j (reverse index, fixed order: bitwise reproducible, no float atomics).
// Thread = (channel quad, edge lane); C % 4 == 0, 256 % (C / 4) == 0.
#define SAF_EPB 256          // edges per workgroup of the edge passes
template <int MODE>          // 0: statistics of y | 1: z = relu(scale*y + shift) | 2: sums of dz' and dz'*yhat
__global__ __launch_bounds__(256) void sa_fold_edge_kernel(const float* __restrict__ u, const float* __restrict__ w, const int* __restrict__ idx,
                                                           int N, int S, int ns, int C, long E, const float* __restrict__ bn /* scale|shift|mean|invstd */,
                                                           const float* __restrict__ dZ, float* __restrict__ Z, double* __restrict__ part) {
    __shared__ float shd[2][256 * 4];
    const int tid = threadIdx.x, tpr = C >> 2, nel = 256 / tpr;
    const int cq = tid % tpr, el = tid / tpr, c = 4 * cq;
    f32x4 sc = {0, 0, 0, 0}, sh = {0, 0, 0, 0}, mu = {0, 0, 0, 0}, is = {0, 0, 0, 0};
    if (MODE != 0) { sc = *(const f32x4*)(bn + c); sh = *(const f32x4*)(bn + C + c); }
    if (MODE == 2) { mu = *(const f32x4*)(bn + 2 * C + c); is = *(const f32x4*)(bn + 3 * C + c); }
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    const long e0 = (long)blockIdx.x * SAF_EPB;
    const long e1 = e0 + SAF_EPB < E ? e0 + SAF_EPB : E;
    for (long eb = e0 + el; eb < e1; eb += 4L * nel) {                     // four edges per thread in flight
        f32x4 uv[4], wv[4], dz[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long e = eb + (long)t * nel;
            const bool ok = e < e1;
            const long ee = ok ? e : e0;
            const long ctr = ee / ns;                                       // global centre row b*S + i
            const long cloud = ctr / S;
            const int j = idx[ee];
            uv[t] = *(const f32x4*)(u + ((size_t)cloud * N + j) * C + c);
            wv[t] = *(const f32x4*)(w + (size_t)ctr * C + c);
            if (MODE == 2) dz[t] = *(const f32x4*)(dZ + (size_t)ee * C + c);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long e = eb + (long)t * nel;
            if (e >= e1) continue;
            f32x4 y = uv[t] - wv[t];
            if (MODE == 0) {
                a0 = a0 + y;
#pragma unroll
                for (int q = 0; q < 4; ++q) a1[q] = fmaf(y[q], y[q], a1[q]);
            } else if (MODE == 1) {
                f32x4 z;
#pragma unroll
                for (int q = 0; q < 4; ++q) z[q] = fmaxf(fmaf(y[q], sc[q], sh[q]), 0.f);
                *(f32x4*)(Z + (size_t)e * C + c) = z;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float d = fmaf(y[q], sc[q], sh[q]) > 0.f ? dz[t][q] : 0.f;
                    a0[q] += d;
                    a1[q] = fmaf(d, (y[q] - mu[q]) * is[q], a1[q]);
                }
            }
        }
    }
    if (MODE == 1) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) { shd[0][tid * 4 + q] = a0[q]; shd[1][tid * 4 + q] = a1[q]; }
    __syncthreads();
    if (tid < tpr) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double s0 = 0.0, s1 = 0.0;
            for (int g = 0; g < nel; ++g) { s0 += (double)shd[0][(g * tpr + tid) * 4 + q]; s1 += (double)shd[1][(g * tpr + tid) * 4 + q]; }
            part[((size_t)blockIdx.x * 2 + 0) * C + 4 * tid + q] = s0;
            part[((size_t)blockIdx.x * 2 + 1) * C + 4 * tid + q] = s1;
        }
    }
}

// dy of one edge, channel quad: scale * (dz' - m1 - yhat * m2)   (eval mode: m1 = m2 = 0)
__device__ __forceinline__ f32x4 saf_dy(const f32x4& uq, const f32x4& wq, const f32x4& dz, const f32x4& sc, const f32x4& sh, const f32x4& mu,
                                        const f32x4& is, const f32x4& m1, const f32x4& m2) {
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float y = uq[q] - wq[q];
        const float d = fmaf(y, sc[q], sh[q]) > 0.f ? dz[q] : 0.f;
        r[q] = sc[q] * (d - m1[q] - (y - mu[q]) * is[q] * m2[q]);
    }
    return r;
}

// dw_i = -sum over the ns edges of centre i.  Workgroup = 256 / (C/4) centres' worth of edge lanes: thread (cq, el) sums the slots
// el, el + nel, ... of ONE centre per pass; the nel partials are added through LDS in lane order.
__global__ __launch_bounds__(256) void sa_fold_bwd_centre_kernel(const float* __restrict__ dZ, const float* __restrict__ u, const float* __restrict__ w,
                                                                 const int* __restrict__ idx, int N, int S, int ns, int C, long ncentres,
                                                                 const float* __restrict__ bn, const float* __restrict__ m1v,
                                                                 const float* __restrict__ m2v, float* __restrict__ dw) {
    __shared__ float shd[256 * 4];
    const int tid = threadIdx.x, tpr = C >> 2, nel = 256 / tpr;
    const int cq = tid % tpr, el = tid / tpr, c = 4 * cq;
    const f32x4 sc = *(const f32x4*)(bn + c), sh = *(const f32x4*)(bn + C + c), mu = *(const f32x4*)(bn + 2 * C + c), is = *(const f32x4*)(bn + 3 * C + c);
    f32x4 m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0};
    if (m1v) { m1 = *(const f32x4*)(m1v + c); m2 = *(const f32x4*)(m2v + c); }
    const long ctr = blockIdx.x;
    if (ctr >= ncentres) return;
    const long cloud = ctr / S;
    const f32x4 wq = *(const f32x4*)(w + (size_t)ctr * C + c);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = el; s < ns; s += nel) {
        const long e = ctr * ns + s;
        const int j = idx[e];
        const f32x4 uq = *(const f32x4*)(u + ((size_t)cloud * N + j) * C + c);
        const f32x4 dz = *(const f32x4*)(dZ + (size_t)e * C + c);
        acc = acc + saf_dy(uq, wq, dz, sc, sh, mu, is, m1, m2);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) shd[tid * 4 + q] = acc[q];
    __syncthreads();
    if (tid < tpr) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < nel; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] += shd[(g * tpr + tid) * 4 + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = -s[q];
        *(f32x4*)(dw + (size_t)ctr * C + 4 * tid) = s;
    }
}

// du_j = sum over the edges (i, s) with idx[i][s] == j, in the order of the reverse index.  One wave per point, lane = channel quad
// group: 64 lanes = (C/4 quads) x (64 / (C/4) entry lanes); entry lanes are combined with xor-shuffles in a fixed order.
__global__ __launch_bounds__(256) void sa_fold_bwd_point_kernel(const float* __restrict__ dZ, const float* __restrict__ u, const float* __restrict__ w,
                                                                const int* __restrict__ rev_off, const int* __restrict__ rev_ent, int N, int S,
                                                                int ns, int C, long P, const float* __restrict__ bn, const float* __restrict__ m1v,
                                                                const float* __restrict__ m2v, float* __restrict__ du,
                                                                const int* __restrict__ rev_cnt, const int* __restrict__ pad_cnt) {
    const int lane = threadIdx.x & 63;
    const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= P) return;
    const int tpr = C >> 2;                      // 16 (C = 64): 4 entry lanes; 32 (C = 128): 2; 64 (C = 256): 1
    const int nen = 64 / tpr;
    const int cq = lane % tpr, en = lane / tpr, c = 4 * cq;
    const long b = j / N;
    const f32x4 sc = *(const f32x4*)(bn + c), sh = *(const f32x4*)(bn + C + c), mu = *(const f32x4*)(bn + 2 * C + c), is = *(const f32x4*)(bn + 3 * C + c);
    f32x4 m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0};
    if (m1v) { m1 = *(const f32x4*)(m1v + c); m2 = *(const f32x4*)(m2v + c); }
    const f32x4 uq = *(const f32x4*)(u + (size_t)j * C + c);
    // compact lists (rev_cnt / pad_cnt, mlsp_group_reverse_compact): the padding slots of a ball-query group are not listed; their
    // rows are identical (same u_j, same w_i, same incoming gradient), so a group's slot-0 entry also adds pad_cnt times the row of its
    // LAST slot -- without this a point that pads many groups collects hundreds of entries
    const int e0 = rev_off[j], e1 = rev_cnt ? e0 + rev_cnt[j] : rev_off[j + 1];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (pad_cnt) {
        for (int t = e0 + en; t < e1; t += nen) {
            const int ent = rev_ent[t];
            const size_t ctr = (size_t)b * S + (ent >> 8);
            const int slot = ent & 255;
            const size_t row = ctr * ns + slot;
            const f32x4 wq = *(const f32x4*)(w + ctr * C + c);
            const f32x4 dz = *(const f32x4*)(dZ + row * C + c);
            acc = acc + saf_dy(uq, wq, dz, sc, sh, mu, is, m1, m2);
            const int pc = slot == 0 ? pad_cnt[ctr] : 0;
            if (pc > 0) {
                const f32x4 dzp = *(const f32x4*)(dZ + (ctr * ns + ns - 1) * C + c);
                const f32x4 d = saf_dy(uq, wq, dzp, sc, sh, mu, is, m1, m2);
                const float fp = (float)pc;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(fp, d[q], acc[q]);
            }
        }
    } else
    for (int t = e0 + en; t < e1; t += 2 * nen) {                          // two entries per lane in flight
        const int t2 = t + nen;
        const int ent = rev_ent[t], ent2 = t2 < e1 ? rev_ent[t2] : ent;
        const size_t ctr = (size_t)b * S + (ent >> 8), ctr2 = (size_t)b * S + (ent2 >> 8);
        const size_t row = ctr * ns + (ent & 255), row2 = ctr2 * ns + (ent2 & 255);
        const f32x4 dz = *(const f32x4*)(dZ + row * C + c), wq = *(const f32x4*)(w + ctr * C + c);
        const f32x4 dzb = *(const f32x4*)(dZ + row2 * C + c), wqb = *(const f32x4*)(w + ctr2 * C + c);
        acc = acc + saf_dy(uq, wq, dz, sc, sh, mu, is, m1, m2);
        if (t2 < e1) acc = acc + saf_dy(uq, wqb, dzb, sc, sh, mu, is, m1, m2);
    }
    for (int o = tpr; o < 64; o <<= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += __shfl_xor(acc[q], o, 64);
    }
    if (en == 0) *(f32x4*)(du + (size_t)j * C + c) = acc;
}

int launch_bn_finalize(hipStream_t st, const double* part, int nparts, double count, int C, const float* gamma, const float* beta,
                       float* run_mean, float* run_var, float momentum, float eps, float* scale, float* shift, float* save_mean,
                       float* save_invstd);
int launch_bn_eval_prepare(hipStream_t st, int C, const float* gamma, const float* beta, const float* run_mean, const float* run_var,
                           float eps, float* scale, float* shift, float* save_mean, float* save_invstd);
int launch_bn_bwd_finalize(hipStream_t st, const double* part, int nparts, double count, int C, float* dgamma, float* dbeta, float* mean_dz,
                           float* mean_dzy);

static bool saf_shape_ok(int C, int ns) { return (C == 16 || C == 32 || C == 64 || C == 128 || C == 256) && ns <= 256; }
int sa_fold_parts(long E) { return (int)((E + SAF_EPB - 1) / SAF_EPB); }

int launch_sa_fold_fwd(hipStream_t st, const float* u, const float* w, const int* idx, int B, int N, int S, int ns, int C, const float* gamma,
                       const float* beta, float* run_mean, float* run_var, float momentum, float eps, int training, float* Z, float* bn_save,
                       double* part) {
    if (!u || !w || !idx || !gamma || !beta || !Z || !bn_save || B <= 0 || N <= 0 || S <= 0 || ns <= 0) return MLSP_ERR_ARG;
    if (!saf_shape_ok(C, ns)) return MLSP_ERR_UNSUPPORTED;
    const long E = (long)B * S * ns;
    const int nparts = sa_fold_parts(E);
    float* scale = bn_save, *shift = bn_save + C, *mean = bn_save + 2 * C, *invstd = bn_save + 3 * C;
    if (training) {
        if (!part) return MLSP_ERR_WORKSPACE;
        hipLaunchKernelGGL((sa_fold_edge_kernel<0>), dim3(nparts), dim3(256), 0, st, u, w, idx, N, S, ns, C, E, (const float*)nullptr,
                           (const float*)nullptr, (float*)nullptr, part);
        int rc = launch_bn_finalize(st, part, nparts, (double)E, C, gamma, beta, run_mean, run_var, momentum, eps, scale, shift, mean, invstd);
        if (rc != MLSP_OK) return rc;
    } else {
        if (!run_mean || !run_var) return MLSP_ERR_ARG;
        int rc = launch_bn_eval_prepare(st, C, gamma, beta, run_mean, run_var, eps, scale, shift, mean, invstd);
        if (rc != MLSP_OK) return rc;
    }
    hipLaunchKernelGGL((sa_fold_edge_kernel<1>), dim3(nparts), dim3(256), 0, st, u, w, idx, N, S, ns, C, E, (const float*)bn_save,
                       (const float*)nullptr, Z, (double*)nullptr);
    return mlsp_launch_status();
}

int launch_sa_fold_bwd(hipStream_t st, const float* dZ, const float* u, const float* w, const int* idx, const int* rev_off, const int* rev_ent,
                       int B, int N, int S, int ns, int C, const float* bn_save, int training, float* du, float* dw, float* dgamma,
                       float* dbeta, double* part, float* mean_dz, float* mean_dzy, const int* rev_cnt, const int* pad_cnt) {
    if ((rev_cnt == nullptr) != (pad_cnt == nullptr)) return MLSP_ERR_ARG;
    if (!dZ || !u || !w || !idx || !rev_off || !rev_ent || !bn_save || !du || !dw || !dgamma || !dbeta || !part || !mean_dz || !mean_dzy)
        return MLSP_ERR_ARG;
    if (!saf_shape_ok(C, ns)) return MLSP_ERR_UNSUPPORTED;
    const long E = (long)B * S * ns, P = (long)B * N, NC = (long)B * S;
    const int nparts = sa_fold_parts(E);
    hipLaunchKernelGGL((sa_fold_edge_kernel<2>), dim3(nparts), dim3(256), 0, st, u, w, idx, N, S, ns, C, E, bn_save, dZ, (float*)nullptr, part);
    int rc = launch_bn_bwd_finalize(st, part, nparts, (double)E, C, dgamma, dbeta, mean_dz, mean_dzy);
    if (rc != MLSP_OK) return rc;
    const float* m1 = training ? mean_dz : nullptr;
    hipLaunchKernelGGL(sa_fold_bwd_centre_kernel, dim3((unsigned)NC), dim3(256), 0, st, dZ, u, w, idx, N, S, ns, C, NC, bn_save, m1,
                       (const float*)mean_dzy, dw);
    hipLaunchKernelGGL(sa_fold_bwd_point_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, st, dZ, u, w, rev_off, rev_ent, N, S, ns, C, P,
                       bn_save, m1, (const float*)mean_dzy, du, rev_cnt, pad_cnt);
    return mlsp_launch_status();
}
