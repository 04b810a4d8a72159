// Label generators of the MLSP target branch, on device.  In the reference these are
// per-cloud CPU loops through python-pcl inside the training step:
//   MLSP/mlsp.py:240-272   cal_density: KD-tree radius search -> neighbour count -> soft 16-bin label
//   PointDA/trainer.py:173-188  kSearchNormalEstimation: PCL NormalEstimation with KSearch(k)
// python-pcl is a third-party, unpinned dependency absent from the reference tree, so PARITY IS UNPINNED: the
// semantics below restate the published PCL/FLANN algorithms (radius search: squared L2 < r^2, at most max_nn
// results; normal = eigenvector of the smallest eigenvalue of the k-neighbourhood covariance, flipped towards
// the viewpoint (0,0,0)); tests pin them to a numpy restatement and to closed-form shapes.
#include "common.h"
#include <math.h>

// count[i] = min(#{j : |x_i - x_j|^2 < r2}, max_nn) - [point 0 of the cloud is within range]   (mlsp.py:254: `ind != 0`)
__global__ __launch_bounds__(256) void radius_count_kernel(const float* __restrict__ x, int ld, int N, float r2, int max_nn,
                                                           int* __restrict__ count) {
    extern __shared__ float pts[];            // [N][3]
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * N * ld;
    for (int e = threadIdx.x; e < N; e += blockDim.x) {
        pts[3 * e + 0] = xb[(size_t)e * ld + 0]; pts[3 * e + 1] = xb[(size_t)e * ld + 1]; pts[3 * e + 2] = xb[(size_t)e * ld + 2];
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    int c = 0;
    for (int j = 0; j < N; ++j) {
        const float dx = px - pts[3 * j], dy = py - pts[3 * j + 1], dz = pz - pts[3 * j + 2];
        const float d2 = dx * dx + dy * dy + dz * dz;
        c += d2 < r2 ? 1 : 0;
    }
    const float d0x = px - pts[0], d0y = py - pts[1], d0z = pz - pts[2];
    const bool zero_in = d0x * d0x + d0y * d0y + d0z * d0z < r2;
    c = min(c, max_nn) - (zero_in ? 1 : 0);
    count[(size_t)b * N + i] = max(c, 0);
}

// smallest-eigenvalue eigenvector of a symmetric 3x3 (closed form: trigonometric eigenvalues, cross-product vector)
__device__ __forceinline__ void smallest_eigvec3(float a00, float a01, float a02, float a11, float a12, float a22, float* n) {
    // scale for robustness
    float sc = fmaxf(fmaxf(fabsf(a00), fabsf(a11)), fmaxf(fabsf(a22), fmaxf(fabsf(a01), fmaxf(fabsf(a02), fabsf(a12)))));
    if (!(sc > 0.f)) { n[0] = 0.f; n[1] = 0.f; n[2] = 1.f; return; }
    const double s = 1.0 / sc;
    const double b00 = a00 * s, b01 = a01 * s, b02 = a02 * s, b11 = a11 * s, b12 = a12 * s, b22 = a22 * s;
    const double q = (b00 + b11 + b22) / 3.0;
    const double p1 = b01 * b01 + b02 * b02 + b12 * b12;
    const double c00 = b00 - q, c11 = b11 - q, c22 = b22 - q;
    const double p2 = c00 * c00 + c11 * c11 + c22 * c22 + 2.0 * p1;
    double lam;                                    // smallest eigenvalue
    if (p2 <= 0.0) lam = q;
    else {
        const double p = sqrt(p2 / 6.0);
        const double d00 = c00 / p, d11 = c11 / p, d22 = c22 / p, d01 = b01 / p, d02 = b02 / p, d12 = b12 / p;
        double r = 0.5 * (d00 * (d11 * d22 - d12 * d12) - d01 * (d01 * d22 - d12 * d02) + d02 * (d01 * d12 - d11 * d02));
        r = fmin(1.0, fmax(-1.0, r));
        const double phi = acos(r) / 3.0;
        lam = q + 2.0 * p * cos(phi + 2.0943951023931953);      // + 2*pi/3 -> the smallest root
    }
    // rows of (B - lam I); the eigenvector is orthogonal to all of them: take the largest cross product
    const double r0x = b00 - lam, r0y = b01, r0z = b02;
    const double r1x = b01, r1y = b11 - lam, r1z = b12;
    const double r2x = b02, r2y = b12, r2z = b22 - lam;
    double v0x = r0y * r1z - r0z * r1y, v0y = r0z * r1x - r0x * r1z, v0z = r0x * r1y - r0y * r1x;
    double v1x = r0y * r2z - r0z * r2y, v1y = r0z * r2x - r0x * r2z, v1z = r0x * r2y - r0y * r2x;
    double v2x = r1y * r2z - r1z * r2y, v2y = r1z * r2x - r1x * r2z, v2z = r1x * r2y - r1y * r2x;
    double n0 = v0x * v0x + v0y * v0y + v0z * v0z, n1 = v1x * v1x + v1y * v1y + v1z * v1z, n2 = v2x * v2x + v2y * v2y + v2z * v2z;
    double bx = v0x, by = v0y, bz = v0z, bn = n0;
    if (n1 > bn) { bx = v1x; by = v1y; bz = v1z; bn = n1; }
    if (n2 > bn) { bx = v2x; by = v2y; bz = v2z; bn = n2; }
    if (!(bn > 0.0)) { n[0] = 0.f; n[1] = 0.f; n[2] = 1.f; return; }
    const double inv = 1.0 / sqrt(bn);
    n[0] = (float)(bx * inv); n[1] = (float)(by * inv); n[2] = (float)(bz * inv);
}

// normals[i] from the k nearest neighbours idx[i][0..k) (self included, as PCL's KSearch returns it)
__global__ __launch_bounds__(256) void knn_normals_kernel(const float* __restrict__ x, int ld, const int* __restrict__ idx, int P, int N,
                                                          int k, float* __restrict__ normals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int base = (i / N) * N;
    const int* row = idx + (size_t)i * k;
    double mx = 0, my = 0, mz = 0;
    for (int s = 0; s < k; ++s) {
        const float* q = x + (size_t)(base + row[s]) * ld;
        mx += q[0]; my += q[1]; mz += q[2];
    }
    mx /= k; my /= k; mz /= k;
    double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    for (int s = 0; s < k; ++s) {
        const float* q = x + (size_t)(base + row[s]) * ld;
        const double dx = q[0] - mx, dy = q[1] - my, dz = q[2] - mz;
        c00 += dx * dx; c01 += dx * dy; c02 += dx * dz; c11 += dy * dy; c12 += dy * dz; c22 += dz * dz;
    }
    float n[3];
    smallest_eigvec3((float)(c00 / k), (float)(c01 / k), (float)(c02 / k), (float)(c11 / k), (float)(c12 / k), (float)(c22 / k), n);
    // flipNormalTowardsViewpoint with vp = (0,0,0):  n . (vp - p) must be >= 0
    const float* p = x + (size_t)i * ld;
    if (n[0] * (-p[0]) + n[1] * (-p[1]) + n[2] * (-p[2]) < 0.f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    normals[(size_t)i * 3 + 0] = n[0]; normals[(size_t)i * 3 + 1] = n[1]; normals[(size_t)i * 3 + 2] = n[2];
}

int launch_radius_count(hipStream_t st, const float* x, int ld, int B, int N, float radius, int max_nn, int* count) {
    size_t lds = (size_t)3 * N * sizeof(float);
    if (lds > 150 * 1024) return MLSP_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = mlsp_lds_limit((const void*)radius_count_kernel, lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(radius_count_kernel, dim3((N + 255) / 256, B), dim3(256), lds, st, x, ld, N, radius * radius, max_nn, count);
    return mlsp_launch_status();
}
int launch_knn_normals(hipStream_t st, const float* x, int ld, const int* idx, int B, int N, int k, float* normals) {
    int P = B * N;
    hipLaunchKernelGGL(knn_normals_kernel, dim3((P + 255) / 256), dim3(256), 0, st, x, ld, idx, P, N, k, normals);
    return mlsp_launch_status();
}
