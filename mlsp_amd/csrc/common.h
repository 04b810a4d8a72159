// Shared device/host helpers for libmlsp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define MLSP_OK 0
#define MLSP_ERR_ARG (-1)        // bad shape / null pointer / unsupported size
#define MLSP_ERR_WORKSPACE (-2)  // workspace too small
#define MLSP_ERR_UNSUPPORTED (-3)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

static inline int mlsp_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MLSP_OK : (int)e;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// simple bump allocator over the caller-provided workspace
struct Workspace {
    char* base;
    size_t size, off;
    Workspace(void* p, size_t n) : base((char*)p), size(n), off(0) {}
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

// counter-based dropout bit: keep iff hash(seed, i) >= p * 2^32   (lowbias32 mix)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t i, uint32_t thresh) {
    uint32_t h = mix32((uint32_t)i ^ mix32((uint32_t)(i >> 32) + (uint32_t)seed) ^ (uint32_t)(seed >> 32) * 0x9e3779b9U);
    return h >= thresh;
}
