// Checks the lane-exchange helpers of knn6.hip's parallel final (values from lane ^ 16 and lane ^ 32 without the LDS crossbar):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/permlane_check tools/micro/permlane_check.hip && /tmp/permlane_check
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned x16(unsigned v, int lane) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return ((lane >> 4) & 1) ? r[0] : r[1];
}
__device__ __forceinline__ unsigned x32(unsigned v, int lane) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane >> 5) ? r[0] : r[1];
}
__global__ void k(unsigned* o) {
    const int lane = threadIdx.x;
    o[lane] = x16(1000u + lane, lane);
    o[64 + lane] = x32(2000u + lane, lane);
}
int main() {
    unsigned* d; hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) { if (h[l] != 1000u + (l ^ 16)) ++bad; if (h[64 + l] != 2000u + (l ^ 32)) ++bad; }
    printf("x16: lane 0 <- %u, lane 16 <- %u, lane 37 <- %u | x32: lane 5 <- %u | mismatches %d\n", h[0], h[16], h[37], h[64 + 5], bad);
    return bad != 0;
}
