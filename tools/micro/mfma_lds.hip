// Micro-benchmark: the GEMM inner loop in isolation (LDS fragment reads + v_mfma_f32_32x32x2f32), no global traffic.
// Reports TF/s and the shader clock actually held during the kernel (clock64 = shader cycles, wall_clock64 = 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SROW 36

// TM x TN MFMA tiles per wave; 4 waves per block; operands in LDS as [rows][36] (row-major, 32 k per tile)
template <int TM, int TN>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int ktiles, long long* clk) {
    __shared__ __attribute__((aligned(16))) float As[64 * TM * SROW], Bs[64 * TN * SROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    for (int i = tid; i < 64 * TM * SROW; i += 256) As[i] = (float)(i % 7) * 0.25f;
    for (int i = tid; i < 64 * TN * SROW; i += 256) Bs[i] = (float)(i % 5) * 0.5f;
    __syncthreads();
    long long c0 = clock64(), w0 = wall_clock64();
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm = wave >> 1, wn = wave & 1;
    const float* Ab = As + (wm * 32 * TM + l31) * SROW + 2 * h;
    const float* Bb = Bs + (wn * 32 * TN + l31) * SROW + 2 * h;
    for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            float2 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const float2*)(Ab + i * 32 * SROW + 4 * m);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const float2*)(Bb + j * 32 * SROW + 4 * m);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
        }
#ifdef WITH_BARRIER
        __syncthreads();
#endif
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int TM, int TN>
void run(int blocks_per_cu, int ktiles) {
    float* out; long long* clk; long long hc[2];
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((loop_kernel<TM, TN>), dim3(grid), dim3(256), 0, 0, out, ktiles, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL((loop_kernel<TM, TN>), dim3(grid), dim3(256), 0, 0, out, ktiles, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)grid * 4 * ktiles * 16.0 * TM * TN * 4096.0;
    printf("wave tile %dx%d blocks/CU=%d: %.1f us  %.1f TF  shader clock %.0f MHz  (MFMA pipe %.0f%% of cycles)\n", 32 * TM, 32 * TN,
           blocks_per_cu, ms * 1e3, flops / (ms * 1e-3) / 1e12, (double)hc[0] / ((double)hc[1] / 100.0),
           100.0 * ktiles * 16.0 * TM * TN * 64.0 * blocks_per_cu / (double)hc[0]);
    hipFree(out); hipFree(clk);
}
int main() {
    run<2, 2>(1, 400); run<2, 2>(2, 400); run<2, 2>(3, 400);
    run<1, 2>(4, 400);
    run<4, 2>(1, 200); run<4, 2>(2, 200);
    run<4, 4>(1, 100);
    return 0;
}
