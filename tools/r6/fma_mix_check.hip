#include <hip/hip_runtime.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float res_lo(float x, float s, unsigned pk) { float r; asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "s"(s), "v"(pk)); return r; }
__device__ __forceinline__ float res_hi(float x, float s, unsigned pk) { float r; asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "s"(s), "v"(pk)); return r; }
__global__ void k(const float* x, unsigned* o, float* r, float s) {
    float x0 = x[threadIdx.x * 2], x1 = x[threadIdx.x * 2 + 1];
    f2 v = {x0, x1}; v = v * s;
    h2 a = __builtin_convertvector(v, h2);
    unsigned pk = __builtin_bit_cast(unsigned, a);
    float r0 = res_lo(x0, s, pk), r1 = res_hi(x1, s, pk);
    f2 w = {r0, r1};
    h2 b = __builtin_convertvector(w, h2);
    o[threadIdx.x * 2] = pk;
    o[threadIdx.x * 2 + 1] = __builtin_bit_cast(unsigned, b);
    r[threadIdx.x*2] = r0; r[threadIdx.x*2+1] = r1;
}
int main() {
    const int n = 64; float hx[2*n]; for (int i = 0; i < 2*n; ++i) hx[i] = (i % 7 - 3) * 0.37123f * (1 + i) + 1e-3f * i;
    float *dx, *dr; unsigned* dout; hipMalloc(&dx, sizeof(hx)); hipMalloc(&dout, 2*n*4); hipMalloc(&dr, 2*n*4);
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    k<<<1, n>>>(dx, dout, dr, 4.0f);
    unsigned ho[2*n]; float hr[2*n]; hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost); hipMemcpy(hr, dr, sizeof(hr), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        _Float16 a0 = ((_Float16*)&ho[2*i])[0], a1 = ((_Float16*)&ho[2*i])[1];
        float e0 = hx[2*i] * 4.0f - (float)a0, e1 = hx[2*i+1] * 4.0f - (float)a1;
        if (e0 != hr[2*i] || e1 != hr[2*i+1]) { if (bad < 5) printf("mismatch %d: %g %g | %g %g\n", i, e0, hr[2*i], e1, hr[2*i+1]); ++bad; }
    }
    printf("bad = %d\n", bad);
    return bad != 0;
}
