// Which float division / sqrt / contraction does torch's fused Adam use on this build?  One step of the update under several
// lowerings; tools/r5/adam_probe/run.py counts the elements that differ from torch._fused_adam_.
#include <hip/hip_runtime.h>
#include <math.h>
extern "C" __global__ void probe(float* p, const float* g, float* m, float* v, int n, double lr, double b1, double b2, double wd, double eps,
                                 float bc1, float bc2s, int mode) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float param = p[i], grad = g[i], ea = m[i], es = v[i];
    if (mode & 8) {
        grad = (float)fma((double)param, wd, (double)grad);
        ea = (float)fma(b1, (double)ea, (1.0 - b1) * (double)grad);
        es = (float)fma(b2, (double)es, ((1.0 - b2) * (double)grad) * (double)grad);
    } else {
        grad = (float)((double)grad + (double)param * wd);
        ea = (float)(b1 * (double)ea + (1.0 - b1) * (double)grad);
        es = (float)(b2 * (double)es + ((1.0 - b2) * (double)grad) * (double)grad);
    }
    const float step_size = (float)(lr / (double)bc1);
    float sq = (mode & 4) ? __builtin_amdgcn_sqrtf(es) : sqrtf(es);
    float q = (mode & 1) ? sq * __builtin_amdgcn_rcpf(bc2s) : sq / bc2s;
    const float denom = (float)((double)q + eps);
    float num = step_size * ea;
    float upd = (mode & 2) ? num * __builtin_amdgcn_rcpf(denom) : num / denom;
    if (mode & 16) param = fmaf(-step_size, ea / denom, param); else param -= upd;
    p[i] = param; m[i] = ea; v[i] = es;
}
extern "C" int run_probe(float* p, const float* g, float* m, float* v, int n, double lr, double b1, double b2, double wd, double eps, int step, int mode) {
    const float bc1 = (float)(1.0 - pow(b1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow(b2, (double)step));
    hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, 0, p, g, m, v, n, lr, b1, b2, wd, eps, bc1, bc2s, mode);
    return (int)hipDeviceSynchronize();
}
