// Times mlsp_gemm_f32 of the BUILT library (mlsp_amd/libmlsp_hip.so) per shape in GEMM precision modes 0 (fp32 MFMA), 2 (bf16x6 split) and 1 (bf16 operands).
//   hipcc -O2 -o tools/x6/lib_bench tools/x6/lib_bench.cpp -ldl        (run from the repo root; MLSP_HIP_LIB overrides the library path)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef int (*gemm_fn)(int, int, int, int, int, const float*, int, const float*, int, float*, int, const float*, int, void*, size_t, void*);   // ABI v8: precision is a per-call argument
typedef size_t (*ws_fn)(int, int, int);
struct Shape { const char* name; int ta, tb, M, N, K; };
int main(int argc, char** argv) {
    const char* path = getenv("MLSP_HIP_LIB") ? getenv("MLSP_HIP_LIB") : "mlsp_amd/libmlsp_hip.so";
    void* h = dlopen(path, RTLD_NOW);
    if (!h) { printf("dlopen %s: %s\n", path, dlerror()); return 1; }
    gemm_fn gemm = (gemm_fn)dlsym(h, "mlsp_gemm_f32");
    ws_fn wsb = (ws_fn)dlsym(h, "mlsp_workspace_bytes");
    const Shape shapes[] = {
        {"conv5 fwd   NT", 0, 1, 32768, 1024, 512}, {"conv5 dgrad NN", 0, 0, 32768, 512, 1024}, {"conv5 wgrad TN", 1, 0, 1024, 512, 32768},
        {"heads fwd   NT", 0, 1, 32768, 1024, 128}, {"head2 fwd   NT", 0, 1, 32768, 256, 256}, {"head2 dgrad NN", 0, 0, 32768, 256, 256},
        {"head2 wgrad TN", 1, 0, 256, 256, 32768}, {"dens1 fwd   NT", 0, 1, 32768, 512, 512}, {"edge4 uv    NT", 0, 1, 32768, 512, 128},
        {"edge4 wgrad TN", 1, 0, 512, 128, 32768}, {"big         NT", 0, 1, 8192, 8192, 1024},
        {"sa 262k fwd NT", 0, 1, 262144, 128, 128}, {"sa 262k dgr NN", 0, 0, 262144, 128, 128}, {"sa 524k fwd NT", 0, 1, 524288, 128, 64},
        {"sa 262k f2  NT", 0, 1, 262144, 256, 128}, {"edge uv K64 NT", 0, 1, 32768, 256, 64}, {"edge K128   NN", 0, 0, 32768, 128, 128}, {"heads K128  NT", 0, 1, 32768, 1024, 128},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        if (argc > 1 && !strstr(s.name, argv[1])) continue;
        const size_t na = (size_t)s.M * s.K, nb = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
        std::vector<float> A(na), B(nb);
        uint32_t st = 1u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
        const bool zero = getenv("ZERO") != nullptr;     // zero operands: same instruction stream, far fewer toggling bits -> is the kernel power-limited?
        for (auto& v : A) v = zero ? 0.f : rnd();
        for (auto& v : B) v = zero ? 0.f : rnd();
        float *dA, *dB, *dC; void* ws;
        const size_t wsz = wsb(s.M, s.K, s.N) + (64u << 20);
        CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dB, nb * 4)); CK(hipMalloc(&dC, nc * 4)); CK(hipMalloc(&ws, wsz));
        CK(hipMemcpy(dA, A.data(), na * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), nb * 4, hipMemcpyHostToDevice));
        double us[4] = {0, 0, 0, 0};
        const int reps = argc > 2 ? atoi(argv[2]) : 20;
        for (int mode = 0; mode <= 3; ++mode) {
            const int lda = s.ta ? s.M : s.K, ldb = s.tb ? s.K : s.N;
            for (int i = 0; i < 3; ++i) { int rc = gemm(s.ta, s.tb, s.M, s.N, s.K, dA, lda, dB, ldb, dC, s.N, nullptr, mode, ws, wsz, nullptr); if (rc) { printf("rc %d\n", rc); return 1; } }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) gemm(s.ta, s.tb, s.M, s.N, s.K, dA, lda, dB, ldb, dC, s.N, nullptr, mode, ws, wsz, nullptr);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            us[mode] = ms * 1000.0 / reps;
        }
        printf("%s M=%6d N=%5d K=%6d   fp32 %7.1f us %6.1f TF | bf16x6 %7.1f us %6.1f TF   %.2fx | f16x3 %7.1f us %6.1f TF  %.2fx of bf16x6 | bf16 operands %7.1f us %6.1f TF\n", s.name, s.M, s.N, s.K, us[0],
               2.0 * s.M * s.N * s.K / us[0] * 1e-6, us[2], 2.0 * s.M * s.N * s.K / us[2] * 1e-6, us[0] / us[2], us[3], 2.0 * s.M * s.N * s.K / us[3] * 1e-6, us[2] / us[3], us[1], 2.0 * s.M * s.N * s.K / us[1] * 1e-6);
        fflush(stdout);
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(ws));
    }
    return 0;
}
