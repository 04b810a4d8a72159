// Stand-alone bench of the six-product bf16 split GEMM variants (development harness for mlsp_amd/csrc/gemm.hip's gemm_split_kernel).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/x6/x6_bench tools/x6/x6_bench.hip
// C[M][N] = sum_k A(m,k) B(n,k);  KA: A stored [K][M] (k-major) else [M][K];  KB: B stored [K][N] else [N][K].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
#include "x6_kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int ka, kb, M, N, K, nsplit; };

static double ref_entry(const std::vector<float>& A, const std::vector<float>& B, const Shape& s, int m, int n, int k0, int k1) {
    double acc = 0;
    for (int k = k0; k < k1; ++k) {
        const double a = s.ka ? A[(size_t)k * s.M + m] : A[(size_t)m * s.K + k];
        const double b = s.kb ? B[(size_t)k * s.N + n] : B[(size_t)n * s.K + k];
        acc += a * b;
    }
    return acc;
}

int main(int argc, char** argv) {
    const int variant_lo = argc > 1 ? atoi(argv[1]) : 0, variant_hi = argc > 2 ? atoi(argv[2]) : X6_NVARIANTS - 1;
    const Shape shapes[] = {
        {"conv5 fwd  ", 0, 0, 32768, 1024, 512, 1},
        {"conv5 dgrad", 0, 1, 32768, 512, 1024, 1},
        {"conv5 wgrad", 1, 1, 1024, 512, 32768, 16},
        {"head2 fwd  ", 0, 0, 32768, 256, 256, 1},
        {"head2 dgrad", 0, 1, 32768, 256, 256, 1},
        {"head2 wgrad", 1, 1, 256, 256, 32768, 128},
        {"tnet c3 fwd", 0, 0, 32768, 1024, 128, 1},
        {"big 8192   ", 0, 0, 8192, 8192, 1024, 1},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        const size_t na = (size_t)s.M * s.K, nb = (size_t)s.N * s.K, nc = (size_t)s.M * s.N * s.nsplit;
        std::vector<float> A(na), B(nb), C(nc);
        uint32_t st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : A) v = rnd() * (1.0f + 3.0f * rnd());
        for (auto& v : B) v = rnd();
        float *dA, *dB, *dC;
        CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dB, nb * 4)); CK(hipMalloc(&dC, nc * 4));
        CK(hipMemcpy(dA, A.data(), na * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), nb * 4, hipMemcpyHostToDevice));
        for (int var = variant_lo; var <= variant_hi; ++var) {
            X6Args p;
            p.A = dA; p.B = dB; p.C = dC; p.M = s.M; p.N = s.N; p.K = s.K;
            p.lda = s.ka ? s.M : s.K; p.ldb = s.kb ? s.N : s.K; p.ldc = s.N;
            p.nsplit = s.nsplit; p.ksplit = s.K / s.nsplit;
            CK(hipMemset(dC, 0, nc * 4));
            if (!x6_launch(var, s.ka, s.kb, p, 0)) continue;
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(C.data(), dC, nc * 4, hipMemcpyDeviceToHost));
            double worst = 0, scale = 0;
            uint32_t q = 777u;
            for (int t = 0; t < 200; ++t) {
                q = q * 1664525u + 1013904223u; const int m = (q >> 4) % s.M;
                q = q * 1664525u + 1013904223u; const int n = (q >> 4) % s.N;
                q = q * 1664525u + 1013904223u; const int sp = (q >> 4) % s.nsplit;
                const double want = ref_entry(A, B, s, m, n, sp * p.ksplit, (sp + 1) * p.ksplit);
                const double got = C[(size_t)sp * s.M * s.N + (size_t)m * s.N + n];
                worst = fmax(worst, fabs(got - want)); scale = fmax(scale, fabs(want));
            }
            for (int i = 0; i < 3; ++i) x6_launch(var, s.ka, s.kb, p, 0);
            CK(hipDeviceSynchronize());
            const int reps = 20;
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) x6_launch(var, s.ka, s.kb, p, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1000.0 / reps;
            printf("%s M=%6d N=%5d K=%6d  var %d %-28s %8.1f us %7.1f TF/s   max err %.2e (scale %.2e)\n", s.name, s.M, s.N, s.K, var,
                   x6_variant_name(var), us, 2.0 * s.M * s.N * s.K / us * 1e-6, worst, scale);
            fflush(stdout);
        }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    return 0;
}
