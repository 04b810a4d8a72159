"""Per-shape timing of the fp32 MFMA GEMM (run on the GPU box): TF/s for the shapes of one DGCNN+MLSP step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlsp_amd import functional as Fh

P, E = 32768, 655360
shapes = [  # (name, ta, tb, M, N, K)
    ("conv5 fwd", 0, 1, P, 1024, 512), ("conv5 dgrad", 0, 0, P, 512, 1024), ("conv5 wgrad", 1, 0, 1024, 512, P),
    ("head1 fwd", 0, 1, P, 256, 512), ("head2 fwd", 0, 1, P, 256, 256), ("head3 fwd", 0, 1, P, 128, 256),
    ("dens1 fwd", 0, 1, P, 512, 512), ("head2 dgrad", 0, 0, P, 256, 256), ("head2 wgrad", 1, 0, 256, 256, P),
    ("edge4 uv", 0, 1, P, 512, 128), ("edge1 uv", 0, 1, P, 128, 3), ("edge4 wgrad", 1, 0, 512, 128, P),
    ("tnet c1 fwd", 0, 1, E, 64, 6), ("tnet c2 fwd", 0, 1, E, 128, 64), ("tnet c2 dgrad", 0, 0, E, 64, 128),
    ("tnet c2 wgrad", 1, 0, 128, 64, E), ("tnet c1 wgrad", 1, 0, 64, 6, E), ("tnet c3 fwd", 0, 1, P, 1024, 128),
    ("fc 32x1024x512", 0, 1, 32, 512, 1024), ("big 8192^2 x 1024", 0, 1, 8192, 8192, 1024),
]
dev = torch.device("cuda:0")
Fh.gemm_precision.set(os.environ.get("GEMM_PRECISION", "fp32"))          # fp32 | bf16 | bf16x6
for name, ta, tb, M, N, K in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    for _ in range(3):
        Fh.gemm(A, B, ta=bool(ta), tb=bool(tb))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        Fh.gemm(A, B, ta=bool(ta), tb=bool(tb))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # library reference point (rocBLAS / hipBLASLt through torch.mm), same operand layouts
    lib_ms = float("nan")
    if "--lib" in sys.argv:
        a = A.t() if ta else A
        b = B.t() if tb else B
        for _ in range(3):
            torch.mm(a, b)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            torch.mm(a, b)
        e1.record()
        torch.cuda.synchronize()
        lib_ms = e0.elapsed_time(e1) / reps
    print("%-20s M=%7d N=%5d K=%7d  %8.1f us  %6.1f TF/s   torch.mm %8.1f us %6.1f TF/s" %
          (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, lib_ms * 1e3, 2.0 * M * N * K / lib_ms / 1e9))
