import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mlsp_amd import functional as Fh
dev = torch.device("cuda:0")
P = 32768
for (ta, tb, M, N, K) in [(0, 0, P, 64, 256), (0, 0, P, 64, 128), (1, 0, 256, 64, P), (1, 0, 128, 64, P), (0, 1, P, 64, 128)]:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    got = Fh.gemm(A, B, ta=bool(ta), tb=bool(tb))
    want = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
    err = (got.double() - want).abs().max().item()
    for _ in range(3): Fh.gemm(A, B, ta=bool(ta), tb=bool(tb))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): Fh.gemm(A, B, ta=bool(ta), tb=bool(tb))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("%s%s M=%d N=%d K=%d: %.1f us %.1f TF  max err %.2e" % ("T" if ta else "N", "T" if tb else "N", M, N, K, us, 2.0 * M * N * K / us / 1e6, err))
