"""Timings of the thin GEMM shapes of one training step against their HBM streaming floor (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlsp_amd import functional as Fh

dev = torch.device("cuda:0")
P = 32768
for (ta, tb, M, N, K) in [(0, 1, P, 128, 3), (0, 0, P, 128, 3), (0, 0, P, 256, 16), (0, 1, P, 3, 128), (0, 1, P, 16, 256), (0, 0, P, 3, 128),
                          (1, 0, 3, 128, P), (1, 0, 16, 256, P), (1, 0, 128, 3, P)]:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    for _ in range(3):
        Fh.gemm(A, B, ta=bool(ta), tb=bool(tb))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        Fh.gemm(A, B, ta=bool(ta), tb=bool(tb))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = 4e-6 * (A.numel() + B.numel() + M * N)
    print("%s%s M=%d N=%d K=%d: %.1f us  (%.1f MB -> %.2f TB/s)" % ("T" if ta else "N", "T" if tb else "N", M, N, K, us, mb, mb / us))
