"""Round 6: accuracy of the two-piece f16 split (gemm_precision("f16x3")) against float64, beside the f32-MFMA and bf16x6 kernels.
    python tools/r6/f16x3_probe.py [scale]     (scale multiplies both operands: where in the f16 range the data sits while the library
                                                has no amax-driven scaling yet)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mlsp_amd import functional as Fh  # noqa: E402

dev = torch.device("cuda:0")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
wide = "wide" in sys.argv
shapes = [(False, True, 4096, 256, 256), (False, True, 32768, 1024, 512), (False, False, 32768, 512, 1024), (True, False, 1024, 512, 32768),
          (False, True, 8192, 256, 128), (True, False, 256, 256, 32768)]
for ta, tb, M, N, K in shapes:
    g = torch.Generator().manual_seed(11)
    shpA, shpB = ((K, M) if ta else (M, K)), ((N, K) if tb else (K, N))
    if wide:
        A = (torch.randn(shpA, generator=g) * torch.exp(2.0 * torch.randn(shpA, generator=g))).to(dev) * scale
        B = (torch.randn(shpB, generator=g) * torch.exp(2.0 * torch.randn(shpB, generator=g))).to(dev) * scale
    else:
        A = torch.randn(shpA, generator=g).to(dev) * scale
        B = torch.randn(shpB, generator=g).to(dev) * scale
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    out = {}
    for mode in ("fp32", "bf16x6", "f16x3"):
        with Fh.gemm_precision(mode):
            c = Fh.gemm(A, B, ta, tb).double()
        out[mode] = (((c - ref).norm() / ref.norm()).item(), ((c - ref).abs().max() / ref.abs().max()).item())
    print("ta=%d tb=%d M=%d N=%d K=%d amaxA=%.3g amaxB=%.3g | rel-L2 fp32 %.2e  bf16x6 %.2e  f16x3 %.2e (%.2fx fp32) | max-abs/max %.2e %.2e %.2e"
          % (ta, tb, M, N, K, A.abs().max().item(), B.abs().max().item(), out["fp32"][0], out["bf16x6"][0], out["f16x3"][0],
             out["f16x3"][0] / out["fp32"][0], out["fp32"][1], out["bf16x6"][1], out["f16x3"][1]))
