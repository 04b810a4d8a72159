"""Forward accuracy of the fused T-Net per-edge stage against a float64 evaluation of the reference's op sequence (training mode), for the
kernel the library picks (MLSP_TNET_FWD_F32=1: f32 MFMA products; MLSP_TNET_FWD_SPLIT=1 or >= 1024 tiles: split products on the bf16 cores).  Prints one line per
shape; `--check` exits non-zero when a relative L2 error exceeds 4e-7."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import torch.nn.functional as F
from mlsp_amd import functional as Fh
from oracle import knn_canon, ref_cpu

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
SHAPES = ((8, 1024, 20), (4, 2048, 40)) if "--small" not in sys.argv else ((4, 1024, 20), (4, 128, 20), (6, 256, 20), (2, 1024, 20), (8, 1024, 20))
for B, N, k in SHAPES:
    P = B * N
    xp = torch.rand(P, 3, generator=g) * 2 - 1
    gout = torch.randn(P, 128, generator=g)                 # upstream gradient for the backward comparison (--small)
    W1, W2 = torch.randn(64, 6, generator=g) * 0.5, torch.randn(128, 64, generator=g) * 0.2
    g1, b1 = torch.rand(64, generator=g) + 0.3, torch.randn(64, generator=g) * 0.1
    g2, b2 = torch.rand(128, generator=g) + 0.3, torch.randn(128, generator=g) * 0.1
    idx = torch.from_numpy(knn_canon.knn_point_major(xp.view(B, N, 3), k).astype(np.int64))
    d = lambda t: t.double()
    if "--small" in sys.argv:
        W1d, W2d = d(W1).requires_grad_(True), d(W2).requires_grad_(True)
    else:
        W1d, W2d = d(W1), d(W2)
    x = d(xp).view(B, N, 3).transpose(2, 1)
    f = ref_cpu.graph_feature(x, idx)
    y = torch.einsum("oc,bcnk->bonk", W1d, f)
    y = F.leaky_relu(F.batch_norm(y, None, None, d(g1), d(b1), True, 0.1, 1e-5), 0.2)
    z = torch.einsum("oc,bcnk->bonk", W2d, y)
    z = F.leaky_relu(F.batch_norm(z, None, None, d(g2), d(b2), True, 0.1, 1e-5), 0.2)
    want = z.max(dim=-1)[0].transpose(2, 1).reshape(P, 128)
    graph = Fh.graph_from_indices(idx.to(dev), B, N, k)
    rs = [torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros(128, device=dev), torch.ones(128, device=dev)]
    W1g, W2g = W1.to(dev).requires_grad_("--small" in sys.argv), W2.to(dev).requires_grad_("--small" in sys.argv)
    got = Fh.tnet_edge(xp.to(dev), graph, W1g, g1.to(dev), b1.to(dev), rs[0], rs[1], W2g, g2.to(dev), b2.to(dev), rs[2], rs[3], True)
    if "--small" in sys.argv:
        (want * d(gout)).sum().backward()
        (got * gout.to(dev)).sum().backward()
        for nm, a_, b_ in (("dW1", W1g.grad, W1d.grad), ("dW2", W2g.grad, W2d.grad)):
            print("   %s rel-L2 vs float64: %.3e" % (nm, ((a_.double().cpu() - b_).norm() / b_.norm()).item()))
        want = want.detach()
        got = got.detach()
    e = (got.double().cpu() - want)
    rel = (e.norm() / want.norm()).item()
    print("B=%d N=%d k=%d: rel-L2 %.3e  max-abs %.3e (scale %.2f)" % (B, N, k, rel, e.abs().max().item(), want.abs().max().item()))
    if "--check" in sys.argv and not rel < 4e-7:
        sys.exit(1)
