"""Forward / backward time of the fused T-Net per-edge stage at the bench shape (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlsp_amd import functional as Fh
dev = torch.device("cuda:0")
B, N, k = int(os.environ.get("TN_B", 32)), int(os.environ.get("TN_N", 1024)), int(os.environ.get("TN_K", 20))
x = (torch.rand(B * N, 3, device=dev) * 2 - 1).requires_grad_(True)
g = Fh.knn_graph(x, B, N, k)
W1 = torch.randn(64, 6, device=dev, requires_grad=True); W2 = (torch.randn(128, 64, device=dev) / 8).requires_grad_(True)
g1, b1 = torch.ones(64, device=dev, requires_grad=True), torch.zeros(64, device=dev, requires_grad=True)
g2, b2 = torch.ones(128, device=dev, requires_grad=True), torch.zeros(128, device=dev, requires_grad=True)
rm1, rv1, rm2, rv2 = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros(128, device=dev), torch.ones(128, device=dev)
def fwd():
    return Fh.tnet_edge(x, g, W1, g1, b1, rm1, rv1, W2, g2, b2, rm2, rv2, True)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tf = timeit(lambda: fwd())
out = fwd(); w = torch.randn_like(out)
tb = timeit(lambda: torch.autograd.grad((fwd() * w).sum(), [x, W1, W2])) - tf
print("tnet_edge B=%d N=%d k=%d: forward op %.1f us, backward op %.1f us" % (B, N, k, tf, tb))
