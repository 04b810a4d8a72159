"""Gradients of the fused T-Net per-edge stage from its three independent backward kernels: the dense split-product Gram form
(tnet_edge_bwds_kernel, the default), the f32 Gram form with the register-indexed sparse half (tnet_edge_bwdg_kernel,
MLSP_TNET_BWD_F32=1) and the round-1 kernel that forms the dZ tile (MLSP_TNET_BWD_OLD=1).  The switches are read once per process, so
each variant runs in its own process (GPU box).
Usage: python tools/cmp_tnet_bwd.py [B N k]  -> one line per gradient: max |variant - old| / max |old| for both Gram forms."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from mlsp_amd import functional as Fh
B, N, k = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dev = torch.device("cuda:0"); torch.manual_seed(0)
x = (torch.rand(B * N, 3, device=dev) * 2 - 1).requires_grad_(True)
g = Fh.knn_graph(x, B, N, k)
W1 = torch.randn(64, 6, device=dev, requires_grad=True); W2 = (torch.randn(128, 64, device=dev) / 8).requires_grad_(True)
g1, b1 = (torch.rand(64, device=dev) + 0.5).requires_grad_(True), torch.randn(64, device=dev, requires_grad=True)
g2, b2 = (torch.rand(128, device=dev) + 0.5).requires_grad_(True), torch.randn(128, device=dev, requires_grad=True)
rm1, rv1, rm2, rv2 = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros(128, device=dev), torch.ones(128, device=dev)
out = Fh.tnet_edge(x, g, W1, g1, b1, rm1, rv1, W2, g2, b2, rm2, rv2, True)
w = torch.randn_like(out)
gr = torch.autograd.grad((out * w).sum(), [x, W1, g1, b1, W2, g2, b2])
torch.save([t.cpu() for t in gr], sys.argv[2])
'''
NAMES = ["dx", "dW1", "dgamma1", "dbeta1", "dW2", "dgamma2", "dbeta2"]


def compare(B=32, N=1024, k=20):
    import torch
    with tempfile.TemporaryDirectory() as d:
        child = os.path.join(d, "child.py")
        open(child, "w").write(CHILD)
        outs = []
        for tag, extra in (("split", {}), ("f32", {"MLSP_TNET_BWD_F32": "1"}), ("old", {"MLSP_TNET_BWD_OLD": "1"})):
            f = os.path.join(d, tag + ".pt")
            subprocess.check_call([sys.executable, child, ROOT, f, str(B), str(N), str(k)], env=dict(os.environ, **extra))
            outs.append(torch.load(f))
    split, f32, old = outs
    res = {}
    for n, a, b, q in zip(NAMES, split, f32, old):
        res[n] = ((a - q).abs().max() / q.abs().max()).item()
        res[n + " (f32 Gram form)"] = ((b - q).abs().max() / q.abs().max()).item()
    return res


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:4]] or [32, 1024, 20]
    for n, e in compare(*a).items():
        print("%-8s rel max err %.3e" % (n, e))
