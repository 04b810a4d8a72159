"""Round 6: EdgeConv conv4 (P = 32768, 128 -> 256, k = 20) forward + backward in every product mode against a float64 evaluation of the
same layer (torch ops on the GPU), at several magnitudes of the incoming gradient: relative L2 error of out / dx / dW per mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from mlsp_amd import functional as Fh

dev = torch.device("cuda:0")
B, N, C, Cout, k = int(os.environ.get("PB", 32)), 1024, 128, 256, 20
P = B * N
g = torch.Generator().manual_seed(5)
UNIFORM = os.environ.get("UNIFORM", "").split(",")          # which of x, W, gamma, beta, d are uniform in [-1, 1) instead of normal


def rnd(name, *shape):
    return torch.rand(*shape, generator=g) * 2 - 1 if (name in UNIFORM or "all" in UNIFORM) else torch.randn(*shape, generator=g)


xp = rnd("x", P, C).to(dev)
W = (rnd("W", Cout, 2 * C) * 0.1).to(dev)
gamma, beta = (rnd("gamma", Cout) + 0.3).to(dev), rnd("beta", Cout).to(dev)
d0 = rnd("d", P, Cout).to(dev)
print("uniform:", UNIFORM, flush=True)
graph = Fh.knn_graph(xp, B, N, k)
idx = graph.idx.view(B, N, k).long()


def ref64(dOut):
    x = xp.double().requires_grad_(True); Wd = W.double().requires_grad_(True)
    xb = x.view(B, N, C)
    nb = torch.gather(xb.unsqueeze(1).expand(B, N, N, C), 2, idx.unsqueeze(-1).expand(B, N, k, C))      # [B,N,k,C]
    f = torch.cat((nb - xb.unsqueeze(2), xb.unsqueeze(2).expand(B, N, k, C)), dim=-1)                  # reference graph feature
    y = f @ Wd.t()                                                                                      # [B,N,k,Cout]
    mu = y.mean(dim=(0, 1, 2)); var = y.var(dim=(0, 1, 2), unbiased=False)
    y = (y - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
    o = F.leaky_relu(y, 0.2).max(dim=2)[0].reshape(P, Cout)
    o.backward(dOut.double())
    return o.detach(), x.grad, Wd.grad


for gs in [float(v) for v in os.environ.get("GS", "1e-6,1,3e4").split(",")]:
    dOut = d0 * gs
    want = ref64(dOut)
    for mode in ("fp32", "bf16x6", "f16x3"):
        xg, Wg, gg, bg = [t.clone().requires_grad_(True) for t in (xp, W, gamma, beta)]
        rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
        with Fh.gemm_precision(mode):
            out = Fh.edgeconv(xg, graph, Wg, gg, bg, rm, rv, True)
            out.backward(dOut)
        got = (out.detach(), xg.grad, Wg.grad)
        errs = [((a.double() - b).norm() / b.norm()).item() for a, b in zip(got, want)]
        print("gscale %g  %-7s out %.2e  dx %.2e  dW %.2e" % (gs, mode, *errs), flush=True)
