"""EdgeConv forward+backward alone on a clustered cloud (hubs in the kNN graph), for kernel traces: python tools/time_edge_bwd.py [Cout] [iters]"""
import sys, torch
sys.path.insert(0, ".")
import mlsp_amd.functional as Fh
Cout = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, N, C, k = 32, 1024, 64, 20
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
z = torch.randn(B * N, 3, generator=g)
lift = torch.randn(3, C, generator=g)
xp = (torch.tanh(z @ lift) + 0.01 * torch.randn(B * N, C, generator=g)).to(dev).requires_grad_(True)
W = (torch.randn(Cout, 2 * C, generator=g) * 0.1).to(dev).requires_grad_(True)
gamma = torch.ones(Cout, device=dev, requires_grad=True); beta = torch.zeros(Cout, device=dev, requires_grad=True)
rm = torch.zeros(Cout, device=dev); rv = torch.ones(Cout, device=dev)
graph = Fh.knn_graph(xp.detach(), B, N, k)
deg = (graph.rev_off[1:] - graph.rev_off[:-1]).float()
print("in-degree: mean %.1f max %d, share of destinations above 32: %.3f, share of entries there: %.3f" % (
    deg.mean().item(), int(deg.max().item()), (deg > 32).float().mean().item(), (deg[deg > 32].sum() / deg.sum()).item()))
for _ in range(iters):
    out = Fh.edgeconv(xp, graph, W, gamma, beta, rm, rv, True)
    out.backward(torch.ones_like(out))
torch.cuda.synchronize()
