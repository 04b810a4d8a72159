"""debug: SA stack gradients -- free-running vs forced with the HIP path's OWN recorded selections (must be identical) vs the f64 oracle's"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import ref_sa_cpu as sa
from mlsp_amd import functional as Fh, pointnet2 as p2
dev = torch.device("cuda:0")
torch.manual_seed(11)
B, N = 8, 1024
xyz = torch.rand(B, N, 3) * 2 - 1
cfgs = [dict(npoint=256, radius=0.25, nsample=32, D=0, mlp=[64, 64, 128], group_all=False),
        dict(npoint=64, radius=0.5, nsample=64, D=128, mlp=[128, 128, 256], group_all=False),
        dict(npoint=None, radius=None, nsample=None, D=256, mlp=[256, 512, 1024], group_all=True)]
layers = [p2.PointNetSetAbstraction(c["npoint"], c["radius"], c["nsample"], 3 + c["D"], c["mlp"], c["group_all"]) for c in cfgs]
starts = [torch.randint(0, N, (B,)), torch.randint(0, 256, (B,)), None]
with torch.no_grad():
    for l in layers:
        for bn in l.mlp_bns:
            bn.weight.mul_(1.0 + 0.3 * torch.randn_like(bn.weight))
            bn.weight[torch.rand_like(bn.weight) < 0.2] *= -1
            bn.bias.add_(0.2 * torch.randn_like(bn.bias))
wgt = torch.randn(B, 1, 1024)
GROUPS = {}
def oracle(dtype, sels=None):
    x, f, prs, rec = xyz.detach().clone().to(dtype), None, [], []
    for li, (l, c, st) in enumerate(zip(layers, cfgs, starts)):
        pr = {k: v.detach().cpu().clone().to(dtype).requires_grad_(True) for k, v in l.named_parameters()}
        bf = {k: v.detach().cpu().clone().to(dtype) for k, v in l.named_buffers() if not k.endswith("num_batches_tracked")}
        x, f, _, gidx, _ = sa.sa_forward(pr, bf, c, x, f, st, sel=None if sels is None else sels[li], sel_out=rec if sels is None else None)
        prs.append(pr); GROUPS[li] = gidx
    (f * wgt.to(dtype)).sum().backward()
    return f.detach(), [{k: v.grad.double().numpy() for k, v in pr.items()} for pr in prs], rec

def hip(sels=None, fold=True, fuse=True, prec='bf16x6'):
    for l in layers:
        l.to(dev).train(); l.zero_grad(set_to_none=True); l.fold_first = fold; l.fuse_max = fuse
    gx, gf = xyz.to(dev), None
    with Fh.gemm_precision(prec), Fh.forced_selections(sels):
        for l, st in zip(layers, starts):
            if st is not None:
                l.fps_start = st
            gx, gf = l(gx, gf)
    (gf * wgt.to(dev)).sum().backward()
    return [{k: p.grad.double().cpu().numpy() for k, p in l.named_parameters()} for l in layers]

rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
o64, g64, sel64 = oracle(torch.float64)
for li, gidx in GROUPS.items():
    if gidx is not None:
        pad = gidx == gidx[..., :1]
        sel64[li] = torch.where(torch.gather(pad, 2, sel64[li]), torch.zeros_like(sel64[li]), sel64[li])
_, g32, _ = oracle(torch.float32, sel64)
fs = [s.reshape(-1, s.shape[-1]) for s in sel64]
runs = {}
for name, kw in (("default", {}), ("nofuse", dict(fuse=False)), ("nofold", dict(fold=False)), ("nofold nofuse fp32", dict(fold=False, fuse=False, prec="fp32"))):
    try:
        runs[name] = hip(sels=[t.clone() for t in fs], **kw)
    except AssertionError as e:
        print(name, "failed:", str(e)[:80])
for li in range(3):
    for k in g64[li]:
        if "mlp_convs" in k and k.endswith("bias"):
            continue
        print("sa%d %-20s yard %.1e | " % (li + 1, k, rel(g32[li][k], g64[li][k])) + "  ".join("%s %.1e" % (n, rel(g[li][k], g64[li][k])) for n, g in runs.items()))

# forward values layer by layer: HIP (free-running) vs the float64 oracle
with torch.no_grad():
    x64, f64 = xyz.double(), None
    gx, gf = xyz.to(dev), None
    for li, (l, c, st) in enumerate(zip(layers, cfgs, starts)):
        pr = {k: v.detach().cpu().double() for k, v in l.named_parameters()}
        bf = {k: v.detach().cpu().double() for k, v in l.named_buffers() if not k.endswith("num_batches_tracked")}
        x64, f64, _, _, _ = sa.sa_forward(pr, bf, c, x64, f64, st)
        l.train(); l.fuse_max = True; l.fold_first = True
        if st is not None:
            l.fps_start = st
        with Fh.recorded_selections() as rec:
            gx, gf = l(gx, gf)
        d = (gf.double().cpu() - f64).abs()
        print("forward sa%d: rel-L2 %.2e  max abs %.2e (|f|max %.2f)  entries off by > 1e-4: %d of %d;  own selection == forced: %.4f" % (
            li + 1, rel(gf.double().cpu().numpy(), f64.numpy()), d.max().item(), f64.abs().max().item(), int((d > 1e-4).sum()), d.numel(),
            (rec.sel[0].cpu().long() == fs[li]).float().mean().item()))
