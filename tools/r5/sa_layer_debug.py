"""debug: ONE set-abstraction layer, forced selections, every gradient vs the float64 oracle, in several configurations of the HIP path"""
import os, sys, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import ref_sa_cpu as sa
from mlsp_amd import functional as Fh, pointnet2 as p2
dev = torch.device("cuda:0")
rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))

def run(name, cfg, B, N, seed, signs):
    torch.manual_seed(seed)
    xyz = torch.rand(B, N, 3) * 2 - 1
    feat = torch.randn(B, N, cfg["D"]) if cfg["D"] else None
    layer = p2.PointNetSetAbstraction(cfg["npoint"], cfg["radius"], cfg["nsample"], 3 + cfg["D"], cfg["mlp"], cfg["group_all"])
    with torch.no_grad():
        for bn in layer.mlp_bns:
            bn.weight.mul_(1.0 + 0.3 * torch.randn_like(bn.weight))
            if signs:
                bn.weight[torch.rand_like(bn.weight) < 0.2] *= -1
            bn.bias.add_(0.2 * torch.randn_like(bn.bias))
    start = None if cfg["group_all"] else torch.randint(0, N, (B,))
    C = cfg["mlp"][-1]
    S = 1 if cfg["group_all"] else cfg["npoint"]
    wgt = torch.randn(B, S, C)

    def oracle(dtype, sel=None):
        pr = {k: v.detach().cpu().clone().to(dtype).requires_grad_(True) for k, v in layer.named_parameters()}
        bf = {k: v.detach().cpu().clone().to(dtype) for k, v in layer.named_buffers() if not k.endswith("num_batches_tracked")}
        x = xyz.detach().clone().to(dtype).requires_grad_(True)
        f = feat.detach().clone().to(dtype).requires_grad_(True) if feat is not None else None
        rec = []
        _, out, _, gidx, _ = sa.sa_forward(pr, bf, cfg, x, f, start, sel=sel, sel_out=rec if sel is None else None)
        (out * wgt.to(dtype)).sum().backward()
        g = {k: v.grad.double().numpy() for k, v in pr.items()}
        g["d_xyz"] = x.grad.double().numpy()
        if f is not None:
            g["d_feat"] = f.grad.double().numpy()
        return g, (rec[0] if rec else None), gidx

    g64, sel, gidx = oracle(torch.float64)
    if gidx is not None:
        pad = gidx == gidx[..., :1]
        sel = torch.where(torch.gather(pad, 2, sel), torch.zeros_like(sel), sel)
    g32, _, _ = oracle(torch.float32, sel)
    for label, fuse, prec in (("fused max, bf16x6", True, "bf16x6"), ("fused max, fp32", True, "fp32"), ("separate max, fp32", False, "fp32")):
        layer.to(dev).train(); layer.zero_grad(set_to_none=True)
        layer.fuse_max = fuse
        if start is not None:
            layer.fps_start = start
        gx = xyz.detach().to(dev).clone().requires_grad_(True)
        gf = feat.detach().to(dev).clone().requires_grad_(True) if feat is not None else None
        try:
            with Fh.gemm_precision(prec), Fh.forced_selections([sel.reshape(-1, C)]):
                _, out = layer(gx, gf)
            (out * wgt.to(dev)).sum().backward()
        except AssertionError as e:
            print(name, label, "no selection hook on this path:", str(e)[:60]); continue
        g = {k: p.grad.double().cpu().numpy() for k, p in layer.named_parameters()}
        if gx.grad is not None:
            g["d_xyz"] = gx.grad.double().cpu().numpy()
        if gf is not None:
            g["d_feat"] = gf.grad.double().cpu().numpy()
        print(name, "|", label)
        for k in g64:
            if ("mlp_convs" in k and k.endswith("bias")) or k not in g:
                continue
            print("   %-22s HIP %.1e   fp32 oracle %.1e" % (k, rel(g[k], g64[k]), rel(g32[k], g64[k])))

run("group_all D=256", dict(npoint=None, radius=None, nsample=None, D=256, mlp=[256, 512, 1024], group_all=True), 8, 64, 1, True)
run("ball D=128", dict(npoint=64, radius=0.5, nsample=64, D=128, mlp=[128, 128, 256], group_all=False), 8, 256, 2, True)
