"""Diagnostic (GPU box): forward error of every SA layer output vs the float64 oracle, HIP fp32 vs oracle fp32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import ref_sa_cpu as sa
from mlsp_amd import pointnet2 as p2
dev = torch.device("cuda:0")
torch.manual_seed(7)
B, N = int(os.environ.get("SA_B", 32)), 2048
xyz = torch.rand(B, N, 3) * 2 - 1
cfgs = [dict(npoint=512, radius=0.2, nsample=32, D=0, mlp=[64, 64, 128], group_all=False),
        dict(npoint=128, radius=0.4, nsample=64, D=128, mlp=[128, 128, 256], group_all=False),
        dict(npoint=None, radius=None, nsample=None, D=256, mlp=[256, 512, 1024], group_all=True)]
layers = [p2.PointNetSetAbstraction(c["npoint"], c["radius"], c["nsample"], 3 + c["D"], c["mlp"], c["group_all"]) for c in cfgs]
starts = [torch.randint(0, N, (B,)), torch.randint(0, 512, (B,)), None]
def oracle(dtype):
    outs, x, f = [], xyz.to(dtype), None
    with torch.no_grad():
        for l, c, st in zip(layers, cfgs, starts):
            pr = {k: v.detach().clone().to(dtype) for k, v in l.named_parameters()}
            bf = {k: v.detach().clone().to(dtype) for k, v in l.named_buffers() if not k.endswith("num_batches_tracked")}
            x, f, _, _, _ = sa.sa_forward(pr, bf, c, x, f, st)
            outs.append(f.double())
    return outs
o32, o64 = oracle(torch.float32), oracle(torch.float64)
outs, gx, gf = [], xyz.to(dev), None
with torch.no_grad():
    for l, st in zip(layers, starts):
        l.to(dev).train()
        if st is not None: l.fps_start = st
        gx, gf = l(gx, gf)
        outs.append(gf.double().cpu())
for i in range(3):
    e_h = (outs[i] - o64[i]).abs(); e_o = (o32[i] - o64[i]).abs()
    print("layer %d out: HIP max abs %.2e rel-L2 %.2e | oracle fp32 max abs %.2e rel-L2 %.2e" % (
        i + 1, e_h.max(), e_h.norm() / o64[i].norm(), e_o.max(), e_o.norm() / o64[i].norm()))
