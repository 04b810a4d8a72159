"""configs[4] at its real shape (B = 16, N = 2048, k = 40, bf16 storage): host ENQUEUE time per step (the loop returns before the GPU is done:
perf_counter around 20 steps without a synchronisation) against the wall time per step.  enqueue ~= wall: the host is the bound.
python tools/r5/enqueue_c4.py"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_common as gcm
from mlsp_amd import seg_models, functional as Fh
from mlsp_amd.optim import FlatAdam
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda:0")
for B in (16, 8, 32):
    N, K = 2048, 40
    seg = seg_models.DGCNN_DefRec(gcm.make_seg_args(dropout=0.5, gpu=True), in_size=3, num_classes=8)
    seg.k = seg.shared_layers.k = K
    seg = seg.to(dev).train()
    opt = FlatAdam(seg.parameters(), lr=1e-3, weight_decay=5e-5)
    x = torch.rand(B, 3, N, device=dev) * 2 - 1
    w = {k: torch.randn(s, device=dev) for k, s in (("seg", (B, N, 8)), ("DefRec", (B, N, 3)), ("Normal", (B, N, 3)), ("density", (B * N, 16)),
                                                    ("density_mse", (B * N,)))}

    def step():
        opt.zero_grad()
        out = seg(x, make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
        sum((out[k].float() * w[k]).mean() for k in w).backward()
        opt.step()

    with Fh.gemm_precision("bf16"), Fh.activation_storage("bf16"):
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        gc.collect(); gc.disable()
        res = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            res.append(((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
        gc.enable()
    res.sort(key=lambda r: r[1])
    print("B=%d: enqueue %.3f ms/step, wall %.3f ms/step (median block of 5)" % (B, res[2][0], res[2][1]), flush=True)
    del seg, opt
