"""How long would the configs[4] step (PointSegDA DGCNN_DefRec, B = 16, N = 2048, k = 40, bf16 storage) take if the host cost nothing?
Captures ONE whole step (zero_grad, forward, synthetic loss, backward, Adam) in a HIP graph and times its replay against the eager step on
the same box.  TIMING PROBE ONLY (as tools/graph_probe.py): the dropout seeds and the product mode are kernel ARGUMENTS, so a replay repeats
one dropout mask -- not a training mode.  python tools/r5/graph_probe_c4.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_common as gc
from mlsp_amd import seg_models, functional as Fh

torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda:0")
B, N, K = int(os.environ.get("C4_B", 16)), 2048, 40
seg = seg_models.DGCNN_DefRec(gc.make_seg_args(dropout=0.5, gpu=True), in_size=3, num_classes=8)
seg.k = seg.shared_layers.k = K
seg = seg.to(dev).train()
opt = torch.optim.Adam(seg.parameters(), lr=1e-3, weight_decay=5e-5, fused=True, capturable=True)
x = torch.rand(B, 3, N, device=dev) * 2 - 1
w = {k: torch.randn(s, device=dev) for k, s in (("seg", (B, N, 8)), ("DefRec", (B, N, 3)), ("Normal", (B, N, 3)), ("density", (B * N, 16)),
                                                ("density_mse", (B * N,)))}


def step():
    opt.zero_grad(set_to_none=False)
    out = seg(x, make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
    sum((out[k].float() * w[k]).mean() for k in w).backward()
    opt.step()


def timed(fn, n=20, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[len(ts) // 2]


with Fh.gemm_precision("bf16"), Fh.activation_storage("bf16"):
    for _ in range(5):
        step()
    print("eager            %.3f ms/step" % timed(step), flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    print("captured", flush=True)
    print("graph replay     %.3f ms/step   (device-bound time of the same launches)" % timed(g.replay), flush=True)
    print("eager            %.3f ms/step" % timed(step), flush=True)
