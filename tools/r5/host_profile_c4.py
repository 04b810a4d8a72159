"""Host enqueue time of the configs[4] step (PointSegDA DGCNN_DefRec, k = 40): the same step on a tiny batch (B = 2, N = 256: the GPU
finishes each launch long before the next arrives) -> ms per step the HOST needs; then cProfile of 20 such steps.
python tools/r5/host_profile_c4.py"""
import cProfile, io, os, pstats, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_common as gcm
from mlsp_amd import seg_models, functional as Fh
from mlsp_amd.optim import FlatAdam
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda:0")
B, N, K = 2, 256, 40
seg = seg_models.DGCNN_DefRec(gcm.make_seg_args(dropout=0.5, gpu=True), in_size=3, num_classes=8)
seg.k = seg.shared_layers.k = K
seg = seg.to(dev).train()
opt = FlatAdam(seg.parameters(), lr=1e-3, weight_decay=5e-5)
x = torch.rand(B, 3, N, device=dev) * 2 - 1
w = {k: torch.randn(s, device=dev) for k, s in (("seg", (B, N, 8)), ("DefRec", (B, N, 3)), ("Normal", (B, N, 3)), ("density", (B * N, 16)),
                                                ("density_mse", (B * N,)))}
mode = os.environ.get("C4_LOSS", "mean")


def step():
    opt.zero_grad()
    out = seg(x, make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
    if mode == "mean":
        sum((out[k].float() * w[k]).mean() for k in w).backward()
    else:
        torch.autograd.backward([out[k] for k in w], [w[k] for k in w])
    opt.step()


with Fh.gemm_precision("bf16"), Fh.activation_storage("bf16"):
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        print("host-bound probe (B=2, N=256, loss=%s): %.3f ms per step" % (mode, 1e3 * (time.perf_counter() - t0) / 100))
    # forward / backward / optimizer split
    tf = tb = to = 0.0
    for _ in range(50):
        torch.cuda.synchronize()
        a = time.perf_counter()
        opt.zero_grad()
        out = seg(x, make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
        loss = sum((out[k].float() * w[k]).mean() for k in w)
        b_ = time.perf_counter()
        loss.backward()
        c = time.perf_counter()
        opt.step()
        d = time.perf_counter()
        tf += b_ - a; tb += c - b_; to += d - c
    print("enqueue from idle: forward+loss %.2f ms, backward %.2f ms, optimizer %.2f ms" % (1e3 * tf / 50, 1e3 * tb / 50, 1e3 * to / 50))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(60)
    print("\n".join(l for l in s.getvalue().splitlines()[:100]))
