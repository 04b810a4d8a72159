"""Is the head stack (pointmlp -> multimlp x2 -> final Linear layers) bitwise reproducible run to run?  python tools/r5/determinism_probe.py"""
import itertools as it
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mlsp_amd import functional as Fh

dev = torch.device("cuda:0")
M, C0, training = int(os.environ.get("M", 16384)), 512, True


def rnd(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * 2 - 1


def run(defer):
    Fh._DEFER_CHAINS = defer
    Fh._seed_counter = it.count(4321)
    g = torch.Generator().manual_seed(5)
    def r(*shape, s=1.0):
        return ((torch.rand(shape, generator=g) * 2 - 1) * s).to(dev).requires_grad_(True)
    X = r(M, C0)
    W1, g1, b1 = r(1024, C0, s=0.1), r(1024), r(1024)
    d0 = [(r(256, 256, s=0.1), None), (r(256, 256, s=0.1), None), (r(256, 512, s=0.1), r(256))]
    d1 = [(r(128, 256, s=0.1), None), (r(128, 256, s=0.1), None), (r(256, 256, s=0.1), r(256))]
    gb0, gb1 = (r(768), r(768)), (r(512), r(512))
    fin = [(r(3, 128, s=0.2), None), (r(3, 128, s=0.2), None), (r(16, 256, s=0.2), r(16))]
    stats = [torch.zeros(1024, device=dev), torch.ones(1024, device=dev), torch.zeros(768, device=dev), torch.ones(768, device=dev),
             torch.zeros(512, device=dev), torch.ones(512, device=dev)]
    h = Fh.pointmlp(X, W1, gamma=g1, beta=b1, run_mean=stats[0], run_var=stats[1], training=training, act=Fh.ACT_RELU, p_drop=0.5, chain=True)
    spec0 = ((256, 0.0, True), (256, 0.0, True), (256, 0.2, True))
    spec1 = ((128, 0.0, False), (128, 0.0, False), (256, 0.2, True))
    h = Fh.multimlp(h, [(0, d0[0][0], d0[0][1]), (256, d0[1][0], d0[1][1]), (512, d0[2][0], d0[2][1])], gb0[0], gb0[1], stats[2], stats[3],
                    Fh.channel_params(dev, spec0), training=training, p_drop=0.5, chain=True, spec=spec0)
    h = Fh.multimlp(h, [(0, d1[0][0], d1[0][1]), (256, d1[1][0], d1[1][1]), (512, d1[2][0], d1[2][1])], gb1[0], gb1[1], stats[4], stats[5],
                    Fh.channel_params(dev, spec1), training=training, p_drop=0.5, chain=True, spec=spec1)
    slices, cols = Fh.split_columns_shared(h, [128, 128, 256])
    outs, col = [], 0
    for sl, (W, b) in zip(slices, fin):
        outs.append(Fh.pointmlp(sl, W, bias=b, training=training, grad_cols=(cols, col)))
        col += sl.shape[1]
    loss = sum((o * rnd(tuple(o.shape), 40 + i).to(dev)).sum() for i, o in enumerate(outs))
    loss.backward()
    leaves = [X, W1, g1, b1] + [t for pair in d0 + d1 + fin for t in pair if t is not None] + list(gb0) + list(gb1)
    names = ["o0", "o1", "o2", "dX", "dW1", "dg1", "db1"] + ["p%d" % i for i in range(len(leaves) - 4)] + ["s%d" % i for i in range(6)]
    return names, [o.detach().cpu() for o in outs] + [t.grad.cpu() for t in leaves] + [s.cpu() for s in stats]


for mode in ("fp32", "bf16x6"):
    for defer in (True, False):
        with Fh.gemm_precision(mode):
            names, ref = run(defer)
            for rep in range(int(os.environ.get("REPS", 6))):
                _, got = run(defer)
                bad = [(n, (a - b).abs().max().item()) for n, a, b in zip(names, ref, got) if not torch.equal(a, b)]
                if bad:
                    print(mode, "defer" if defer else "plain", "rep", rep, "NOT reproducible:", bad[:6])
        print(mode, "defer" if defer else "plain", "done")

# defer vs plain per mode (what the test compares), optionally with torch.empty() filled with NaN (NANFILL=1): an uninitialised read shows up as NaN
if os.environ.get("NANFILL"):
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
for mode in ("fp32", "bf16x6"):
    with Fh.gemm_precision(mode):
        names, a = run(True)
        _, b = run(False)
    for n, x, y in zip(names, a, b):
        rel = ((x - y).double().norm() / (y.double().norm() + 1e-30)).item()
        nan = bool(torch.isnan(x).any() or torch.isnan(y).any())
        if rel > 0 or nan:
            print(mode, n, tuple(x.shape), "rel %.3e" % rel, "NaN" if nan else "")
    print(mode, "compared")

# cross-process reproducibility: a checksum per result tensor and configuration (compare the lines of several fresh processes)
import hashlib
for mode in ("fp32", "bf16x6"):
    for defer in (True, False):
        with Fh.gemm_precision(mode):
            names, res = run(defer)
        print("SUM", mode, "defer" if defer else "plain", " ".join("%s:%s" % (n, hashlib.md5(t.numpy().tobytes()).hexdigest()[:6]) for n, t in zip(names, res)))
