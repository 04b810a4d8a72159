"""BASELINE.json configs[4] on one MI355X (run on the GPU box): PointSegDA DGCNN_DefRec, N = 2048, k = 40, all heads, fwd + bwd + Adam,
B = 16 clouds per GPU (the reference trainer's default batch, PointSegDA/trainer.py:103), in three arithmetic modes:
fp32 | bf16 GEMM operands (fp32 storage) | bf16 operands + bf16 activation storage (the configs[4] mode).  One JSON line per mode."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_common as gc
from mlsp_amd import seg_models, functional as Fh

dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)      # as bench.py: the backward on the calling thread (host-side settings)
B, N, K = int(os.environ.get("C4_B", 16)), 2048, int(os.environ.get("C4_K", 40))
seg = seg_models.DGCNN_DefRec(gc.make_seg_args(dropout=0.5, gpu=True), in_size=3, num_classes=8)
seg.k = seg.shared_layers.k = K
seg = seg.to(dev).train()
if os.environ.get("MLSP_BENCH_TORCH_ADAM"):
    opt = torch.optim.Adam(seg.parameters(), lr=1e-3, weight_decay=5e-5, fused=True)
else:
    from mlsp_amd.optim import FlatAdam
    opt = FlatAdam(seg.parameters(), lr=1e-3, weight_decay=5e-5)       # as bench.py make_adam
x = torch.rand(B, 3, N, device=dev) * 2 - 1
w = {k: torch.randn(s, device=dev) for k, s in (("seg", (B, N, 8)), ("DefRec", (B, N, 3)), ("Normal", (B, N, 3)), ("density", (B * N, 16)),
                                                ("density_mse", (B * N,)))}


wg = {k: w[k] / w[k].numel() for k in w}
SEEDED = bool(os.environ.get("C4_SEEDED"))        # backward seeded with the synthetic loss' output gradients (no loss kernels): bench.py `seeded_backward`


def step():
    opt.zero_grad()
    out = seg(x, make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
    if SEEDED:
        torch.autograd.backward([out[k] for k in w], [wg[k] for k in w])
    else:
        sum((out[k].float() * w[k]).mean() for k in w).backward()
    opt.step()


MODES = (("fp32 (f32 MFMA)", "fp32", "fp32"), ("fp32-accurate split products bf16x6, fp32 storage (deferred activations)", "bf16x6", "fp32"),
         ("fp32-accurate split products f16x3, fp32 storage (deferred activations)", "f16x3", "fp32"),
         ("bf16 operands", "bf16", "fp32"), ("bf16 operands + bf16 activation storage", "bf16", "bf16"))
if os.environ.get("C4_MODE") == "bf16":          # profiling: the configs[4] mode only
    MODES = MODES[4:]
for name, prec, store in MODES:
    with Fh.gemm_precision(prec), Fh.activation_storage(store):
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        if os.environ.get("MLSP_PROF_DUMP"):               # one line per GEMM-family launch of one step (gemm.hip mlsp_profile_end)
            from mlsp_amd import _lib
            import ctypes
            _lib.load().mlsp_profile_begin(); step(); torch.cuda.synchronize(); _lib.load().mlsp_profile_end((ctypes.c_double * 4)())
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 10)
    ms = sorted(ts)[2] * 1e3
    print(json.dumps({"workload": "PointSegDA DGCNN_DefRec fwd+bwd+Adam, B=%d N=%d k=%d, all heads" % (B, N, K), "mode": name,
                      "ms_per_step": round(ms, 3), "points_per_s": round(B * N / ms * 1e3)}))
