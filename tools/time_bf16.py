"""Step time of the DGCNN bench workload and of PointSegDA (N=2048) with fp32 vs bf16-operand GEMMs (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import golden_common as gc
from mlsp_amd import Models, mlsp, seg_models, functional as Fh

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


args = bench.make_args()
torch.manual_seed(0)
model = Models.DGCNN(args).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
batch = bench.synth_batch(32, 1024, dev, seed=1)
for mode in ("fp32", "bf16"):
    with Fh.gemm_precision(mode):
        ms = timeit(lambda: bench.gpu_step(model, mlsp, args, batch, opt))
    print("DGCNN+MLSP B=32 N=1024 k=20 step, %s GEMM operands: %.2f ms (%.2f M pts/s)" % (mode, ms, 32 * 1024 / ms / 1e3))

seg = seg_models.DGCNN_DefRec(gc.make_seg_args(dropout=0.5, gpu=True), in_size=3, num_classes=8).to(dev).train()
sopt = torch.optim.Adam(seg.parameters(), lr=1e-3, fused=True)
xs = torch.rand(16, 3, 2048, device=dev) * 2 - 1


def seg_step():
    sopt.zero_grad()
    out = seg(xs, activate_density_normal_ondef=True)
    sum(v.float().mean() for v in out.values()).backward()
    sopt.step()


for mode in ("fp32", "bf16"):
    with Fh.gemm_precision(mode):
        ms = timeit(seg_step, 5)
    print("PointSegDA B=16 N=2048 k=20 step, %s GEMM operands: %.2f ms (%.2f M pts/s)" % (mode, ms, 16 * 2048 / ms / 1e3))
