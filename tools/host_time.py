"""How long does the host take to ENQUEUE one step (no sync)?  Tells how far the step is from host-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from mlsp_amd import Models, mlsp
from mlsp_amd.ddp import FlatGradSync
dev = torch.device("cuda:0")
args = bench.make_args()
torch.manual_seed(0)
model = Models.DGCNN(args).to(dev).train()
sync = FlatGradSync(model)
opt = sync.wrap(torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5, fused=True))
batch = bench.synth_batch(32, 1024, dev)
for _ in range(5):
    bench.gpu_step(model, mlsp, args, batch, opt)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.gpu_step(model, mlsp, args, batch, opt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("host enqueue ms: %.2f   step wall ms (from idle): %.2f" % (1e3 * sum(a for a, _ in ts) / len(ts), 1e3 * sum(b for _, b in ts) / len(ts)))
