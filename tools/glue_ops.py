"""Lists the torch (non-mlsp) GPU ops of one training step with their Python call sites (run on the GPU box)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from torch.profiler import profile, ProfilerActivity

from mlsp_amd import Models, mlsp, ddp
dev = torch.device("cuda:0")
args = bench.make_args()
torch.manual_seed(0)
model = Models.DGCNN(args).to(dev)
model.train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5, fused=True)
batch = bench.synth_batch(32, 1024, dev, seed=0)
for _ in range(3):
    bench.gpu_step(model, mlsp, args, batch, opt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.gpu_step(model, mlsp, args, batch, opt)
    torch.cuda.synchronize()
agg = collections.Counter()
tim = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    kt = sum(k.duration for k in ev.kernels) if ev.kernels else 0
    if not ev.kernels:
        continue
    chain = []
    par = ev.cpu_parent
    while par is not None:
        chain.append(par.name)
        par = par.cpu_parent
    site = " < ".join(chain[:3]) if chain else "?"
    agg[(ev.name, site)] += 1
    tim[(ev.name, site)] += kt
for (name, site), n in sorted(agg.items(), key=lambda kv: -tim[kv[0]]):
    print("%-28s x%-3d %8.1f us  %s" % (name, n, tim[(name, site)], site))
