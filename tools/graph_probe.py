"""Is the headline step enqueue-bound anywhere?  Captures ONE whole step (zero_grad, forward, losses, backward, fused Adam) in a HIP graph
and times its replay against the eager step on the same box.  TIMING PROBE ONLY: a replay re-runs the captured dropout seeds (the seeds are
kernel arguments), so replayed steps reuse one dropout mask -- not a training mode."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from mlsp_amd import Models, mlsp

dev = torch.device("cuda:0")
args = bench.make_args()
torch.manual_seed(0)
model = Models.DGCNN(args).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5, fused=True, capturable=True)
batch = bench.synth_batch(32, 1024, dev, seed=1000)
step = lambda: bench.gpu_step(model, mlsp, args, batch, opt)
print("eager %.3f ms/step" % bench.median_block_ms(step, 20, 5, 5), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
print("captured; loss", loss.item(), flush=True)
for _ in range(2):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print("graph replay %.3f ms/step (blocks %s) loss %.6f" % (sorted(ts)[2], ["%.3f" % t for t in ts], loss.item()), flush=True)
    print("eager %.3f ms/step" % bench.median_block_ms(step, 20, 5, 5), flush=True)
