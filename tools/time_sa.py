"""Times the PointNet++ SA encoder stack at BASELINE.json configs[3] (B=32, N=2048) fwd+bwd on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlsp_amd import pointnet2 as p2

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, N = 32, 2048
xyz = (torch.rand(B, N, 3) * 2 - 1).to(dev)
layers = [p2.PointNetSetAbstraction(512, 0.2, 32, 3, [64, 64, 128], False), p2.PointNetSetAbstraction(128, 0.4, 64, 131, [128, 128, 256], False),
          p2.PointNetSetAbstraction(None, None, None, 259, [256, 512, 1024], True)]
for l in layers:
    l.to(dev)
params = [p for l in layers for p in l.parameters()]


def step():
    for p in params:
        p.grad = None
    x, f = xyz, None
    for l in layers:
        x, f = l(x, f)
    f.sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print("SA encoder (configs[3]) fwd+bwd: %.2f ms/step, %.2f M points/s" % (dt * 1e3, B * N / dt / 1e6))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    idx = p2.farthest_point_sample(xyz, 512, start=torch.zeros(B, dtype=torch.long))
e1.record(); torch.cuda.synchronize()
print("FPS 2048 -> 512: %.1f us" % (e0.elapsed_time(e1) / 5 * 1e3))
new_xyz = p2.index_points(xyz, idx)
e0.record()
for _ in range(5):
    p2.query_ball_point(0.2, 32, xyz, new_xyz)
e1.record(); torch.cuda.synchronize()
print("ball query 512 x 2048, nsample 32: %.1f us" % (e0.elapsed_time(e1) / 5 * 1e3))
if os.environ.get("DUMP"):        # every GEMM launch of one step with its HIP-event time on stderr
    import bench
    from mlsp_amd import _lib
    os.environ["MLSP_PROF_DUMP"] = "1"
    bench.profiled_steps(_lib.load(), step, 1)
