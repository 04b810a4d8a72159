"""Where does the host spend its time while it enqueues one step?  cProfile over 20 steps (no sync inside), top entries by own and by
cumulative time.  python tools/r5/host_profile.py"""
import cProfile, os, pstats, sys, time, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
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
N = 20
t0 = time.perf_counter()
for _ in range(N):
    bench.gpu_step(model, mlsp, args, batch, opt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("unprofiled: enqueue %.2f ms/step, wall %.2f ms/step" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
# forward / backward / optimizer split of the enqueue time
tf = tb = to = 0.0
for _ in range(N):
    torch.cuda.synchronize()
    a = time.perf_counter()
    opt.zero_grad()
    logits = model(batch["x"], activate_density_normal_ondef=True)
    loss = mlsp.calc_loss(args, logits, batch["gold"], batch["mask"])
    loss = loss + mlsp.calc_masked_normal_loss(args, logits["Normal"], batch["normal_gt"], batch["mask_cord"])
    kl, mae = mlsp.densityloss(args, logits, batch["dens_val"], batch["dens_vec"], mask=batch["mask_cord"].reshape(-1))
    loss = loss + kl + mae
    b = time.perf_counter()
    loss.backward()
    c = time.perf_counter()
    opt.step()
    d = time.perf_counter()
    tf += b - a; tb += c - b; to += d - c
print("enqueue from idle: forward+losses %.2f ms, backward %.2f ms, optimizer %.2f ms" % (1e3 * tf / N, 1e3 * tb / N, 1e3 * to / N))
# host-bound probe: the same step on a tiny batch (B = 2, N = 128: the GPU finishes each launch long before the next one arrives),
# 200 steps back to back -> ms per step = what the host needs to enqueue one step (same launch count as the bench step)
tb_ = bench.synth_batch(2, 128, dev)
for _ in range(20):
    bench.gpu_step(model, mlsp, args, tb_, opt)
torch.cuda.synchronize()
import gc
gc.collect(); gc.disable()
t0 = time.perf_counter()
for _ in range(200):
    bench.gpu_step(model, mlsp, args, tb_, opt)
torch.cuda.synchronize()
gc.enable()
print("host-bound probe (B=2, N=128): %.3f ms per step" % (1e3 * (time.perf_counter() - t0) / 200))
if os.environ.get("NO_BWD_THREADS"):
    torch.autograd.set_multithreading_enabled(False)
    t0 = time.perf_counter()
    for _ in range(200):
        bench.gpu_step(model, mlsp, args, tb_, opt)
    torch.cuda.synchronize()
    print("host-bound probe, backward on the calling thread: %.3f ms per step" % (1e3 * (time.perf_counter() - t0) / 200))
    model_step = lambda: bench.gpu_step(model, mlsp, args, tb_, opt)
if os.environ.get("PROBE_ONLY"):
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    bench.gpu_step(model, mlsp, args, tb_ if os.environ.get("NO_BWD_THREADS") else batch, opt)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print("\n".join(l for l in s.getvalue().splitlines()[:90]))
