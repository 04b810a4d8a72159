import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
from mlsp_amd import Models, mlsp
from mlsp_amd.ddp import FlatGradSync
dev = torch.device("cuda:0")
args = bench.make_args(); torch.manual_seed(0)
model = Models.DGCNN(args).to(dev).train()
sync = FlatGradSync(model)
opt = sync.wrap(torch.optim.Adam(model.parameters(), lr=1e-3, fused=True))
batch = bench.synth_batch(4, 256, dev)
bench.gpu_step(model, mlsp, args, batch, opt)
opt.zero_grad()
logits = model(batch["x"], activate_density_normal_ondef=True)
mlsp.calc_loss(args, logits, batch["gold"], batch["mask"]).backward()
g0 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
flat = sync.pack()
dist.all_reduce(flat); torch.cuda.synchronize()
ok = all(torch.equal(g0[n], p.grad) for n, p in model.named_parameters() if p.grad is not None)
none = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
print("rccl allreduce ok, packed grads intact:", ok, "| params without grad this step:", len(none), "| bucket MB %.1f" % (flat.numel() * 4 / 1e6))
dist.destroy_process_group()
