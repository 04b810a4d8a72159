"""Round 6: which operands the two-piece f16 products measure in one bench step (MLSP_AMAX_DUMP=1 python tools/r6/amax_dump.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from mlsp_amd import Models, mlsp
dev = torch.device("cuda:0")
args = bench.make_args()
torch.manual_seed(0)
model = Models.DGCNN(args).to(dev).train()
opt = bench.make_adam(model.parameters())
batch = bench.synth_batch(32, 1024, dev, seed=1000)
for i in range(3):
    if i == 2:
        torch.cuda.synchronize(); sys.stderr.write("==== step ====\n"); sys.stderr.flush()
    bench.gpu_step(model, mlsp, args, batch, opt)
torch.cuda.synchronize()
