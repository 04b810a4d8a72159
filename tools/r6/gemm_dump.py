"""Round 6: one line per GEMM-family launch of one headline bench step (MLSP_PROF_DUMP=1 python tools/r6/gemm_dump.py): shape, split-K,
tile height, kernel family, HIP-event time.  (The configs[4] step: MLSP_PROF_DUMP=1 C4_MODE=bf16 python tools/time_config4.py.)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from mlsp_amd import Models, mlsp, _lib
dev = torch.device("cuda:0")
args = bench.make_args()
torch.manual_seed(0)
model = Models.DGCNN(args).to(dev).train()
opt = bench.make_adam(model.parameters())
batch = bench.synth_batch(32, 1024, dev, seed=1000)
for i in range(3):
    bench.gpu_step(model, mlsp, args, batch, opt)
torch.cuda.synchronize()
lib = _lib.load()
lib.mlsp_profile_begin()
bench.gpu_step(model, mlsp, args, batch, opt)
torch.cuda.synchronize()
lib.mlsp_profile_end((ctypes.c_double * 4)())
