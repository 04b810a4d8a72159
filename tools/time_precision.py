"""Headline step (BASELINE.json configs[1]) timed under each GEMM precision mode on ONE box, interleaved.
   python tools/time_precision.py [modes...]      default: fp32 bf16x6 fp32 bf16x6"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from mlsp_amd import Models, mlsp, functional as Fh

dev = torch.device("cuda:0")
args = bench.make_args()
torch.manual_seed(0)
model = Models.DGCNN(args).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5, fused=True)
batch = bench.synth_batch(32, 1024, dev, seed=1000)
for mode in (sys.argv[1:] or ["fp32", "bf16x6", "fp32", "bf16x6"]):
    with Fh.gemm_precision(mode):
        ms = bench.median_block_ms(lambda: bench.gpu_step(model, mlsp, args, batch, opt), 20, 5, 5)
    print(mode, "%.3f ms/step" % ms, flush=True)

if os.environ.get("DUMP"):        # one profiled step per mode: every GEMM launch with its HIP-event time on stderr (MLSP_PROF_DUMP)
    from mlsp_amd import _lib
    os.environ["MLSP_PROF_DUMP"] = "1"
    for mode in ("fp32", "bf16x6"):
        print("==== per-launch listing,", mode, file=sys.stderr, flush=True)
        with Fh.gemm_precision(mode):
            bench.profiled_steps(_lib.load(), lambda: bench.gpu_step(model, mlsp, args, batch, opt), 1)

if os.environ.get("DEFER_AB"):    # what the deferred BN+activation chains (operand transform in the consumer GEMM, fp32 kernel only) are worth
    for defer in (True, False, True, False):
        Fh._DEFER_CHAINS = defer
        ms = bench.median_block_ms(lambda: bench.gpu_step(model, mlsp, args, batch, opt), 20, 5, 5)
        print("fp32 defer_chains=%s %.3f ms/step" % (defer, ms), flush=True)
    Fh._DEFER_CHAINS = True
