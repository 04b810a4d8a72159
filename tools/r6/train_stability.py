"""Round 6: 300 optimizer steps of the headline workload (one fixed synthetic batch, lr 1e-3) in product modes f16x3 and bf16x6 from the
same initial weights and the same dropout streams: loss at steps 1, 2, 5, 10, 20 and every 50 (the two trajectories agree to rounding at first and
part as every fp32 training run does once a max-pool selection flips), finiteness of every parameter, and -- f16x3 -- the invariant the mode rests on: the bound
FlatAdam publishes for every 2-D weight (per-tile maxima left by the step kernel) is >= max|W| and within 2x of it, at every checkpoint."""
import os, sys, copy, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from mlsp_amd import Models, mlsp, functional as Fh

dev = torch.device("cuda:0")
args = bench.make_args()
torch.manual_seed(0)
base = Models.DGCNN(args).to(dev).train()
batch = bench.synth_batch(32, 1024, dev, seed=1000)
STEPS = int(os.environ.get("STEPS", 300))
for mode in ("f16x3", "bf16x6"):
    model = copy.deepcopy(base)
    opt = bench.make_adam(model.parameters())
    torch.manual_seed(1)
    Fh._seed_counter = itertools.count()          # the same dropout streams in both modes
    losses = []
    with Fh.gemm_precision(mode):
        for it in range(1, STEPS + 1):
            loss = bench.gpu_step(model, mlsp, args, batch, opt)
            if it % 50 == 0 or it in (1, 2, 5, 10, 20):
                l = loss.item()
                fin = all(torch.isfinite(p).all().item() for p in model.parameters())
                line = "%s step %4d loss %.5f params finite %s" % (mode, it, l, fin)
                if mode == "f16x3" and hasattr(opt, "weight_bounds"):
                    worst_lo, worst_hi, n = 1e9, 0.0, 0
                    for name, p in model.named_parameters():
                        W = p.view(p.shape[0], -1) if p.dim() >= 2 else None
                        if W is None or W.shape[1] < 32:
                            continue
                        wb = opt.weight_bounds(W)
                        if wb is None:
                            continue
                        ta = opt._flat["tile_amax"]
                        t0 = (wb[0] - ta.data_ptr()) // 4
                        bound = ta[t0:t0 + wb[1]].max().item()
                        r = bound / W.abs().max().item()
                        worst_lo, worst_hi, n = min(worst_lo, r), max(worst_hi, r), n + 1
                    line += "  published bound / max|W| over %d weights: %.4f .. %.4f" % (n, worst_lo, worst_hi)
                print(line, flush=True)
