"""Diagnostic (GPU box): distance of EVERY parameter gradient of the B = 32, N = 1024 step from the float64 oracle, for the HIP path and
for the stock-torch fp32 CPU run (the yardstick), with the heads' first-layer gradient split into its x_cat / x5 halves.
    python tools/grad_diag.py [B]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import golden_common as gc
from oracle import knn_canon, ref_cpu, ref_torch_modules as rtm
from mlsp_amd import Models, mlsp, functional as Fh

dev = torch.device("cuda:0")
seed, B, N = 5, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 1024
torch.manual_seed(seed)
m = Models.DGCNN(gc.make_args(dropout=0.0, cuda=True)); gc.perturb_params(m, seed); m = m.to(dev)
ref = copy.deepcopy(m).cpu().double()
params = dict(ref.state_dict(keep_vars=True))
inp = gc.make_inputs(seed, B, N)
inp64 = {k: (v.double() if v.is_floating_point() else v) for k, v in inp.items()}
want, _, ctx = ref_cpu.dgcnn_forward(params, inp64["x"], training=True, dropout_p=0.0, knn_fn=lambda x_, k_: knn_canon.knn(x_.float(), k_),
                                     activate_density_normal_ondef=True, return_ctx=True)
wloss, _ = gc.total_loss(gc.make_args(), ref_cpu, want, inp64)
wloss.backward()
wg = {k: p.grad.numpy() for k, p in ref.named_parameters() if p.grad is not None}
graphs = [i.clone() for i in ctx.knn_idx]
it = iter(graphs)
stock = rtm.StockDGCNN(gc.make_args(dropout=0.0), knn_fn=lambda x_, k_: next(it)).train()
stock.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()}, strict=True)
sl, _ = gc.total_loss(gc.make_args(), ref_cpu, stock(inp["x"], activate_density_normal_ondef=True), inp)
sl.backward()
sg = {k: p.grad.double().numpy() for k, p in stock.named_parameters() if p.grad is not None}
ginp = {k: v.to(dev) for k, v in inp.items()}
m.train()
with Fh.forced_graphs(graphs):
    logits = m(ginp["x"], activate_density_normal_ondef=True)
args = gc.make_args(cuda=True)
loss = mlsp.calc_loss(args, logits, ginp["gold"], ginp["mask"])
mc = ginp["mask"].permute(0, 2, 1)[:, :, 0] * 26 + 1
loss = loss + mlsp.calc_masked_normal_loss(args, logits["Normal"], ginp["normal_gt"], mc)
kl, mae = mlsp.densityloss(args, logits, ginp["dens_val"], ginp["dens_vec"], mask=mc.reshape(-1))
loss = loss + kl + mae + torch.nn.functional.cross_entropy(logits["cls"], ginp["cls_label"])
loss.backward()
hg = {k: p.grad.double().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
rel = lambda a, b: np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300)
print("%-48s %10s %10s %7s" % ("parameter", "HIP", "CPU fp32", "ratio"))
for k in wg:
    if k not in hg:
        continue
    a, c = rel(hg[k].reshape(wg[k].shape), wg[k]), rel(sg[k].reshape(wg[k].shape), wg[k])
    print("%-48s %10.2e %10.2e %7.1f" % (k, a, c, a / max(c, 1e-30)))
    if k.endswith("conv1.weight") and wg[k].shape[1] == 1536:
        for nm, slc in (("  [:, :512] x_cat half", slice(0, 512)), ("  [:, 512:] x5 half", slice(512, 1536))):
            print("%-48s %10.2e %10.2e" % (nm, rel(hg[k][:, slc], wg[k][:, slc]), rel(sg[k][:, slc], wg[k][:, slc])))
