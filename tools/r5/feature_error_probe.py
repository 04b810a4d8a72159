"""How far are the four EdgeConv outputs (x1..x4, the features the dynamic graph is rebuilt on) from the float64 truth -- HIP path vs the
reference's operators in fp32 on the CPU -- with the truth's graphs forced on both?  python tools/r5/feature_error_probe.py [seed B N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_common as gc
from oracle import ref_torch_modules as rtm
from mlsp_amd import Models, functional as Fh

seed, B, N = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (3, 4, 256)))
dev = torch.device("cuda:0")
torch.manual_seed(seed)
m = Models.DGCNN(gc.make_args(dropout=0.0, cuda=True)); gc.perturb_params(m, seed); m = m.to(dev).train()
inp = gc.make_inputs(seed, B, N)


def stock(dtype, knn_fn):
    rec, feats = [], []
    def spy(x_, k_):
        i = knn_fn(x_, k_); rec.append(i.clone().long()); return i
    ref = rtm.StockDGCNN(gc.make_args(dropout=0.0), knn_fn=spy)
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()}, strict=True)
    ref = ref.to(dtype).train()
    hooks = [c.register_forward_hook(lambda mod, i, o: feats.append(o.max(dim=-1)[0].detach().double())) for c in (ref.conv1, ref.conv2, ref.conv3, ref.conv4)]
    with torch.no_grad():
        ref(inp["x"].to(dtype), activate_density_normal_ondef=True)
    return rec, feats


idx64, f64 = stock(torch.float64, rtm.knn_matmul_topk)
it = iter(idx64)
_, f32 = stock(torch.float32, lambda x_, k_: next(it))
cap = {}
orig = Fh.join_columns
def spy_join(base, parts, accs=None):
    cap["x_cat"] = base.detach().clone()
    return orig(base, parts, accs)
Fh.join_columns = spy_join
for mode in ("bf16x6", "fp32"):
    with torch.no_grad(), Fh.gemm_precision(mode), Fh.forced_graphs(idx64):
        m(inp["x"].to(dev), activate_density_normal_ondef=True)
    xc = cap["x_cat"].cpu().double().view(B, N, -1)
    o = 0
    for li, (t, s) in enumerate(zip(f64, f32)):
        c = t.shape[1]
        h = xc[:, :, o:o + c].transpose(1, 2)
        o += c
        cen = t - t.mean(dim=2, keepdim=True)
        print("%s x%d [%d ch]: rel-L2 from float64  HIP %.2e | stock fp32 %.2e   (|x| / |x - cloud mean| = %.1f)"
              % (mode, li + 1, c, ((h - t).norm() / t.norm()).item(), ((s - t).norm() / t.norm()).item(), (t.norm() / cen.norm()).item()))
