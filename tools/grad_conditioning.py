"""Where do the 1-3e-2 gradient differences between two fp32 evaluations of the DGCNN step come from?  Evaluates the ORACLE
(oracle/ref_cpu.py) in fp32 and in fp64 on a golden fixture's inputs with its graphs forced and prints, per gradient key, the
relative L2 distance oracle32-reference32, oracle32-oracle64 and reference32-oracle64: both fp32 runs sit ~1e-2 from the
double-precision value at N = 1024 (sparse Chamfer gradients through train-mode BatchNorm sums), i.e. fp32 rounding, not a
restatement error.  Uses the committed fixtures only (no /root/reference).

    python tools/grad_conditioning.py dgcnn_s1_B4_N1024.npz 1 4 1024
"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import golden_common as gc
from oracle import ref_cpu
from mlsp_amd import Models
fname, seed, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', fname)))
def run(dtype):
    torch.manual_seed(seed)
    m = Models.DGCNN(gc.make_args(dropout=0.0)); gc.perturb_params(m, seed)
    m = m.to(dtype)
    params = dict(m.state_dict(keep_vars=True))
    inp = {k:(v.to(dtype) if v.is_floating_point() else v) for k,v in gc.make_inputs(seed,B,N).items()}
    forced = iter([torch.from_numpy(g["knn%d"%i].astype(np.int64)) for i in range(5)])
    logits, nb, ctx = ref_cpu.dgcnn_forward(params, inp["x"], training=True, dropout_p=0.0, knn_fn=lambda x_,k_: next(forced), activate_density_normal_ondef=True, return_ctx=True)
    loss, parts = gc.total_loss(gc.make_args(), ref_cpu, logits, inp)
    loss.backward()
    return {k: p.grad.double().numpy() for k,p in m.named_parameters() if p.grad is not None}
g32 = run(torch.float32); g64 = run(torch.float64)
for key in [k[5:] for k in g if k.startswith("grad/")]:
    ref = g["grad/"+key].astype(np.float64); n=ref.shape[0]
    a, b = g32[key][:n], g64[key][:n]
    r = lambda x,y: np.linalg.norm(x-y)/(np.linalg.norm(y)+1e-30)
    print("%-50s oracle32-ref %.2e  oracle32-f64 %.2e  ref-f64 %.2e" % (key, r(a,ref), r(a,b), r(ref,b)))
