"""Which Python lines of the step issue torch's own small device kernels (fills, copies, element-wise adds)?  One bench step under
torch.profiler with stacks; prints, per aten op that launched a kernel, the innermost mlsp_amd / bench frame.  Run on the GPU box."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from mlsp_amd import Models, mlsp
from mlsp_amd.ddp import FlatGradSync

dev = torch.device("cuda:0")
args = bench.make_args()
torch.manual_seed(0)
model = Models.DGCNN(args).to(dev).train()
opt = FlatGradSync(model).wrap(torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5, fused=True))
batch = bench.synth_batch(32, 1024, dev, seed=1000)
for _ in range(4):
    bench.gpu_step(model, mlsp, args, batch, opt)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode


class Rec(TorchDispatchMode):
    """Every aten op issued from this thread with the innermost mlsp_amd / bench frame (autograd's own nodes run on the engine
    thread and do not show up here: what is missing from this list compared with the profiler's is theirs)."""
    def __init__(self):
        super().__init__()
        self.cnt = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("fill", "zero", "cat", "add", "copy", "ones", "full", "mul", "sub", "div", "sum", "mean")):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if ("mlsp_amd" in fr.filename or "bench.py" in fr.filename or "optim" in fr.filename) and "find_small_ops" not in fr.filename:
                    where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, (fr.line or "")[:90])
                    break
            shapes = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))[:2]
            self.cnt[(name, where, str(shapes)[:60])] += 1
        return func(*args, **(kwargs or {}))


rec = Rec()
with rec:
    bench.gpu_step(model, mlsp, args, batch, opt)
torch.cuda.synchronize()
for (name, where, shapes), n in sorted(rec.cnt.items(), key=lambda x: -x[1]):
    print("%3d  %-28s %-60s %s" % (n, name, shapes, where))
