"""Time of the reverse-neighbour-list build (mlsp_knn_reverse) at the bench shape (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlsp_amd import functional as Fh, _lib
dev = torch.device("cuda:0")
B, N, k = int(os.environ.get("TN_B", 32)), int(os.environ.get("TN_N", 1024)), int(os.environ.get("TN_K", 20))
x = torch.rand(B * N, 3, device=dev)
g = Fh.knn_graph(x, B, N, k)
lib = _lib.load()
ro, re = torch.empty_like(g.rev_off), torch.empty_like(g.rev_ent)
def run():
    _lib.check(lib.mlsp_knn_reverse(g.idx.data_ptr(), B, N, k, ro.data_ptr(), re.data_ptr(), _lib.stream()), "rev")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print("knn_reverse B=%d N=%d k=%d: %.1f us; equal to graph build: %s" % (B, N, k, e0.elapsed_time(e1) / 50 * 1e3,
      bool((ro == g.rev_off).all() and (re == g.rev_ent).all())))
