"""kNN timings across the BASELINE.json configs (run on the GPU box): (B, N, C, k)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlsp_amd import functional as Fh

dev = torch.device("cuda:0")
OFFSET = float(os.environ.get("OFFSET", "0"))         # OFFSET=1.9: clouds as far from the origin as graph-stage features (max over k of BN + LeakyReLU outputs)
for (B, N, C, k) in [(32, 1024, 3, 20), (32, 1024, 64, 20), (32, 1024, 128, 20), (16, 2048, 3, 40), (16, 2048, 64, 40), (16, 2048, 128, 40),
                     (16, 2048, 64, 32), (16, 2048, 64, 20)]:
    x = (torch.rand(B * N, C, device=dev) * 2 - 1) * (0.9 if OFFSET else 1.0) + OFFSET
    for _ in range(2):
        Fh.knn_graph(x, B, N, k, need_reverse=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        Fh.knn_graph(x, B, N, k, need_reverse=False)
    e1.record(); torch.cuda.synchronize()
    print("B=%d N=%d C=%d k=%d: %.1f us" % (B, N, C, k, e0.elapsed_time(e1) / 5 * 1e3))
