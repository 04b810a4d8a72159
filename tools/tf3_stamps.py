"""In-kernel phase stamps of tnet_edge_fwd3_kernel (diagnostic build: hipcc -DTF3_STAMPS of tnet.hip linked against the shipped objects, built
on the GPU box; the stamps overwrite the BN2 partials, the results of this run are wrong by design).   python tools/tf3_stamps.py [B N k]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "mlsp_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DTF3_STAMPS", "-c",
                       os.path.join(src, "tnet.hip"), "-o", "/tmp/tf3_stamp.o"])
objs = [o for o in glob.glob(os.path.join(src, "build", "*.o")) if not o.endswith("tnet.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libtf3_stamp.so"] + objs + ["/tmp/tf3_stamp.o"])
os.environ["MLSP_HIP_LIB"] = "/tmp/libtf3_stamp.so"
import numpy as np
import torch
from mlsp_amd import functional as Fh, _lib
B, N, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 1024, 20)
dev = torch.device("cuda:0")
x = (torch.rand(B * N, 3, device=dev) * 2 - 1)
g = Fh.knn_graph(x, B, N, k)
W1 = torch.randn(64, 6, device=dev); W2 = torch.randn(128, 64, device=dev) / 8
g1, b1, g2, b2 = torch.ones(64, device=dev), torch.zeros(64, device=dev), torch.ones(128, device=dev), torch.zeros(128, device=dev)
rm1, rv1, rm2, rv2 = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros(128, device=dev), torch.ones(128, device=dev)
with torch.no_grad():
    for _ in range(3):
        Fh.tnet_edge(x, g, W1, g1, b1, rm1, rv1, W2, g2, b2, rm2, rv2, True)
torch.cuda.synchronize()
names = ["gather + next indices issued", "barrier (H free)", "centre rows + barrier", "convert + image writes", "barrier (image ready)",
         "products + riding epilogue", "last block's epilogue + outputs", "-"]
for ws in _lib._workspaces.values():
    a = ws.cpu().numpy()
    a = a[: a.size // 8 * 8].view(np.uint64)
    hits = np.nonzero(a == np.uint64(0x5446335354414d50))[0]
    if hits.size:
        st = np.stack([a[h + 1:h + 9] for h in hits if h + 9 <= a.size]).astype(np.float64)
        med = np.median(st, axis=0)
        print("wave 0 of %d workgroups, median clocks per workgroup: " % len(hits) + ", ".join("%s %d" % (n, v) for n, v in zip(names, med)), "| sum %.0f" % med.sum())
