"""In-kernel phase stamps of tnet_edge_bwds_kernel (diagnostic build: hipcc -DTB_STAMPS of tnet.hip linked against the shipped objects, built
on the GPU box; the stamps overwrite the G1 partial, the gradients of this run are wrong by design).   python tools/tb_stamps.py [B N k]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "mlsp_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DTB_STAMPS", "-c",
                       os.path.join(src, "tnet.hip"), "-o", "/tmp/tb_stamp.o"])
objs = [o for o in glob.glob(os.path.join(src, "build", "*.o")) if not o.endswith("tnet.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libtb_stamp.so"] + objs + ["/tmp/tb_stamp.o"])
os.environ["MLSP_HIP_LIB"] = "/tmp/libtb_stamp.so"
import numpy as np
import torch
from mlsp_amd import functional as Fh, _lib
B, N, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 1024, 20)
dev = torch.device("cuda:0")
x = (torch.rand(B * N, 3, device=dev) * 2 - 1)
g = Fh.knn_graph(x, B, N, k)
W1 = torch.randn(64, 6, device=dev, requires_grad=True); W2 = (torch.randn(128, 64, device=dev) / 8).requires_grad_(True)
g1, b1 = torch.ones(64, device=dev, requires_grad=True), torch.zeros(64, device=dev, requires_grad=True)
g2, b2 = torch.ones(128, device=dev, requires_grad=True), torch.zeros(128, device=dev, requires_grad=True)
rm1, rv1, rm2, rv2 = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros(128, device=dev), torch.ones(128, device=dev)
for _ in range(3):
    out = Fh.tnet_edge(x, g, W1, g1, b1, rm1, rv1, W2, g2, b2, rm2, rv2, True)
    out.backward(torch.ones_like(out))
torch.cuda.synchronize()
names = ["top (loads issued)", "gsel W2 | G1+Gram loop", "H' M", "epilogue", "barrier 1", "staging stores", "barrier 2", "-"]
for ws in _lib._workspaces.values():
    a = ws.cpu().numpy()
    a = a[: a.size // 8 * 8].view(np.uint64)
    for role, magic in (("dH wave (0)", 0x5354414d50533031), ("G1 wave (4)", 0x5354414d50533032)):
        hits = np.nonzero(a == np.uint64(magic))[0]
        if hits.size == 0:
            continue
        st = np.stack([a[h + 1:h + 9] for h in hits if h + 9 <= a.size]).astype(np.float64)
        med = np.median(st, axis=0)
        print("%s, %d workgroups, median clocks per workgroup: " % (role, len(hits)) + ", ".join("%s %d" % (n, v) for n, v in zip(names, med)),
              "| sum %.0f" % med.sum())
