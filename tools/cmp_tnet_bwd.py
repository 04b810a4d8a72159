import os, sys, subprocess, torch
sys.path.insert(0, '/root/repo')
# run in two processes (env switch is read once per process)
code = r'''
import os, sys, torch
sys.path.insert(0, "/root/repo")
from mlsp_amd import functional as Fh
dev = torch.device("cuda:0"); torch.manual_seed(0)
B, N, k = 32, 1024, 20
x = (torch.rand(B * N, 3, device=dev) * 2 - 1).requires_grad_(True)
g = Fh.knn_graph(x, B, N, k)
W1 = torch.randn(64, 6, device=dev, requires_grad=True); W2 = (torch.randn(128, 64, device=dev) / 8).requires_grad_(True)
g1, b1 = (torch.rand(64, device=dev) + 0.5).requires_grad_(True), torch.randn(64, device=dev, requires_grad=True)
g2, b2 = (torch.rand(128, device=dev) + 0.5).requires_grad_(True), torch.randn(128, device=dev, requires_grad=True)
rm1, rv1, rm2, rv2 = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros(128, device=dev), torch.ones(128, device=dev)
out = Fh.tnet_edge(x, g, W1, g1, b1, rm1, rv1, W2, g2, b2, rm2, rv2, True)
w = torch.randn_like(out)
gr = torch.autograd.grad((out * w).sum(), [x, W1, g1, b1, W2, g2, b2])
torch.save([t.cpu() for t in gr], sys.argv[1])
'''
open('/tmp/one.py', 'w').write(code)
subprocess.check_call([sys.executable, '/tmp/one.py', '/tmp/new.pt'])
subprocess.check_call([sys.executable, '/tmp/one.py', '/tmp/old.pt'], env=dict(os.environ, MLSP_TNET_BWD_OLD='1'))
a, b = torch.load('/tmp/new.pt'), torch.load('/tmp/old.pt')
for n, p, q in zip(['dx', 'dW1', 'dg1', 'db1', 'dW2', 'dg2', 'db2'], a, b):
    print(n, 'rel max err %.3e' % ((p - q).abs().max() / q.abs().max()).item())
