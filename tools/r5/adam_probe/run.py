"""bash: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC tools/r5/adam_probe/probe.hip -o gpurun_out/adam_probe.so && python tools/r5/adam_probe/run.py"""
import ctypes
import torch
lib = ctypes.CDLL("gpurun_out/adam_probe.so")
dev = torch.device("cuda:0")
n = 1 << 22
g = torch.Generator(device=dev).manual_seed(0)
p0 = torch.randn(n, device=dev, generator=g) * 0.05
gr = torch.randn(n, device=dev, generator=g) * 0.01
m0 = torch.randn(n, device=dev, generator=g) * 0.005
v0 = torch.rand(n, device=dev, generator=g) * 1e-4
lr, b1, b2, wd, eps, step = 1e-3, 0.9, 0.999, 5e-5, 1e-8, 7
pt, mt, vt = p0.clone(), m0.clone(), v0.clone()
torch._fused_adam_([pt], [gr], [mt], [vt], [], [torch.tensor(float(step), device=dev)], lr=lr, beta1=b1, beta2=b2, weight_decay=wd, eps=eps,
                   amsgrad=False, maximize=False, grad_scale=None, found_inf=None)
D = ctypes.c_double
for mode in range(32):
    p, m, v = p0.clone(), m0.clone(), v0.clone()
    rc = lib.run_probe(ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(gr.data_ptr()), ctypes.c_void_p(m.data_ptr()), ctypes.c_void_p(v.data_ptr()),
                       n, D(lr), D(b1), D(b2), D(wd), D(eps), step, mode)
    print("mode %2d rc %d  param != %7d  exp_avg != %7d  exp_avg_sq != %7d" % (mode, rc, (p != pt).sum().item(), (m != mt).sum().item(), (v != vt).sum().item()))
