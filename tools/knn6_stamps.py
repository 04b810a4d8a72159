"""In-kernel phase stamps of knn6_kernel (diagnostic build: hipcc -DK6_STAMP of knn6.hip linked against the shipped objects, built on the
GPU box): median shader cycles per phase over the first 64 workgroups x 4 waves.   python tools/knn6_stamps.py"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

src = os.path.join(ROOT, "mlsp_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DK6_STAMP", "-c",
                       os.path.join(src, "knn6.hip"), "-o", "/tmp/k6_stamp.o"])
objs = [o for o in glob.glob(os.path.join(src, "build", "*.o")) if not o.endswith("knn6.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libk6_stamp.so"] + objs + ["/tmp/k6_stamp.o"])
lib = ctypes.CDLL("/tmp/libk6_stamp.so")
P = ctypes.c_void_p
lib.mlsp_knn_f32.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, P, P, ctypes.c_size_t, P]
dev = torch.device("cuda:0")
names = ["prologue", "pass A sweep", "xch+barrier", "tau", "pass B sweep", "ovf barrier", "F0+barrier", "F1+F2", "(F2 inside)", "items", "flushes", "total"]
for (B, N, C, k) in [(32, 1024, 3, 20), (32, 1024, 64, 20), (32, 1024, 128, 20)]:
    x = torch.rand(B * N, C, device=dev) * 2 - 1
    idx = torch.zeros(B * N, k, dtype=torch.int32, device=dev)
    ws = torch.empty(1 << 27, dtype=torch.uint8, device=dev)
    for _ in range(3):
        rc = lib.mlsp_knn_f32(x.data_ptr(), C, B, N, C, k, idx.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None)
        assert rc == 0, rc
    torch.cuda.synchronize()
    # workspace layout of mlsp_knn_f32 (api.hip): xx [B*N] floats (256-byte aligned bump allocator), then the fragment image, then the stamps
    CT = 16 if C <= 16 else 64 if C <= 64 else 128
    off = ((B * N * 4 + 255) // 256) * 256 + B * N * CT * 4 + B * N * 4
    st = ws[off:off + 32 * 8 * 16 * 4].view(torch.int32).view(256, 16).cpu().numpy()[:, :12]
    med = np.median(st, axis=0)
    print("C=%d:" % C, ", ".join("%s %d" % (n, v) for n, v in zip(names, med)), "| total us at 2.4 GHz: %.1f" % (med[11] / 2400.0))
