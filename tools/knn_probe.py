"""Times the kNN kernel (and probe variants built on the box with -D flags) at the BASELINE shape."""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

def build(flags, out):
    src = os.path.join(ROOT, "mlsp_amd", "csrc")
    objs = []
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", out] + flags + \
          [os.path.join(src, f) for f in ("api.hip", "gemm.hip", "knn.hip", "bn.hip", "edge.hip", "loss.hip", "tnet.hip", "colmax.hip", "labels.hip", "skinny.hip", "sa.hip", "corrupt.hip", "thin.hip")]
    subprocess.check_call(cmd)

def time_lib(path, C, B=32, N=1024, k=20, reps=10):
    lib = ctypes.CDLL(path)
    P = ctypes.c_void_p
    lib.mlsp_knn_f32.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, P, P, ctypes.c_size_t, P]
    dev = torch.device("cuda:0")
    x = torch.rand(B * N, C, device=dev) * 2 - 1
    idx = torch.empty(B * N, k, dtype=torch.int32, device=dev)
    ws = torch.empty(1 << 24, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        rc = lib.mlsp_knn_f32(x.data_ptr(), C, B, N, C, k, idx.data_ptr(), None, None, ws.data_ptr(), ws.numel(), st)
        assert rc == 0, rc
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

if __name__ == "__main__":
    variants = {"v5": [], "sweepA": ["-DKNN5_PROBE=4"], "passA": ["-DKNN5_PROBE=1"],
                "v5 fp32 pass A": ["-DKNN5_NO_BF16A"], "sweepA fp32": ["-DKNN5_PROBE=4", "-DKNN5_NO_BF16A"]}
    for name, fl in variants.items():
        out = "/tmp/libknn_%s.so" % name
        build(fl, out)
        print(name, {C: "%.1f us" % time_lib(out, C) for C in (3, 64, 128)})
