"""Times mlsp_gemm_f32 for probe variants of gemm.hip built on the box with -D flags (kernel-design experiments)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

SRC = os.path.join(ROOT, "mlsp_amd", "csrc")
FILES = ("api.hip", "gemm.hip", "knn.hip", "bn.hip", "edge.hip", "loss.hip", "tnet.hip", "colmax.hip", "labels.hip", "skinny.hip", "sa.hip", "corrupt.hip", "thin.hip")


def build(flags, out):
    # only gemm.hip changes between variants: the other objects are compiled once
    os.makedirs("/tmp/gp", exist_ok=True)
    objs = []
    for f in FILES:
        o = "/tmp/gp/%s.o" % f if f != "gemm.hip" else out + ".gemm.o"
        if f == "gemm.hip" or not os.path.exists(o):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                                   "-c", "-o", o] + (flags if f == "gemm.hip" else []) + [os.path.join(SRC, f)])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", out] + objs)


SHAPES = [("NT", 0, 1, 32768, 256, 256), ("NN", 0, 0, 32768, 256, 256), ("NT", 0, 1, 32768, 128, 256)] if os.environ.get("GP_SHORT") else [("NT", 0, 1, 32768, 512, 512), ("NT", 0, 1, 32768, 1024, 512), ("NT", 0, 1, 32768, 256, 256), ("NT", 0, 1, 32768, 128, 256),
          ("NN", 0, 0, 32768, 512, 512), ("NN", 0, 0, 32768, 256, 256), ("TN", 1, 0, 512, 512, 32768), ("TN", 1, 0, 256, 256, 32768),
          ("NT", 0, 1, 32768, 512, 4096), ("NT", 0, 1, 8192, 8192, 1024)]


def time_lib(path):
    lib = ctypes.CDLL(path)
    P = ctypes.c_void_p
    I = ctypes.c_int
    lib.mlsp_gemm_f32.argtypes = [I, I, I, I, I, P, I, P, I, P, I, P, I, P, ctypes.c_size_t, P]
    dev = torch.device("cuda:0")
    ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for name, ta, tb, M, N, K in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        B = torch.randn((N, K) if tb else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        def run():
            rc = lib.mlsp_gemm_f32(ta, tb, M, N, K, A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], C.data_ptr(), N, None,
                                   int(os.environ.get('GEMM_PRECISION_CODE', '2')), ws.data_ptr(), ws.numel(), st)
            assert rc == 0, rc
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        extra = ""
        if os.environ.get("GP_TIMELINE") and not ta:
            torch.cuda.synchronize()
            bm = 64 if (M // 128) * ((N + 127) // 128) < 1536 and M >= 256 else 128
            d = C.view(M, N)[0::bm, 0::128, ].reshape(-1)            # first element of every tile row 0
            T = C.view(M // bm, bm, N // 128, 128)[:, 0, :, :4].reshape(-1, 4).double().cpu()
            t_start = T[:, 0]
            T = C.view(M // bm, bm, N // 128, 128)[:, 0, :, :6].reshape(-1, 6).double().cpu()
            t_start = T[:, 0]
            dur = T[:, 1] + T[:, 2] + T[:, 3]
            if os.environ.get("GP_PERCU"):
                hw = T[:, 4].long(); xcc = T[:, 5].long()
                cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
                key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
                t0 = t_start.min()
                print("  %s %dx%dx%d: %d distinct CUs" % (name, M, N, K, key.unique().numel()))
                for x in range(8):
                    m = xcc == x
                    if m.sum() == 0: continue
                    print("   xcc %d: blocks %d  CUs %d  loop mean %.1f min %.1f max %.1f  last end %.1f us" % (
                        x, int(m.sum()), key[m].unique().numel(), T[m, 2].mean() / 100, T[m, 2].min() / 100, T[m, 2].max() / 100,
                        ((t_start + dur)[m].max() - t0) / 100))
                cnt = torch.bincount(key - key.min())
                cnt = cnt[cnt > 0]
                print("   blocks per CU: min %d max %d ; histogram %s" % (int(cnt.min()), int(cnt.max()), torch.bincount(cnt).tolist()))
                # the blocks of the first few CUs: end of loop / end of block, relative to the kernel start
                for k in key.unique()[:6].tolist():
                    m = key == k
                    print("   cu %d: start %s loop_end %s end %s" % (k, ((t_start[m] - t0) / 100).tolist(), ((t_start + T[:, 1] + T[:, 2] - t0)[m] / 100).tolist(),
                                                                    ((t_start + dur - t0)[m] / 100).tolist()))
            end = t_start + dur
            extra = " | per-block (us): prologue %.1f loop %.1f (min %.1f max %.1f) epilogue %.1f (max %.1f); block total mean %.1f max %.1f; starts span %.0f us, kernel span (first start .. last end) %.1f us" % (
                T[:, 1].mean() / 100, T[:, 2].mean() / 100, T[:, 2].min() / 100, T[:, 2].max() / 100, T[:, 3].mean() / 100, T[:, 3].max() / 100,
                dur.mean() / 100, dur.max() / 100, (t_start.max() - t_start.min()) / 100, (end.max() - t_start.min()) / 100)
        out.append("%s %dx%dx%d %.0fus %.0fTF%s" % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, extra))
    return out


if __name__ == "__main__":
    variants = {"base": []}
    for a in sys.argv[1:]:
        name, _, fl = a.partition("=")
        variants[name] = fl.split(",") if fl else []
    for name, fl in variants.items():
        out = "/tmp/gp/lib_%s.so" % name
        build(fl, out)
        print(name, " | ".join(time_lib(out)), flush=True)
