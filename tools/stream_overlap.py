"""Does running the dgrad and the wgrad of one layer on two HIP streams (fork / join by events, as the library would do inside
mlsp_pointmlp_bwd_*) beat issuing them back to back on one stream?  Run on the GPU box."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mlsp_amd import _lib, functional as Fh

lib = _lib.load()
dev = torch.device("cuda:0")
P = 32768


def mk(ta, tb, M, N, K):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(1 << 27, dtype=torch.uint8, device=dev)
    def run(stream):
        rc = lib.mlsp_gemm_f32(ta, tb, M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), N, None,
                               Fh.gemm_precision.code(), ws.data_ptr(), ws.numel(), ctypes.c_void_p(stream.cuda_stream))
        assert rc == 0, rc
    return run


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


s0 = torch.cuda.current_stream()
s1 = torch.cuda.Stream()
s2 = torch.cuda.Stream()
for (Cin, Cout) in ((256, 256), (512, 256), (512, 512), (256, 128), (128, 64)):
    dgrad = mk(0, 0, P, Cin, Cout)
    wgrad = mk(1, 0, Cout, Cin, P)
    fwd = mk(0, 1, P, Cout, Cin)
    ef, ej = torch.cuda.Event(), torch.cuda.Event()

    def seq():
        dgrad(s0); wgrad(s0)

    def par():
        ef.record(s0); s1.wait_event(ef)
        wgrad(s1); ej.record(s1)
        dgrad(s0)
        s0.wait_event(ej)

    def seq3():
        fwd(s0); dgrad(s0); wgrad(s0)

    ej2 = torch.cuda.Event()
    def par3():
        ef.record(s0); s1.wait_event(ef); s2.wait_event(ef)
        wgrad(s1); ej.record(s1)
        fwd(s2); ej2.record(s2)
        dgrad(s0)
        s0.wait_event(ej); s0.wait_event(ej2)

    print("Cin=%d Cout=%d: dgrad %.1f wgrad %.1f fwd %.1f | seq %.1f  fork/join %.1f | seq3 %.1f  fork/join3 %.1f us" % (
        Cin, Cout, timed(lambda: dgrad(s0)), timed(lambda: wgrad(s0)), timed(lambda: fwd(s0)), timed(seq), timed(par), timed(seq3), timed(par3)),
        flush=True)

# a chain of small dependent kernels (finalizer-like, ~5 us each) beside a big GEMM
big = mk(0, 1, P, 1024, 512)
x = torch.zeros(4096, device=dev)
def small_chain(n=10):
    for _ in range(n):
        x.add_(1.0)
ef, ej = torch.cuda.Event(), torch.cuda.Event()
def seq_s():
    big(s0); small_chain()
def par_s():
    ef.record(s0); s1.wait_event(ef)
    with torch.cuda.stream(s1):
        small_chain()
    ej.record(s1)
    big(s0)
    s0.wait_event(ej)
print("big %.1f  small chain %.1f | seq %.1f  fork/join %.1f us" % (timed(lambda: big(s0)), timed(small_chain), timed(seq_s), timed(par_s)))
