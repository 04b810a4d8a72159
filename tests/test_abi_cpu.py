"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mlsp_hip.h declares with the argument counts the ctypes binding uses; host-side mirror of the
reference API (names, state_dict keys) is intact; the product refuses to run without a GPU."""
import inspect
import os
import re

import pytest
import torch

import golden_common as gc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_prototypes():
    src = open(os.path.join(ROOT, "include", "mlsp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(mlsp_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
        protos[m.group(2)] = (m.group(1), n, args)
    return protos


def test_library_exports_every_declared_symbol():
    from mlsp_amd import _lib
    lib = _lib.load()
    protos = _header_prototypes()
    assert len(protos) >= 24
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, (ret, nargs, args) in protos.items():
        assert hasattr(lib, name), name
        assert len(_lib.SIGNATURES[name]) == nargs, (name, len(_lib.SIGNATURES[name]), nargs)
        # pointer / scalar kinds line up
        kinds = ["p" if "*" in a or "mlsp_stream_t" in a else "s" for a in args.split(",")] if nargs else []
        import ctypes
        for kind, ct in zip(kinds, _lib.SIGNATURES[name]):
            assert (ct is ctypes.c_void_p) == (kind == "p"), (name, args)
    assert lib.mlsp_abi_version() == _lib.ABI_VERSION == 13
    assert b"workspace" in lib.mlsp_strerror(-2)
    assert lib.mlsp_workspace_bytes(32768, 512, 1024) > 32768 * 1024 * 4


def test_reference_api_surface_cpu():
    from mlsp_amd import Models, model_utils, mlsp
    for name in ("knn", "get_graph_feature", "conv_2d", "fc_layer", "transform_net", "classifier", "density_classifier"):
        assert hasattr(model_utils, name)
    for name in ("DGCNN", "RegionReconstruction", "Normal_prediction", "Density_prediction"):
        assert hasattr(Models, name)
    for name in ("calc_loss", "calc_scan_loss", "reconstruction_loss", "calc_normal_loss", "normal_prediction_loss",
                 "densityloss", "chamfer_distance", "DefRec_SCALER"):
        assert hasattr(mlsp, name)
    assert mlsp.DefRec_SCALER == 20.0
    sig = inspect.signature(Models.DGCNN.forward)
    assert list(sig.parameters)[1:] == ["x", "visualization", "activate_DefRec", "activate_normal", "activate_scan",
                                        "activate_density", "activate_density_normal_ondef"]
    assert list(inspect.signature(mlsp.densityloss).parameters) == ["args", "logits", "target", "target_vec", "mask"]
    args = gc.make_args(dropout=0.5)
    args.num_class = 10.0                                  # trainer.py:97 parses it as float
    m = Models.DGCNN(args)
    keys = list(m.state_dict().keys())
    assert len(keys) == 159 and keys[0] == "input_transform_net.conv2d1.conv.0.weight"
    assert sum(p.numel() for p in m.parameters()) == 4548915
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 4548899
    # Adam over model.parameters() (trainer.py:258) sees the reference's parameter order
    assert [n for n, _ in m.named_parameters()][:3] == ["input_transform_net.conv2d1.conv.0.weight",
                                                        "input_transform_net.conv2d1.conv.1.weight",
                                                        "input_transform_net.conv2d1.conv.1.bias"]


def test_no_cpu_fallback():
    from mlsp_amd import Models, model_utils, mlsp, _lib
    m = Models.DGCNN(gc.make_args())
    with pytest.raises(_lib.MlspLibraryError):
        m(torch.zeros(2, 3, 64))
    with pytest.raises(_lib.MlspLibraryError):
        model_utils.knn(torch.zeros(1, 3, 32), 20)
    with pytest.raises(_lib.MlspLibraryError):
        mlsp.reconstruction_loss(torch.zeros(1, 8, 3), torch.zeros(1, 3, 8), torch.ones(1, 3, 8))


def test_missing_library_fails_loudly(monkeypatch):
    from mlsp_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmlsp_hip.so")
    with pytest.raises(_lib.MlspLibraryError, match="no CPU fallback"):
        _lib.load()


def test_no_register_spills_in_hot_kernels():
    """The Makefile leaves per-kernel resource remarks next to the objects: no kernel of the hot path may spill VGPRs or
    use scratch (a spilling kernel runs 5-20x slower on gfx950).  No exceptions (the round-1 list-merge kNN that spilled at
    k = 40 is deleted).  SGPR spills are a different thing on gfx950 -- the compiler parks scalars in lanes of a VGPR
    (v_writelane_b32 / v_readlane_b32, no memory traffic: ScratchSize stays 0, which is asserted) -- and the tile kernels with a dozen
    buffer descriptors and pre-computed scalar offsets all have some; test_sgpr_spills_stay_out_of_the_hot_loops bounds them."""
    import glob, re
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "..", "mlsp_amd", "csrc", "build", "*.remarks")))
    if not files:
        pytest.skip("library not built here (build/*.remarks absent)")
    bad = []
    for f in files:
        txt = open(f).read()
        names = re.findall(r"Function Name: (\S+)", txt)
        spills = re.findall(r"VGPRs Spill: (\d+)", txt)
        scratch = re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", txt)
        assert len(names) == len(spills) == len(scratch)
        for n, s, c in zip(names, spills, scratch):
            if int(s) or int(c):
                bad.append((os.path.basename(f), n, int(s), int(c)))
    assert not bad, bad


# kernel family -> ceiling of "SGPRs Spill" (the figures of the round-4 build plus a margin).  A scalar spill is a v_writelane / v_readlane
# pair into a spare VGPR, never memory; what would hurt is a reload inside a K loop, so the ceilings are held where they were measured
# harmless: gemm_split_kernel's K-tile body (gemm_split_body_wm*.inc) addresses everything through loop-invariant descriptors and
# immediate offsets, and DESIGN.md section 10 lists the readlane count of the loop bodies.
SGPR_SPILL_CEILING = {"gemm_split_kernel": 64, "gemm_f32_kernel": 64, "gemm_bf16_kernel": 64, "edge_reduce_lds_kernel": 180, "edge_reduce_wide_kernel": 180,
                      "knn_mfma5_kernel": 220, "knn_mfma4_kernel": 96, "knn_kernel": 300, "knn_query_kernel": 80,
                      "knn6_kernel": 160, "knn6w_kernel": 40, "tnet_edge_fwd2_kernel": 96, "tnet_edge_fwd3_kernel": 96, "tnet_edge_bwdg_kernel": 112, "tnet_edge_bwds_kernel": 32,
                      # (the operand-transform instantiations of the streaming kernels: the XfDev argument block outlives the 104 SGPRs of a wave)
                      "thin_smalln_kernel": 32, "thin_tn_kernel": 32}


def test_sgpr_spills_stay_bounded():
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "..", "mlsp_amd", "csrc", "build", "*.remarks")))
    if not files:
        pytest.skip("library not built here (build/*.remarks absent)")
    over, seen = [], set()
    for f in files:
        txt = open(f).read()
        names = re.findall(r"Function Name: (\S+)", txt)
        spills = re.findall(r"SGPRs Spill: (\d+)", txt)
        assert len(names) == len(spills)
        for n, s in zip(names, spills):
            fam = next((k for k in SGPR_SPILL_CEILING if re.search(r"\d" + k + r"[A-Z]", n)), None)
            if int(s) and fam is None:
                over.append((n, int(s), "no ceiling recorded"))
            elif fam is not None:
                seen.add(fam)
                if int(s) > SGPR_SPILL_CEILING[fam]:
                    over.append((n, int(s), SGPR_SPILL_CEILING[fam]))
    assert not over, over
    assert {"gemm_split_kernel", "edge_reduce_lds_kernel"} <= seen


def test_forward_bookkeeping_is_per_thread():
    """nn.DataParallel runs one replica per Python thread (PointDA/trainer.py:251-252): the queue of num_batches_tracked
    increments and the forward nesting depth are thread-local, so interleaved forwards of two replicas lose no increment."""
    import threading
    from mlsp_amd import model_utils as mu

    class Rep(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.bn = torch.nn.BatchNorm1d(4)
            self.inner = Inner()

        @mu.flushing_forward
        def forward(self, barrier):
            mu._bn_buffers(self.bn, True)
            barrier.wait()                    # both threads are inside a forward, each with a queued counter
            self.inner(barrier)
            return int(self.bn.num_batches_tracked)

    class Inner(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.bn = torch.nn.BatchNorm1d(4)

        @mu.flushing_forward
        def forward(self, barrier):
            mu._bn_buffers(self.bn, True)
            barrier.wait()

    reps, barrier, errs = [Rep(), Rep()], threading.Barrier(2), []

    def run(r):
        try:
            for it in range(50):
                assert r(barrier) == it       # the nested forward must not flush; the outermost one flushes on exit
        except Exception as e:                # noqa: BLE001
            errs.append(e)
            barrier.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in reps]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for r in reps:
        assert int(r.bn.num_batches_tracked) == 50 and int(r.inner.bn.num_batches_tracked) == 50


def test_no_process_wide_dispatch_state_in_the_library():
    """the compute entry points hold no global mutable state.  The GEMM product mode is an ARGUMENT of every entry point that
    reaches the matrix cores (ABI v8): the v7 switch is neither exported nor present in the sources, and no file-scope mutable `g_*`
    variable takes part in dispatch (the only process-wide objects are the measurement hook's event tables in gemm.hip)."""
    import ctypes, glob
    from mlsp_amd import _lib
    lib = _lib.load()
    assert not hasattr(lib, "mlsp_set_gemm_precision")
    protos = _header_prototypes()
    with_prec = [n for n, (_, _, args) in protos.items() if re.search(r"\bint precision\b", args)]
    assert len(with_prec) == 21, sorted(with_prec)     # (19 compute entries + the two *_bwd_stats_parts shape queries)
    for n in ("mlsp_gemm_f32", "mlsp_pointmlp_fwd_f32", "mlsp_pointmlp_bwd_f32", "mlsp_edgeconv_fwd_f32", "mlsp_tnet_edge_fwd_f32",
              "mlsp_multimlp_fwd_f32", "mlsp_pointmlp_colmax_bwd_f32"):
        assert n in with_prec, n
    bad = []
    for f in glob.glob(os.path.join(ROOT, "mlsp_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "mlsp_amd", "csrc", "*.h")):
        for i, line in enumerate(open(f), 1):
            if re.match(r"\s*static\s+(?!const|constexpr|inline|__device__|__global__|thread_local)[^()=;]*\bg_\w+\s*(=|;)", line):
                bad.append((os.path.basename(f), i, line.strip()))
    assert not bad, bad
    # a bad mode is an argument error, not a silent default
    assert lib.mlsp_gemm_f32(0, 0, 8, 8, 8, None, 8, None, 8, None, 8, None, 7, None, 0, None) == -1


def test_gemm_precision_is_per_thread_and_saved_per_call():
    """`with gemm_precision(m)` changes what THIS thread's forwards pass; another thread keeps the process default; set() moves the default."""
    import threading
    from mlsp_amd import functional as Fh
    seen = {}
    ev_in, ev_out = threading.Event(), threading.Event()

    def other():
        ev_in.wait(10)
        seen["other"] = (Fh.gemm_precision.current, Fh.gemm_precision.code())
        ev_out.set()
    t = threading.Thread(target=other)
    t.start()
    base = Fh.gemm_precision.current
    with Fh.gemm_precision("fp32"):
        assert Fh.gemm_precision.current == "fp32" and Fh.gemm_precision.code() == 0
        with Fh.gemm_precision("bf16"):
            assert Fh.gemm_precision.code() == 1
            ev_in.set()
            ev_out.wait(10)
        assert Fh.gemm_precision.current == "fp32"
    t.join()
    assert Fh.gemm_precision.current == base
    assert seen["other"] == (base, Fh._lib.GEMM_PRECISION_MODES[base])
    Fh.gemm_precision.set("fp32")
    try:
        assert Fh.gemm_precision.current == "fp32"
    finally:
        Fh.gemm_precision.set(base)


def test_activation_storage_is_scoped_like_gemm_precision():
    """the two switches are read from the same place -- a `with` block is the calling thread's, set() the process default that
    nn.DataParallel replica threads see -- so a replica never combines one thread's precision with another thread's storage."""
    import threading
    from mlsp_amd import functional as Fh
    seen = {}

    def other():
        seen["other"] = (Fh.gemm_precision.current, Fh.activation_storage.current)
    with Fh.gemm_precision("bf16"), Fh.activation_storage("bf16"):
        assert Fh.activation_storage.current == "bf16"
        t = threading.Thread(target=other)
        t.start()
        t.join()
    assert Fh.activation_storage.current == "fp32"
    assert seen["other"] == (Fh.gemm_precision.default, "fp32")           # neither half of the main thread's block leaked
    base_p = Fh.gemm_precision.default
    Fh.gemm_precision.set("bf16"); Fh.activation_storage.set("bf16")
    try:
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert seen["other"] == ("bf16", "bf16")                          # the defaults reach every thread together
    finally:
        Fh.gemm_precision.set(base_p); Fh.activation_storage.set("fp32")
    with pytest.raises(ValueError):
        Fh.activation_storage.set("fp16")


def test_gemm_precision_default_and_env_override():
    """The Python mirror passes "f16x3" (fp32-accurate split products, two f16 pieces) unless told otherwise; MLSP_GEMM_PRECISION overrides the process default and
    an unknown name fails at import, loudly."""
    import subprocess, sys
    from mlsp_amd import _lib, functional as Fh
    assert _lib.GEMM_PRECISION_MODES == {"fp32": 0, "bf16": 1, "bf16x6": 2, "f16x3": 3}
    assert Fh.gemm_precision.current == _lib.DEFAULT_GEMM_PRECISION
    code = "from mlsp_amd import functional as Fh; print(Fh.gemm_precision.current)"
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("MLSP_GEMM_PRECISION", None)
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "f16x3"
    env["MLSP_GEMM_PRECISION"] = "fp32"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "fp32"
    env["MLSP_GEMM_PRECISION"] = "fp16"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "MLSP_GEMM_PRECISION" in r.stderr
    with pytest.raises(ValueError):
        Fh.gemm_precision("tf32")
