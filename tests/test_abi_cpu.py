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
    assert lib.mlsp_abi_version() == _lib.ABI_VERSION == 7
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
    k = 40 is deleted)."""
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


def test_gemm_precision_default_and_env_override():
    """The Python mirror starts the library in "bf16x6" (fp32-accurate split products); MLSP_GEMM_PRECISION overrides the process default and
    an unknown name fails at import, loudly."""
    import subprocess, sys
    from mlsp_amd import _lib, functional as Fh
    assert _lib.GEMM_PRECISION_MODES == {"fp32": 0, "bf16": 1, "bf16x6": 2}
    assert Fh.gemm_precision.current == _lib.DEFAULT_GEMM_PRECISION
    code = "from mlsp_amd import functional as Fh; print(Fh.gemm_precision.current)"
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("MLSP_GEMM_PRECISION", None)
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "bf16x6"
    env["MLSP_GEMM_PRECISION"] = "fp32"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "fp32"
    env["MLSP_GEMM_PRECISION"] = "fp16"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "MLSP_GEMM_PRECISION" in r.stderr
    with pytest.raises(ValueError):
        Fh.gemm_precision("tf32")
