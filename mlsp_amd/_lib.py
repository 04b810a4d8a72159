"""ctypes binding of libmlsp_hip.so (C ABI: include/mlsp_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a tensor is not on a GPU,
every op raises.  Build the library with `python -c "import __graft_entry__ as g; g.build()"`
(or `make -C mlsp_amd/csrc`).
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLSP_HIP_LIB") or os.path.join(_HERE, "libmlsp_hip.so")   # env override: A/B kernel builds
ABI_VERSION = 13

_c = ctypes
_P, _I, _F, _U64, _SZ = _c.c_void_p, _c.c_int, _c.c_float, _c.c_uint64, _c.c_size_t
_D, _I64 = _c.c_double, _c.c_int64

# name -> argtypes (restype is int unless listed in _RESTYPE)
SIGNATURES = {
    "mlsp_abi_version": [],
    "mlsp_adam_flat_f32": [_P, _P, _P, _P, _P, _P, _I, _D, _D, _D, _D, _D, _I64, _P, _P, _P],
    "mlsp_strerror": [_I],
    "mlsp_workspace_bytes": [_I, _I, _I],
    "mlsp_knn_f32": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _SZ, _P],
    "mlsp_knn_reverse": [_P, _I, _I, _I, _P, _P, _P],
    "mlsp_graph_feature_fwd_f32": [_P, _P, _I, _I, _I, _I, _P, _P],
    "mlsp_graph_feature_bwd_f32": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "mlsp_edgeconv_fwd_f32": [_P, _I, _P, _P, _P, _P, _P, _P, _F, _F, _I, _F, _I, _I, _I, _I, _I, _I,
                              _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _SZ, _P],
    "mlsp_edgeconv_bwd_f32": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _F, _I, _I, _I, _I, _I, _I,
                              _P, _I, _I, _P, _P, _P, _I, _P, _SZ, _P],
    "mlsp_tnet_edge_fwd_f32": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _I, _I, _I, _I, _I, _I, _I,
                               _P, _P, _P, _P, _P, _P, _P, _I, _P, _SZ, _P],
    "mlsp_tnet_edge_bwd_f32": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _I,
                               _P, _P, _P, _P, _P, _P, _P, _I, _P, _SZ, _P],
    "mlsp_pointmlp_fwd_f32": [_P, _I, _I, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _F, _F, _I, _I, _F, _F, _U64,
                              _P, _P, _P, _P, _I, _P, _SZ, _P],
    "mlsp_pointmlp_bwd_f32": [_P, _P, _I, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _F, _F, _U64, _I, _I,
                              _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _SZ, _P],
    "mlsp_pointmlp_bwd_stats_parts": [_I, _I, _I, _I, _I, _I],
    "mlsp_multimlp_bwd_stats_parts": [_I, _P, _I, _I, _I, _I],
    "mlsp_pointmlp_fwd_chain_f32": [_P, _I, _P, _I, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _F, _F, _I, _I, _F, _F, _U64,
                                    _P, _P, _P, _P, _I, _P, _SZ, _P],
    "mlsp_pointmlp_bwd_chain_f32": [_P, _P, _I, _P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _F, _F, _U64, _I, _I,
                                    _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _SZ, _P],
    "mlsp_pointmlp_mx_supported": [_I, _I, _I, _I, _I, _I, _I],
    "mlsp_pointmlp_fwd_mx": [_P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _F, _F, _I, _I, _F, _F, _U64,
                             _P, _P, _I, _P, _I, _P, _SZ, _P],
    "mlsp_pointmlp_bwd_mx": [_P, _P, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P, _I, _I, _F, _F, _U64, _I, _I,
                             _P, _I, _I, _P, _P, _P, _P, _P, _I, _P, _SZ, _P],
    "mlsp_pointmlp_colmax_fwd_f32": [_P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _F, _F, _I, _I, _F, _P, _P, _P, _P, _I, _P, _SZ, _P],
    "mlsp_pointmlp_colmax_bwd_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _F, _P, _I, _P, _P, _P, _I, _P, _SZ, _P],
    "mlsp_segmax_fwd_f32": [_P, _I, _I, _I, _P, _P, _P],
    "mlsp_segmax_bwd_f32": [_P, _P, _I, _I, _I, _P, _P],
    "mlsp_colmax_fwd_f32": [_P, _I, _I, _I, _P, _P, _P],
    "mlsp_colmax_bwd_f32": [_P, _P, _I, _I, _I, _P, _P],
    "mlsp_chamfer_masked_fwd_f32": [_P, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P],
    "mlsp_chamfer_masked_bwd_f32": [_P, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P],
    "mlsp_chamfer_dir_fwd_f32": [_P, _P, _P, _I, _I, _P, _P, _P, _P],
    "mlsp_chamfer_dir_bwd_f32": [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "mlsp_normal_loss_fwd_f32": [_P, _P, _P, _I, _F, _P, _P, _SZ, _P],
    "mlsp_normal_loss_bwd_f32": [_P, _P, _P, _I, _F, _P, _P, _P, _P],
    "mlsp_density_tail_fwd_f32": [_P, _P, _I, _I, _P, _P, _P],
    "mlsp_density_tail_bwd_f32": [_P, _P, _P, _P, _I, _I, _P, _P],
    "mlsp_density_loss_fwd_f32": [_P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _SZ, _P],
    "mlsp_density_loss_bwd_f32": [_P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P],
    "mlsp_radius_count_f32": [_P, _I, _I, _I, _F, _I, _P, _P],
    "mlsp_knn_normals_f32": [_P, _I, _P, _I, _I, _I, _P, _P],
    "mlsp_knn_query_f32": [_P, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "mlsp_interp3_fwd_f32": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "mlsp_interp3_bwd_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "mlsp_fps_f32": [_P, _I, _I, _I, _I, _P, _P, _P],
    "mlsp_ball_query_f32": [_P, _I, _P, _I, _I, _I, _I, _F, _I, _P, _P],
    "mlsp_group_reverse": [_P, _I, _I, _I, _I, _P, _P, _P],
    "mlsp_sa_group_fwd_f32": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P],
    "mlsp_sa_group_bwd_f32": [_P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P],
    "mlsp_pointmlp_segmax_fwd_f32": [_P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _F, _F, _I, _I, _F, _I, _P, _P, _P, _P, _P, _I, _P, _SZ, _P],
    "mlsp_pointmlp_segmax_bwd_f32": [_P, _P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _F, _I, _P, _I, _P, _P, _P, _P, _I, _P, _SZ, _P],
    "mlsp_sa_fold_fwd_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _SZ, _P],
    "mlsp_sa_fold_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _SZ, _P],
    "mlsp_group_reverse_compact": [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "mlsp_region_assign_f32": [_P, _I, _I, _I, _P, _I, _F, _P, _P],
    "mlsp_deform_regions_f32": [_P, _I, _I, _I, _P, _P, _I, _P, _P, _I, _I, _P, _P],
    "mlsp_transform3_fwd_f32": [_P, _P, _I, _I, _P, _P],
    "mlsp_transform3_bwd_f32": [_P, _P, _P, _I, _I, _P, _P, _P],
    "mlsp_compose_linear_fwd_f32": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "mlsp_compose_linear_bwd_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "mlsp_collapse_to_point_f32": [_P, _I, _I, _P, _P, _P, _F, _I, _P, _P, _P],
    "mlsp_scan_select_f32": [_P, _I, _I, _I, _P, _I, _P, _P, _P],
    "mlsp_multimlp_supported": [_I, _P, _I, _I],
    "mlsp_multimlp_fwd_f32": [_P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _F, _F, _I, _P, _F, _U64, _P, _P, _P, _I, _P, _SZ, _P],
    "mlsp_multimlp_bwd_f32": [_P, _P, _I, _I, _P, _I, _P, _P, _P, _I, _P, _F, _U64, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _SZ, _P],
    "mlsp_profile_begin": [],
    "mlsp_profile_end": [_P],
    "mlsp_profile_classes": [_P, _I],
    "mlsp_profile_split_kinds": [_P],
    "mlsp_gemm_f32": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _SZ, _P],
    "mlsp_operand_bounds_next": [_P, _I],
}


class Seg(_c.Structure):
    """mlsp_seg_t of include/mlsp_hip.h"""
    _fields_ = [("W", _P), ("bias", _P), ("ldw", _I), ("x_col", _I), ("Cin", _I), ("Cout", _I)]


class Bound(_c.Structure):
    """mlsp_bound_t of include/mlsp_hip.h: caller-owned partial maxima of one GEMM operand (functional.OperandBounds)"""
    _fields_ = [("ptr", _P), ("rows", _c.c_long), ("cols", _I), ("ld", _I), ("partials", _P), ("valid", _I), ("n", _I)]


# Sources of ready-made bounds for weight operands (objects with .weight_bounds(tensor) -> (device pointer, n) | None): mlsp_amd.optim.FlatAdam
# registers itself -- its step kernel leaves the per-tile maxima of the parameters it just updated (functional._weight_bounds asks them).
import weakref as _weakref
weight_bound_providers = _weakref.WeakSet()


class Defer(_c.Structure):
    """mlsp_defer_t of include/mlsp_hip.h: the producer of a chained layer's input (its pre-BatchNorm matrix is what X points into)"""
    _fields_ = [("bn_save", _P), ("ld", _I), ("col", _I), ("act", _I), ("slope", _F), ("p_drop", _F), ("seed", _U64)]


_RESTYPE = {"mlsp_strerror": _c.c_char_p, "mlsp_workspace_bytes": _SZ}

_lib = None


class MlspLibraryError(RuntimeError):
    pass


# GEMM products (include/mlsp_hip.h `precision`, a per-call argument; functional.gemm_precision chooses what the mirror passes).  Default "f16x3"
# (round 6): fp32-accurate products on the f16 matrix cores -- every operand value, scaled by a per-workgroup power of two, split into two
# f16 pieces, three piece products, fp32 accumulation; "bf16x6": the same on the bf16 cores with three pieces / six products (no scale
# needed; the round 3-5 default).  Both measure below the f32-MFMA chain's error against float64.  MLSP_GEMM_PRECISION=fp32 selects the f32
# MFMA for every launch; an unknown name fails loudly at import.
GEMM_PRECISION_MODES = {"fp32": 0, "bf16": 1, "bf16x6": 2, "f16x3": 3}
DEFAULT_GEMM_PRECISION = os.environ.get("MLSP_GEMM_PRECISION", "f16x3")
if DEFAULT_GEMM_PRECISION not in GEMM_PRECISION_MODES:
    raise ValueError("MLSP_GEMM_PRECISION=%r: expected one of %s" % (DEFAULT_GEMM_PRECISION, sorted(GEMM_PRECISION_MODES)))


def load():
    """Load libmlsp_hip.so and bind every symbol of include/mlsp_hip.h.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MlspLibraryError(
            "mlsp_amd: HIP library %s not found -- build it with `make -C mlsp_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`.  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)         # AttributeError if a declared symbol is not exported
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, _I)
        if _SZ in args and _P in args:
            # An entry point that takes the caller's workspace enqueues SEVERAL kernels that hand data to each other through it.  ctypes
            # drops the GIL inside the call, so two Python threads driving one device (nn.DataParallel's replica threads on one GPU,
            # torch's parallel_apply) could interleave their launches on the shared stream and overwrite each other's scratch data:
            # one call at a time per process (an uncontended lock costs ~0.1 us; the launches themselves stay asynchronous).
            setattr(lib, name, _serialised(fn))
    if lib.mlsp_abi_version() != ABI_VERSION:
        raise MlspLibraryError("mlsp_amd: ABI mismatch: library %d, python %d" % (lib.mlsp_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


_call_lock = threading.Lock()


def _serialised(fn):
    lock = _call_lock

    def call(*a):
        with lock:
            return fn(*a)
    call.__name__ = getattr(fn, "__name__", "mlsp_entry")
    return call


def check(rc, what):
    if rc != 0:
        msg = load().mlsp_strerror(rc)
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    """raw handle of torch's current HIP stream on the current device (the C-level getter: torch.cuda.current_stream() builds a Python
    Stream object per call, ~10 us -- sixty of them per step were a tenth of the enqueue time)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MlspLibraryError("mlsp_amd: tensor on %s -- the MI355X path needs GPU tensors; there is no CPU "
                                   "fallback (the CPU restatement lives in oracle/ and is test-only)" % t.device)


# one scratch buffer per (device, stream), grown on demand (the launches of one stream are ordered; two streams must not share one)
_workspaces = {}


_ws_need = {}


def workspace(device, rows, cin, cout):
    """(pointer, bytes) of this (device, stream)'s scratch buffer, grown to what mlsp_workspace_bytes asks for the shape (memoised: a pure
    function of the three sizes)."""
    shape = (int(rows), int(cin), int(cout))
    need = _ws_need.get(shape)
    if need is None:
        need = _ws_need[shape] = load().mlsp_workspace_bytes(*shape)
    key = (device.index if device.index is not None else torch._C._cuda_getDevice(), stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(int(need), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws.data_ptr(), ws.numel()


def release_workspaces():
    _workspaces.clear()
