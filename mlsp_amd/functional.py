"""torch.autograd bindings of the HIP hot path (one Function per C-ABI block of include/mlsp_hip.h).

Layout convention inside this file: activations are POINT-major [rows, C] fp32 (rows = B*N points
or B*N*k edges).  The module layer (Models.py / model_utils.py) converts from and to the
reference's channel-major [B, C, N] at its boundary only.
"""
import itertools
import os
import threading

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
_seed_counter = itertools.count(1)


def _stream_id():
    """What distinguishes this process' dropout stream from its peers': the data-parallel rank (the nn.DataParallel
    replicas this replaces drew independent masks) -- ranks are seeded identically (bench.py), so without it
    every rank would drop the same elements."""
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _next_seed():
    # counter-based dropout stream: (torch seed, rank, call counter) -> 64-bit key hashed per element in-kernel
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + next(_seed_counter) * 0xD1B54A32D192ED03
            + _stream_id() * 0xA24BAED4963EE407) & 0xFFFFFFFFFFFFFFFF


def _rows(t, allow_bf16=False):
    """[rows, C] fp32 (or, where the op takes bf16 activation storage, bf16) matrix with unit channel stride (row pitch may exceed C)."""
    assert t.dim() == 2 and (t.dtype == torch.float32 or (allow_bf16 and t.dtype == torch.bfloat16)), (t.shape, t.dtype)
    if t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


_SHARE_BOUNDS = os.environ.get("MLSP_NO_OPERAND_BOUNDS") is None       # read-once A/B switch: off = every C call measures what it needs


class OperandBounds:
    """Caller-owned bound of ONE tensor's magnitude for the two-piece f16 products (include/mlsp_hip.h mlsp_bound_t): 256 partial
    maxima on the tensor's device, measured by the first C call that needs them (one streaming launch) and reused by every later call
    that is offered the same object -- the layers that read one activation (conv5 and the heads read the concatenated encoder features
    four times per step), a layer's backward that reads its weight again.  Valid for as long as the tensor's CONTENTS do not change:
    create one per forward for an activation, one per forward call for a weight (the optimizer changes it between steps)."""
    __slots__ = ("buf", "valid")

    def __init__(self, device):
        self.buf, self.valid = torch.empty(256, dtype=torch.float32, device=device), False


class SliceBounds:
    """`n` floats at element `offset` of a caller-owned device buffer: either ROOM for a bound that a C call leaves of one of its outputs
    (valid False: the EdgeConv layers fill their column slices of the concatenation's bound vector with |gamma| sqrt(P k) + |beta| per
    channel), or -- once filled -- the bound itself, for the layers that read the tensor (a run of slices is again a SliceBounds)."""
    __slots__ = ("buf", "offset", "n", "valid")

    def __init__(self, buf, offset, n, valid=False):
        self.buf, self.offset, self.n, self.valid = buf, int(offset), int(n), bool(valid)

    @property
    def ptr(self):
        return self.buf.data_ptr() + 4 * self.offset


class ProvidedBounds:
    """bounds that already exist somewhere (n partial maxima at a device pointer): what _lib.weight_bound_providers hand out"""
    __slots__ = ("ptr", "n", "valid", "keep")

    def __init__(self, ptr, n, keep=None):
        self.ptr, self.n, self.valid, self.keep = int(ptr), int(n), True, keep


def _weight_bounds(W, device):
    """bounds object for a weight operand of the next C call: ready-made ones when a registered provider covers W (FlatAdam: the maxima
    its step kernel left), else a fresh OperandBounds that the call measures into"""
    if _SHARE_BOUNDS:
        for prov in _lib.weight_bound_providers:
            r = prov.weight_bounds(W)
            if r is not None:
                return ProvidedBounds(r[0], r[1], prov)
    return OperandBounds(device)


class _offer_bounds:
    """`with _offer_bounds((tensor, bounds), ...):` around ONE C call: hands the call the bounds of those of its 2-D fp32 operands the
    caller keeps bounds for (None entries are skipped; nothing happens outside mode "f16x3"), and records which ones the call measured."""
    __slots__ = ("pairs", "tab")

    def __init__(self, prec, *pairs):
        self.pairs = ([(t, b) for t, b in pairs if b is not None and t is not None and t.dim() == 2 and t.dtype == torch.float32]
                      if prec == 3 and _SHARE_BOUNDS else [])
        self.tab = None

    def __enter__(self):
        if self.pairs:
            self.tab = (_lib.Bound * len(self.pairs))()
            for e, (t, b) in zip(self.tab, self.pairs):
                if isinstance(b, SliceBounds):
                    e.ptr, e.rows, e.cols, e.ld, e.partials, e.valid, e.n = t.data_ptr(), t.shape[0], t.shape[1], t.stride(0), b.ptr, int(b.valid), b.n
                elif isinstance(b, ProvidedBounds):
                    e.ptr, e.rows, e.cols, e.ld, e.partials, e.valid, e.n = t.data_ptr(), t.shape[0], t.shape[1], t.stride(0), b.ptr, 1, b.n
                else:
                    e.ptr, e.rows, e.cols, e.ld, e.partials, e.valid, e.n = t.data_ptr(), t.shape[0], t.shape[1], t.stride(0), b.buf.data_ptr(), int(b.valid), 0
            _lib.load().mlsp_operand_bounds_next(self.tab, len(self.pairs))
        return self

    def __exit__(self, *exc):
        if self.tab is not None:
            _lib.load().mlsp_operand_bounds_next(None, 0)          # (a call that failed before its precision scope leaves nothing behind)
            for e, (_, b) in zip(self.tab, self.pairs):
                if not isinstance(b, ProvidedBounds):
                    b.valid = bool(e.valid)
        return False


class _PrecisionMeta(type):
    @property
    def current(cls):
        """name of the mode the calling thread's next forward will pass to the library"""
        return getattr(cls._tls, "mode", None) or cls.default


class gemm_precision(metaclass=_PrecisionMeta):
    """Which products the GEMM family computes -- the `precision` ARGUMENT of every C entry point that reaches the matrix cores
    (include/mlsp_hip.h; the library itself keeps no switch).  "bf16x6" (default): fp32-ACCURATE products on the bf16 matrix cores --
    every operand split exactly into three bf16 pieces, six piece products per multiply, fp32 accumulation; against float64 its error
    is below the f32-MFMA chain's, and it is 1.55-1.65x faster (DVFS-limited either way: DESIGN.md).  "fp32": the f32 MFMA for every
    launch (exact fp32 products).  "bf16": operands ROUNDED to bf16, fp32 accumulation (BASELINE.json configs[4]; reduced precision,
    opt-in).  kNN distances, BatchNorm statistics, reductions and losses are fp32 in every mode (the kNN always exact, canonical order).

    `with gemm_precision(mode):` applies to the forwards the CALLING THREAD runs inside the block; every autograd Function stores the
    mode of its forward and hands the same one to its backward, whenever and on whichever thread that runs.  `gemm_precision.set(mode)`
    changes the process default (threads without a `with` block: nn.DataParallel replicas)."""
    _MODES = _lib.GEMM_PRECISION_MODES
    default = _lib.DEFAULT_GEMM_PRECISION
    _tls = threading.local()

    def __init__(self, mode):
        if mode not in self._MODES:
            raise ValueError("precision must be 'fp32', 'bf16', 'bf16x6' or 'f16x3'")
        self.mode, self.prev = mode, None

    @classmethod
    def set(cls, mode):
        if mode not in cls._MODES:
            raise ValueError("precision must be 'fp32', 'bf16', 'bf16x6' or 'f16x3'")
        cls.default = mode

    @classmethod
    def code(cls):
        """the integer the C ABI takes (MLSP_PREC_*) for the calling thread's current mode"""
        return cls._MODES[cls.current]

    def __enter__(self):
        self.prev = getattr(gemm_precision._tls, "mode", None)
        gemm_precision._tls.mode = self.mode
        return self

    def __exit__(self, *exc):
        gemm_precision._tls.mode = self.prev
        return False


class _StorageMeta(type):
    @property
    def current(cls):
        """the storage mode the calling thread's next forward uses"""
        return getattr(cls._tls, "mode", None) or cls.default


class activation_storage(metaclass=_StorageMeta):
    """Context manager / switch for how the per-point MLP stacks keep their activations in HBM: "fp32" (default; the parity contract
    of the fp32 configs) or "bf16" (BASELINE.json configs[4]: Y / Z of chained Linear+BN+act layers and their gradients are stored as
    bf16, bf16 x bf16 products with fp32 accumulation; weights, BN statistics, kNN distances, reductions, losses stay fp32).
    Layers opt in with pointmlp(..., chain=True) (their consumer is another pointmlp); shapes the bf16 kernels do not cover stay fp32.

    Scoped like gemm_precision: `with activation_storage(mode):` applies to the forwards the CALLING THREAD runs inside the block,
    `activation_storage.set(mode)` changes the process default (what nn.DataParallel replica threads see) -- the two switches are always
    read from the same place, so a replica thread never combines one thread's precision with another's storage."""
    default = "fp32"
    _tls = threading.local()

    def __init__(self, mode):
        if mode not in ("fp32", "bf16"):
            raise ValueError("storage must be 'fp32' or 'bf16'")
        self.mode, self.prev = mode, None

    @classmethod
    def set(cls, mode):
        if mode not in ("fp32", "bf16"):
            raise ValueError("storage must be 'fp32' or 'bf16'")
        cls.default = mode

    def __enter__(self):
        self.prev = getattr(activation_storage._tls, "mode", None)
        activation_storage._tls.mode = self.mode
        return self

    def __exit__(self, *exc):
        activation_storage._tls.mode = self.prev
        return False


class KnnGraph:
    """kNN indices of one graph stage plus the reverse index needed by the backward passes."""
    __slots__ = ("idx", "rev_off", "rev_ent", "B", "N", "k")

    def __init__(self, idx, rev_off, rev_ent, B, N, k):
        self.idx, self.rev_off, self.rev_ent, self.B, self.N, self.k = idx, rev_off, rev_ent, B, N, k


_forced_graphs = None   # test hook: list of index tensors consumed by successive knn_graph calls


class forced_graphs:
    """Context manager (tests only): successive knn_graph() calls return these neighbour indices
    instead of computing them, so that everything downstream of the (discontinuous) dynamic graph
    can be compared tightly against fixtures."""

    def __init__(self, idx_list):
        self.idx_list = list(idx_list)

    def __enter__(self):
        global _forced_graphs
        _forced_graphs = list(self.idx_list)
        return self

    def __exit__(self, *exc):
        global _forced_graphs
        _forced_graphs = None
        return False


_sel_forced = None      # test hook: arg-max selections consumed by successive selecting ops (forced_selections)
_sel_record = None      # test hook: list the selecting ops append their own selections to (recorded_selections)


class forced_selections:
    """Context manager (tests only), the counterpart of forced_graphs for the network's OTHER discrete choices: the arg-max of every
    max-pool (T-Net per-edge stage, T-Net / conv5 max over the points, the four EdgeConv max over k).  Successive selecting ops overwrite
    the selection they computed with the given one before saving it for the backward, so a gradient comparison against a reference no
    longer depends on which of two candidates that agree to the last bit wins (a re-routed maximum moves a whole gradient row; the
    forward VALUE differs by that last bit only).  Layouts: max over k -> [P, C] slot numbers; max over N -> [B, C] point numbers local
    to the cloud.  Call order in DGCNN: tnet_edge, its colmax, EdgeConv 1-4, conv5's colmax; in a set-abstraction stack: the fused last
    conv + neighbourhood max of every layer ([B*S, C] slot numbers)."""

    def __init__(self, sel_list):
        self.sel_list = list(sel_list)

    def __enter__(self):
        global _sel_forced
        _sel_forced = list(self.sel_list)
        return self

    def __exit__(self, *exc):
        global _sel_forced
        _sel_forced = None
        return False


class recorded_selections:
    """Context manager (tests only): `.sel` collects a copy of every selecting op's arg-max tensor, in call order."""

    def __enter__(self):
        global _sel_record
        self.sel = _sel_record = []
        return self

    def __exit__(self, *exc):
        global _sel_record
        _sel_record = None
        return False


def _selection_hook(arg):
    """`arg`: the selection a forward kernel just wrote (uint8 slots or int32 rows), before it is saved for the backward."""
    if _sel_record is not None:
        _sel_record.append(arg.detach().clone())
    if _sel_forced is not None:
        want = _sel_forced.pop(0)
        assert tuple(want.shape) == tuple(arg.shape), (want.shape, arg.shape)
        arg.copy_(want.to(device=arg.device, dtype=arg.dtype))


_identity_rows = {}


def identity_row(K, device, dtype=torch.float32):
    """flattened K x K identity [1, K*K] on `device`, built once per (K, device, dtype): the T-Nets add it to their output every
    forward (PointDA/model_utils.py:122-125 builds it with torch.eye each time: three launches)"""
    key = (K, device, dtype)
    t = _identity_rows.get(key)
    if t is None:
        t = _identity_rows[key] = torch.eye(K, device=device, dtype=dtype).view(1, K * K)
    return t


def knn_graph(xp, B, N, k, need_reverse=True):
    """xp [B*N, C] point-major (detached use only: indices are not differentiable,
    PointDA/model_utils.py:15).  Returns KnnGraph with int32 idx [B*N, k] (local indices)."""
    if _forced_graphs is not None:
        return graph_from_indices(_forced_graphs.pop(0).to(xp.device), B, N, k)
    lib = _lib.load()
    xp = _rows(xp.detach())
    _lib.require_gpu(xp)
    P, C = xp.shape
    assert P == B * N
    idx = torch.empty((P, k), dtype=torch.int32, device=xp.device)
    rev_off = rev_ent = None
    need_reverse = need_reverse and torch.is_grad_enabled()      # only a backward pass reads the reverse index
    if need_reverse:
        rev_off = torch.empty((P + 1,), dtype=torch.int32, device=xp.device)
        rev_ent = torch.empty((P * k,), dtype=torch.int32, device=xp.device)
    ws, wsn = _lib.workspace(xp.device, P, C, 1)
    _lib.check(lib.mlsp_knn_f32(xp.data_ptr(), xp.stride(0), B, N, C, k, idx.data_ptr(), _lib.ptr(rev_off),
                                _lib.ptr(rev_ent), ws, wsn, _lib.stream()), "mlsp_knn_f32")
    return KnnGraph(idx, rev_off, rev_ent, B, N, k)


def graph_from_indices(idx, B, N, k):
    """Wrap caller-provided neighbour indices [B,N,k] (any integer dtype) into a KnnGraph."""
    lib = _lib.load()
    _lib.require_gpu(idx)
    idx32 = idx.reshape(B * N, k).to(torch.int32).contiguous()
    rev_off = torch.empty((B * N + 1,), dtype=torch.int32, device=idx.device)
    rev_ent = torch.empty((B * N * k,), dtype=torch.int32, device=idx.device)
    _lib.check(lib.mlsp_knn_reverse(idx32.data_ptr(), B, N, k, rev_off.data_ptr(), rev_ent.data_ptr(), _lib.stream()),
               "mlsp_knn_reverse")
    return KnnGraph(idx32, rev_off, rev_ent, B, N, k)


class _GraphFeature(Function):
    @staticmethod
    def forward(ctx, xp, graph):
        lib = _lib.load()
        xp = xp.contiguous()
        _lib.require_gpu(xp)
        P, C = xp.shape
        F = torch.empty((P * graph.k, 2 * C), dtype=torch.float32, device=xp.device)
        _lib.check(lib.mlsp_graph_feature_fwd_f32(xp.data_ptr(), graph.idx.data_ptr(), graph.B, graph.N, C, graph.k,
                                                  F.data_ptr(), _lib.stream()), "mlsp_graph_feature_fwd_f32")
        ctx.graph, ctx.C = graph, C
        return F

    @staticmethod
    @once_differentiable
    def backward(ctx, dF):
        lib = _lib.load()
        g = ctx.graph
        dF = dF.contiguous()
        dx = torch.empty((g.B * g.N, ctx.C), dtype=torch.float32, device=dF.device)
        _lib.check(lib.mlsp_graph_feature_bwd_f32(dF.data_ptr(), g.rev_off.data_ptr(), g.rev_ent.data_ptr(), g.B, g.N,
                                                  ctx.C, g.k, dx.data_ptr(), _lib.stream()), "mlsp_graph_feature_bwd_f32")
        return dx, None


def graph_feature(xp, graph):
    """[P, C] -> edge-major [P*k, 2C] = [x_j - x_i ; x_i]  (PointDA/model_utils.py:18-42)."""
    return _GraphFeature.apply(xp, graph)


class SharedInputGrad:
    """Hand-off between the consumers of ONE activation matrix (conv5 and the three heads all read the concatenated encoder
    features): the first consumer to run backward allocates the input-gradient buffer, the others add into it through the
    beta = 1 epilogue of their dgrad GEMM and return the same tensor; `fan_out` then passes it upstream once.  Replaces three
    67 MB element-wise adds of the autograd engine per step."""
    __slots__ = ("buf", "task")

    def __init__(self):
        self.buf = None
        self.task = None

    def claim(self, shape, device):
        """-> (buffer, accumulate flag).  The buffer belongs to ONE backward pass (autograd graph task): a buffer left behind by
        a pass that never reached the fan_out node (torch.autograd.grad w.r.t. head parameters only, an exception mid-backward)
        is dropped instead of being accumulated into."""
        task = torch._C._current_graph_task_id()
        if self.buf is None or self.task != task:
            self.buf = torch.empty(shape, dtype=torch.float32, device=device)
            self.task = task
            return self.buf, 0
        return self.buf, 1


class _FanOut(Function):
    @staticmethod
    def forward(ctx, X, acc, n):
        ctx.acc = acc
        acc.buf = None
        return tuple(X.view_as(X) for _ in range(n))

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        acc, total, seen = ctx.acc, None, False
        if acc.task != torch._C._current_graph_task_id():
            acc.buf = None                      # stale: not written by this pass
        for g in grads:
            if g is None:
                continue
            if acc.buf is not None and g.data_ptr() == acc.buf.data_ptr():
                if seen:
                    continue                    # the shared buffer already holds the sum of every consumer that used it
                seen = True
            total = g if total is None else total + g
        acc.buf = acc.task = None
        return total, None, None


def fan_out(X, n):
    """n aliases of X for n consumers that support `grad_accum=` (pointmlp, pointmlp_colmax): -> (aliases, SharedInputGrad)."""
    acc = SharedInputGrad()
    return _FanOut.apply(X, acc, n), acc


class _JoinColumns(Function):
    """The [P, sum C_i] matrix whose column slices the producers have ALREADY written (edgeconv(out=...)): forward is free,
    backward hands each producer its column slice of the gradient as a strided view (no torch.cat, no split copies)."""

    @staticmethod
    def forward(ctx, base, accs, *parts):
        ctx.widths = [p.shape[1] for p in parts]
        ctx.accs = accs
        return base.view_as(base)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        outs, o = [], 0
        task = torch._C._current_graph_task_id()
        for i, w in enumerate(ctx.widths):
            sl = g[:, o:o + w]
            outs.append(sl)
            if ctx.accs is not None and ctx.accs[i] is not None:
                # the OTHER consumer of this producer's output (the next EdgeConv layer) adds its input gradient into the slice
                ctx.accs[i].buf, ctx.accs[i].task = sl, task
            o += w
        return (None, None) + tuple(outs)


def join_columns(base, parts, accs=None):
    """`accs[i]` (optional SharedInputGrad of a fan_out(parts[i], 2)): see _JoinColumns.backward."""
    return _JoinColumns.apply(base, accs, *parts)


class _SplitColumns(Function):
    """(W[:, :c], W[:, c:]) with ONE concatenation in backward (plain slicing costs a zero-fill, a copy and an add per half)."""

    @staticmethod
    def forward(ctx, W, c):
        ctx.c = c
        return W[:, :c], W[:, c:]

    @staticmethod
    @once_differentiable
    def backward(ctx, ga, gb):
        return torch.cat((ga, gb), dim=1), None


def split_columns(W, c):
    return _SplitColumns.apply(W, c)


class _RowBlocks(Function):
    """[sum n_i, ...] matrix made of the row blocks `parts` (parameters of several modules that one merged layer uses as ONE operand).
    When the blocks already lie back to back in one storage (rehome_adjacent) the result is a zero-copy view of it; otherwise one
    torch.cat.  Backward hands every part its row slice of the gradient as a view (no split copies)."""

    @staticmethod
    def forward(ctx, *parts):
        ctx.rows = [p.shape[0] for p in parts]
        if _adjacent(parts):
            p0 = parts[0]
            shape = (sum(ctx.rows),) + tuple(p0.shape[1:])
            return p0.new_empty(0).set_(p0.untyped_storage(), p0.storage_offset(), shape, p0.stride())
        return torch.cat([p.detach() for p in parts], dim=0)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        outs, o = [], 0
        for n in ctx.rows:
            outs.append(g[o:o + n])
            o += n
        return tuple(outs)


def _adjacent(parts):
    """are the (contiguous, same dtype / device) tensors laid out back to back in ONE storage, in this order?"""
    p0 = parts[0]
    if not p0.is_contiguous():
        return False
    base, end = p0.untyped_storage().data_ptr(), p0.data_ptr() + p0.numel() * p0.element_size()
    for p in parts[1:]:
        if (not p.is_contiguous() or p.dtype != p0.dtype or p.device != p0.device or p.untyped_storage().data_ptr() != base
                or p.data_ptr() != end or tuple(p.shape[1:]) != tuple(p0.shape[1:])):
            return False
        end = p.data_ptr() + p.numel() * p.element_size()
    return True


def rehome_adjacent(parts):
    """Move the storage of module parameters / buffers `parts` into ONE buffer, back to back (each keeps its identity, shape and
    values: `.data` is re-pointed at a view), so that a merged layer reads them as one operand without a concatenation per step.
    Only for genuine leaves (nn.Parameter or plain buffers outside any autograd graph); idempotent; survives optimizer updates
    (in place) and load_state_dict (copy_); after model.to() / deepcopy / DataParallel replication the tensors may be separate again
    and row_blocks() falls back to torch.cat (or this is called again)."""
    if _adjacent(parts):
        return True
    p0 = parts[0]
    for p in parts:
        if (p.grad_fn is not None or p.dtype != p0.dtype or p.device != p0.device or tuple(p.shape[1:]) != tuple(p0.shape[1:])
                or (p.requires_grad and not isinstance(p, torch.nn.Parameter))):
            return False
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(p.shape[0], -1) for p in parts], dim=0).contiguous()
        o = 0
        for p in parts:
            p.data = flat[o:o + p.shape[0]].view(p.shape)
            o += p.shape[0]
    return _adjacent(parts)


def row_blocks(parts, rehome=True):
    """One operand out of the row blocks `parts` (see _RowBlocks); `rehome`: make them adjacent first when they are leaves."""
    parts = list(parts)
    if len(parts) == 1:
        return parts[0]
    if rehome:
        rehome_adjacent(parts)
    return _RowBlocks.apply(*parts)


class merged_buffers:
    """Running statistics of several BatchNorm modules as ONE vector for a merged layer.  `.tensor` is a zero-copy view when the
    buffers are (or could be made) adjacent, else a concatenated copy that `.writeback()` scatters back after the kernel updated it."""

    def __init__(self, bufs, rehome=True):
        self.bufs = list(bufs)
        if len(self.bufs) == 1:
            self.tensor, self.copied = self.bufs[0], False
        elif (rehome_adjacent(self.bufs) if rehome else _adjacent(self.bufs)):
            b0 = self.bufs[0]
            self.tensor = b0.new_empty(0).set_(b0.untyped_storage(), b0.storage_offset(), (sum(b.shape[0] for b in self.bufs),), (1,))
            self.copied = False
        else:
            self.tensor, self.copied = torch.cat(self.bufs), True

    def writeback(self):
        if self.copied:
            with torch.no_grad():
                torch._foreach_copy_(self.bufs, list(torch.split(self.tensor, [b.shape[0] for b in self.bufs])))


class SharedColumnGrad:
    """Gradient of a [M, sum w_i] matrix whose consumers each read ONE column slice (the merged first layer of the heads feeds three
    stacks): the consumers write their input gradients straight into the column slices of one buffer (row pitch = the full width)
    and `split_columns_shared` passes that buffer upstream -- no concatenation of the slices' gradients (134 MB per step)."""
    __slots__ = ("buf", "task", "width")

    def __init__(self, width):
        self.buf, self.task, self.width = None, None, width

    def claim(self, M, col, w, device, dtype):
        task = torch._C._current_graph_task_id()
        if self.buf is None or self.task != task or self.buf.shape[0] != M or self.buf.dtype != dtype:
            self.buf = torch.empty((M, self.width), dtype=dtype, device=device)
            self.task = task
        return self.buf[:, col:col + w]


class _SplitColumnsShared(Function):
    @staticmethod
    def forward(ctx, H, acc, widths):
        ctx.acc, ctx.widths = acc, widths
        ctx.meta = (H.shape[0], H.dtype, H.device)
        outs, o = [], 0
        for w in widths:
            outs.append(H[:, o:o + w])
            o += w
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        acc = ctx.acc
        M, dtype, dev = ctx.meta
        if acc.buf is None or acc.task != torch._C._current_graph_task_id() or acc.buf.shape[0] != M or acc.buf.dtype != dtype:
            acc.buf = torch.empty((M, acc.width), dtype=dtype, device=dev)
        buf, o = acc.buf, 0
        for g, w in zip(grads, ctx.widths):
            sl = buf[:, o:o + w]
            if g is None:
                sl.zero_()
            elif g.data_ptr() != sl.data_ptr() or g.stride() != sl.stride():
                sl.copy_(g)
            o += w
        acc.buf = acc.task = None
        return buf, None, None


def split_columns_shared(H, widths):
    """column slices of H for consumers that support `grad_cols=` (pointmlp): -> (slices, SharedColumnGrad).  H may be a DeferredAct
    (its slices are DeferredActs over views of the pre-BN matrix)."""
    acc = SharedColumnGrad(sum(widths))
    if isinstance(H, DeferredAct):
        outs, subs, o = _SplitColumnsShared.apply(H.y, acc, tuple(widths)), [], 0
        for sl, w in zip(outs, widths):
            subs.append(H.sub(sl, o))
            o += w
        return tuple(subs), acc
    return _SplitColumnsShared.apply(H, acc, tuple(widths)), acc


class _MultiMLP(Function):
    """Several Linear (+bias) layers side by side under ONE BatchNorm + activation (+dropout) pass (include/mlsp_hip.h
    mlsp_multimlp_*_f32; csrc/multi.hip).  X [M, ldx] fp32; segment s reads columns x_cols[s] .. + Cin_s and writes its Cout_s columns.
    `in_defs` (tuple of DeferredAct.desc() per segment, or None): X is the previous merged layer's PRE-BatchNorm output and every
    segment's GEMM applies that layer's scale / shift / activation / dropout while it stages its operand.  `defer_out`: this layer's own
    activation pass is left to its consumers: the Function returns Y (pre-BN) in place of Z."""

    @staticmethod
    def forward(ctx, X, gamma, beta, run_mean, run_var, chan, x_cols, training, p_drop, seed, momentum, eps, in_defs, defer_out, in_stats, out_stats, *wb):
        lib = _lib.load()
        prec = ctx.prec = gemm_precision.code()     # the backward gets the same products, whenever it runs
        n = len(x_cols)
        Ws, bs = list(wb[:n]), list(wb[n:])
        X = _rows(X)
        _lib.require_gpu(X, gamma, chan, *Ws)
        Ws = [w if w.stride(1) == 1 else w.contiguous() for w in Ws]
        bs = [b.contiguous() if b is not None else None for b in bs]
        M = X.shape[0]
        Ctot = sum(w.shape[0] for w in Ws)
        dev = X.device
        segs = (_lib.Seg * n)()
        for i, (w, b, xc) in enumerate(zip(Ws, bs, x_cols)):
            segs[i].W, segs[i].bias, segs[i].ldw, segs[i].x_col, segs[i].Cin, segs[i].Cout = w.data_ptr(), _lib.ptr(b), w.stride(0), xc, w.shape[1], w.shape[0]
        defs = None
        if in_defs is not None:
            defs = (_lib.Defer * n)(*[_defer_struct(d) for d in in_defs])
        Y = torch.empty((M, Ctot), dtype=torch.float32, device=dev)
        Z = None if defer_out else torch.empty((M, Ctot), dtype=torch.float32, device=dev)
        bn_save = torch.empty((4, Ctot), dtype=torch.float32, device=dev)
        p = float(p_drop) if training else 0.0
        ws, wsn = _lib.workspace(dev, M, X.stride(0), Ctot)
        # (mode "f16x3": the weights' magnitude bounds measured by this call are read again by the backward)
        ctx.w_bounds = [_weight_bounds(w, dev) for w in Ws] if prec == 3 else []
        with _offer_bounds(prec, *zip(Ws, ctx.w_bounds)):
            _lib.check(lib.mlsp_multimlp_fwd_f32(X.data_ptr(), X.stride(0), M, segs, n, defs, gamma.data_ptr(), beta.data_ptr(), _lib.ptr(run_mean),
                                                 _lib.ptr(run_var), momentum, eps, int(training), chan.data_ptr(), p, seed, Y.data_ptr(),
                                                 _lib.ptr(Z), bn_save.data_ptr(), prec, ws, wsn, _lib.stream()), "mlsp_multimlp_fwd_f32")
        ctx.save_for_backward(X, Y, bn_save, chan, *Ws)
        ctx.cfg = (tuple(x_cols), bool(training), p, seed, [b is not None for b in bs])
        ctx.in_defs = in_defs
        ctx.in_stats, ctx.out_stats = None, (out_stats if defer_out else None)       # fused BatchNorm-backward sums: see _PointMLP.forward
        if in_stats is not None and in_defs is not None:
            parts = 0
            if _FUSE_BWD_STATS and ctx.needs_input_grad[0]:
                key = ("m", M, X.stride(0), X.shape[1], prec) + tuple((xc, w.shape[0], w.shape[1], w.stride(0), b is not None) for xc, w, b in zip(x_cols, Ws, bs))
                parts = _stats_parts(key, lambda: lib.mlsp_multimlp_bwd_stats_parts(M, segs, n, X.stride(0), X.shape[1], prec))
            for d, w in zip(in_defs, Ws):
                in_stats.promise(d[2], w.shape[1], parts)
            ctx.in_stats = in_stats
        ctx.mark_non_differentiable(bn_save)
        ctx.set_materialize_grads(False)
        return (Y.view_as(Y) if defer_out else Z), bn_save

    @staticmethod
    @once_differentiable
    def backward(ctx, dZ, _dbn=None):
        x_cols, training, p, seed, has_b = ctx.cfg
        n = len(x_cols)
        if dZ is None:
            return (None,) * (16 + 2 * n)
        lib = _lib.load()
        X, Y, bn_save, chan = ctx.saved_tensors[:4]
        Ws = ctx.saved_tensors[4:]
        dZ = dZ.contiguous()
        M, Ctot = Y.shape
        dev = dZ.device
        segs = (_lib.Seg * n)()
        for i, (w, xc) in enumerate(zip(Ws, x_cols)):
            segs[i].W, segs[i].bias, segs[i].ldw, segs[i].x_col, segs[i].Cin, segs[i].Cout = w.data_ptr(), None, w.stride(0), xc, w.shape[1], w.shape[0]
        defs = None
        if ctx.in_defs is not None:
            defs = (_lib.Defer * n)(*[_defer_struct(d) for d in ctx.in_defs])
        dX = None
        if ctx.needs_input_grad[0]:
            covered = sorted((xc, xc + w.shape[1]) for xc, w in zip(x_cols, Ws))
            full = covered[0][0] == 0 and covered[-1][1] == X.shape[1] and all(a[1] >= b[0] for a, b in zip(covered, covered[1:]))
            dX = (torch.empty if full else torch.zeros)((M, X.shape[1]), dtype=torch.float32, device=dev)
        # (one buffer, segment after segment: the weight gradients of a run of identical segments come out of one block-diagonal launch)
        dflat = torch.empty((sum(w.shape[0] * w.shape[1] for w in Ws),), dtype=torch.float32, device=dev)
        dWs, o = [], 0
        for w in Ws:
            dWs.append(dflat[o:o + w.shape[0] * w.shape[1]].view(w.shape[0], w.shape[1]))
            o += w.shape[0] * w.shape[1]
        dwp = (_lib._c.c_void_p * n)(*[d.data_ptr() for d in dWs])
        dbias = torch.empty((Ctot,), dtype=torch.float32, device=dev) if any(has_b) else None
        dgamma = torch.empty((Ctot,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((Ctot,), dtype=torch.float32, device=dev)
        ws, wsn = _lib.workspace(dev, M, X.stride(0), Ctot)
        pre = ctx.out_stats.take() if ctx.out_stats is not None else None
        pre_ptr, pre_n = (pre[0].data_ptr() if pre[0] is not None else None, pre[1]) if pre is not None else (None, 0)
        ins = None
        if ctx.in_stats is not None and dX is not None and ctx.in_stats.agreed():
            ins = ctx.in_stats.buffer(dev)
            for _ in range(n - 1):
                ctx.in_stats.buffer(dev)            # (one delivery per promised segment: this call writes all of them)
        with _offer_bounds(ctx.prec, *zip(Ws, ctx.w_bounds)):
            _lib.check(lib.mlsp_multimlp_bwd_f32(dZ.data_ptr(), X.data_ptr(), X.stride(0), M, segs, n, defs, Y.data_ptr(), bn_save.data_ptr(),
                                                 int(training), chan.data_ptr(), p, seed, _lib.ptr(dX), dX.stride(0) if dX is not None else 0,
                                                 dwp, _lib.ptr(dbias), dgamma.data_ptr(), dbeta.data_ptr(), _lib.ptr(ins), pre_ptr, pre_n, ctx.prec, ws, wsn,
                                                 _lib.stream()), "mlsp_multimlp_bwd_f32")
        dbs, o = [], 0
        for w, hb in zip(Ws, has_b):
            dbs.append(dbias[o:o + w.shape[0]] if hb else None)
            o += w.shape[0]
        return (dX, dgamma, dbeta) + (None,) * 13 + tuple(dWs) + tuple(dbs)


_chan_cache = {}


def channel_params(device, spec):
    """[2, C] device tensor of per-channel activation parameters for multimlp, built once per configuration:
    spec = ((Cout, negative-side factor, dropout on), ...) per segment."""
    key = (device, tuple(spec))
    t = _chan_cache.get(key)
    if t is None:
        sl = torch.cat([torch.full((c,), float(f)) for c, f, _ in spec])
        dr = torch.cat([torch.full((c,), 1.0 if d else 0.0) for c, _, d in spec])
        t = _chan_cache[key] = torch.stack((sl, dr)).contiguous().to(device)
    return t


def multimlp_supported(M, X, Ws, x_cols):
    if isinstance(X, DeferredAct):
        X = X.y
    if (activation_storage.current != "fp32" or gemm_precision.current == "bf16" or X.dtype != torch.float32 or X.dim() != 2
            or X.stride(1) != 1 or not X.is_cuda):
        return False
    n = len(Ws)
    segs = (_lib.Seg * n)()
    for i, (w, xc) in enumerate(zip(Ws, x_cols)):
        segs[i].W, segs[i].bias, segs[i].ldw, segs[i].x_col, segs[i].Cin, segs[i].Cout = 16, None, w.shape[1], xc, w.shape[1], w.shape[0]   # (shape query: W is not read)
    return bool(_lib.load().mlsp_multimlp_supported(int(M), segs, n, gemm_precision.code()))


def multimlp(X, segs, gamma, beta, run_mean, run_var, chan, training=True, p_drop=0.0, momentum=0.1, eps=1e-5, chain=False, spec=None):
    """segs = [(x_col, W [Cout, Cin], bias | None), ...] -> Z [M, sum Cout]: every segment's Linear on its column slice of X, then ONE
    BatchNorm (batch statistics over the M rows, all channels), per-channel activation and dropout (chan from channel_params).
    X may be a DeferredAct (the previous merged layer's pre-BN output).  `chain=True` (every consumer is a pointmlp / multimlp) with
    `spec` = the channel_params spec of THIS layer: the result is a DeferredAct, the BN + activation pass is left to the consumers."""
    x_cols = tuple(int(s[0]) for s in segs)
    Ws = [s[1] for s in segs]
    bs = [s[2] for s in segs]
    any_drop = training and p_drop > 0
    seed = _next_seed() if any_drop else 0
    in_defs = in_stats = None
    if isinstance(X, DeferredAct):
        in_defs = tuple(X.desc(xc, w.shape[1]) for xc, w in zip(x_cols, Ws))
        assert X.col == 0 and X.y.shape[1] == X.ld, "multimlp reads column slices of the producer's whole matrix"
        in_stats, X = X.stats, X.y
    defer_out = bool(chain) and spec is not None and _can_defer(X.shape[0], X.dtype)
    out_stats = BwdStats(sum(w.shape[0] for w in Ws)) if defer_out else None
    out, bn_save = _MultiMLP.apply(X, gamma, beta, run_mean, run_var, chan, x_cols, training, p_drop, seed, momentum, eps, in_defs, defer_out,
                                   in_stats, out_stats, *Ws, *bs)
    if defer_out:
        p = float(p_drop) if training else 0.0
        return DeferredAct(out, bn_save, out.shape[1], 0, tuple((c, f, bool(d) and p > 0) for c, f, d in spec), p, seed, out_stats)
    return out


class _EdgeConv(Function):
    @staticmethod
    def forward(ctx, xp, W2d, gamma, beta, run_mean, run_var, graph, training, act, slope, momentum, eps, out_buf=None, grad_accum=None,
                out_bounds=None, x_bounds=None):
        lib = _lib.load()
        prec = ctx.prec = gemm_precision.code()     # the backward gets the same products, whenever it runs
        xp = _rows(xp)
        _lib.require_gpu(xp, W2d, gamma)
        W2d = W2d.contiguous()
        P, C = xp.shape
        Cout = W2d.shape[0]
        assert W2d.shape[1] == 2 * C, (W2d.shape, C)
        dev = xp.device
        if out_buf is None:
            out = torch.empty((P, Cout), dtype=torch.float32, device=dev)
        else:                                   # a column slice of the caller's concatenation buffer, written in place
            assert out_buf.shape == (P, Cout) and out_buf.stride(1) == 1 and out_buf.dtype == torch.float32
            out = out_buf
        uv = torch.empty((P, 2 * Cout), dtype=torch.float32, device=dev)
        msel = torch.empty((P, Cout), dtype=torch.float32, device=dev)
        s1 = torch.empty((P, Cout), dtype=torch.float32, device=dev)
        argsel = torch.empty((P, Cout), dtype=torch.uint8, device=dev)
        bn_save = torch.empty((4, Cout), dtype=torch.float32, device=dev)
        # the folded weight, kept for the backward's dgrad (grad mode is OFF inside Function.forward: ctx.needs_input_grad is what tells
        # whether a backward will want dx; `torch.is_grad_enabled() and ...` here never kept it and the backward rebuilt it, 4 launches per step)
        keep_wd = bool(ctx.needs_input_grad[0])
        Wd = torch.empty((2 * Cout, C), dtype=torch.float32, device=dev) if keep_wd else None
        ws, wsn = _lib.workspace(dev, P, C, 2 * Cout)
        # (mode "f16x3", the layers whose [u|v] GEMM runs on the split kernel: the input's and the folded weight's bounds, measured by this
        # call, are read again by the backward's weight gradient and dgrad)
        # out_bounds (SliceBounds, optional): room for the analytic bound of this layer's output, which the call fills in training mode;
        # x_bounds: the input's bound when the caller has one (the previous layer's out_bounds)
        ctx.bounds = ((x_bounds if x_bounds is not None and x_bounds.valid else OperandBounds(dev)),
                      OperandBounds(dev) if Wd is not None else None) if prec == 3 and C >= 128 else (None, None)
        with _offer_bounds(prec, (xp, ctx.bounds[0]), (Wd, ctx.bounds[1]), (out, out_bounds)):
            _lib.check(lib.mlsp_edgeconv_fwd_f32(
                xp.data_ptr(), xp.stride(0), graph.idx.data_ptr(), W2d.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                _lib.ptr(run_mean), _lib.ptr(run_var), momentum, eps, act, slope, int(training), graph.B, graph.N, C, Cout,
                graph.k, out.data_ptr(), out.stride(0), uv.data_ptr(), msel.data_ptr(), argsel.data_ptr(), s1.data_ptr(),
                bn_save.data_ptr(), _lib.ptr(Wd), prec, ws, wsn, _lib.stream()), "mlsp_edgeconv_fwd_f32")
        if _sel_record is not None or _sel_forced is not None:
            _selection_hook(argsel)
        if out_buf is not None:
            out = out_buf.view_as(out_buf)      # a fresh alias: the Function's output, distinct from its (non-differentiable) input
        ctx.save_for_backward(xp, W2d, out, uv, msel, argsel, s1, bn_save, Wd)
        ctx.cfg = (graph, training, act, slope, C, Cout)
        ctx.grad_accum = grad_accum
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dOut):
        lib = _lib.load()
        xp, W2d, out, uv, msel, argsel, s1, bn_save, Wd = ctx.saved_tensors
        graph, training, act, slope, C, Cout = ctx.cfg
        if graph.rev_off is None:
            raise RuntimeError("EdgeConv backward needs the reverse neighbour index (knn_graph(need_reverse=True))")
        dOut = _rows(dOut)                      # a column slice of the concatenation's gradient is read in place
        dev = dOut.device
        P = xp.shape[0]
        dx, accumulate = None, 0
        if ctx.needs_input_grad[0]:
            if ctx.grad_accum is not None:      # the input is a column slice of the concatenation: add into that slice of its gradient
                dx, accumulate = ctx.grad_accum.claim((P, C), dev)
            else:
                dx = torch.empty((P, C), dtype=torch.float32, device=dev)
        dW = torch.empty_like(W2d)
        dgamma = torch.empty((Cout,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((Cout,), dtype=torch.float32, device=dev)
        ws, wsn = _lib.workspace(dev, P, C, 2 * Cout)
        with _offer_bounds(ctx.prec, (xp, ctx.bounds[0]), (Wd, ctx.bounds[1])):
            _lib.check(lib.mlsp_edgeconv_bwd_f32(
                dOut.data_ptr(), dOut.stride(0), xp.data_ptr(), xp.stride(0), graph.rev_off.data_ptr(), graph.rev_ent.data_ptr(),
                W2d.data_ptr(), out.data_ptr(), out.stride(0), uv.data_ptr(), msel.data_ptr(), argsel.data_ptr(), s1.data_ptr(),
                bn_save.data_ptr(), _lib.ptr(Wd), act, slope, int(training), graph.B, graph.N, C, Cout, graph.k, _lib.ptr(dx),
                dx.stride(0) if dx is not None else C, accumulate, dW.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ctx.prec, ws, wsn,
                _lib.stream()), "mlsp_edgeconv_bwd_f32")
        return (dx, dW, dgamma, dbeta) + (None,) * 12


def edgeconv(xp, graph, W2d, gamma, beta, run_mean, run_var, training, act=ACT_LRELU, slope=0.2, momentum=0.1, eps=1e-5, out=None,
             grad_accum=None, out_bounds=None, x_bounds=None):
    """Fused get_graph_feature + conv_2d + max over k (Models.py:115-129).  xp [P,C] -> [P,Cout].  `out`: a [P,Cout] column
    slice of a wider buffer to write the result into (see join_columns).  `grad_accum`: the SharedInputGrad of a fan_out whose alias
    xp is (join_columns(..., accs=) points it at xp's slice of the concatenation's gradient: the input gradient is added there).
    `out_bounds` / `x_bounds` (SliceBounds, mode "f16x3"): room for the analytic bound of the output / the input's bound."""
    return _EdgeConv.apply(xp, W2d, gamma, beta, run_mean, run_var, graph, training, act, slope, momentum, eps, out, grad_accum,
                           out_bounds, x_bounds)


class _TnetEdge(Function):
    @staticmethod
    def forward(ctx, xp, W1, g1, b1, rm1, rv1, W2, g2, b2, rm2, rv2, graph, training, slope, momentum, eps, out_bounds=None):
        lib = _lib.load()
        prec = ctx.prec = gemm_precision.code()     # the backward gets the same products, whenever it runs
        xp = _rows(xp)
        _lib.require_gpu(xp, W1, W2)
        W1, W2 = W1.contiguous(), W2.contiguous()
        P, C = xp.shape
        C1, C2 = W1.shape[0], W2.shape[0]
        assert W1.shape[1] == 2 * C and W2.shape[1] == C1
        dev = xp.device
        out = torch.empty((P, C2), dtype=torch.float32, device=dev)
        uv = torch.empty((P, 2 * C1), dtype=torch.float32, device=dev)
        s1 = torch.empty((P, C1), dtype=torch.float32, device=dev)
        bn1 = torch.empty((4, C1), dtype=torch.float32, device=dev)
        zsel = torch.empty((P, C2), dtype=torch.float32, device=dev)
        argsel = torch.empty((P, C2), dtype=torch.uint8, device=dev)
        bn2 = torch.empty((4, C2), dtype=torch.float32, device=dev)
        ws, wsn = _lib.workspace(dev, P * graph.k, C1, C1)
        with _offer_bounds(prec, (out, out_bounds)):          # (out_bounds: room for the analytic bound of the stage's output, see _EdgeConv)
            _lib.check(lib.mlsp_tnet_edge_fwd_f32(
                xp.data_ptr(), xp.stride(0), graph.idx.data_ptr(), W1.data_ptr(), g1.data_ptr(), b1.data_ptr(), _lib.ptr(rm1),
                _lib.ptr(rv1), W2.data_ptr(), g2.data_ptr(), b2.data_ptr(), _lib.ptr(rm2), _lib.ptr(rv2), momentum, eps, slope,
                int(training), graph.B, graph.N, C, C1, C2, graph.k, out.data_ptr(), uv.data_ptr(), s1.data_ptr(), bn1.data_ptr(),
                zsel.data_ptr(), argsel.data_ptr(), bn2.data_ptr(), prec, ws, wsn, _lib.stream()), "mlsp_tnet_edge_fwd_f32")
        if _sel_record is not None or _sel_forced is not None:
            _selection_hook(argsel)
        ctx.save_for_backward(xp, W1, W2, out, uv, s1, bn1, zsel, argsel, bn2)
        ctx.cfg = (graph, training, slope, C, C1, C2)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dOut):
        lib = _lib.load()
        xp, W1, W2, out, uv, s1, bn1, zsel, argsel, bn2 = ctx.saved_tensors
        graph, training, slope, C, C1, C2 = ctx.cfg
        dOut = dOut.contiguous()
        dev = dOut.device
        P = xp.shape[0]
        dx = torch.empty((P, C), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        if graph.rev_off is None and (dx is not None or C > 4):
            raise RuntimeError("tnet_edge backward with an input gradient needs the reverse neighbour index (knn_graph(need_reverse=True))")
        dW1, dW2 = torch.empty_like(W1), torch.empty_like(W2)
        dg1 = torch.empty((C1,), dtype=torch.float32, device=dev)
        db1 = torch.empty((C1,), dtype=torch.float32, device=dev)
        dg2 = torch.empty((C2,), dtype=torch.float32, device=dev)
        db2 = torch.empty((C2,), dtype=torch.float32, device=dev)
        ws, wsn = _lib.workspace(dev, P * graph.k, C1, C1)
        _lib.check(lib.mlsp_tnet_edge_bwd_f32(
            dOut.data_ptr(), xp.data_ptr(), xp.stride(0), graph.idx.data_ptr(), _lib.ptr(graph.rev_off),
            _lib.ptr(graph.rev_ent), W1.data_ptr(), W2.data_ptr(), out.data_ptr(), uv.data_ptr(), s1.data_ptr(), bn1.data_ptr(),
            zsel.data_ptr(), argsel.data_ptr(), bn2.data_ptr(), slope, int(training), graph.B, graph.N, C, C1, C2, graph.k,
            _lib.ptr(dx), dW1.data_ptr(), dg1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), dg2.data_ptr(), db2.data_ptr(),
            ctx.prec, ws, wsn, _lib.stream()), "mlsp_tnet_edge_bwd_f32")
        return (dx, dW1, dg1, db1, None, None, dW2, dg2, db2) + (None,) * 8


def tnet_edge(xp, graph, W1, g1, b1, rm1, rv1, W2, g2, b2, rm2, rv2, training, slope=0.2, momentum=0.1, eps=1e-5, out_bounds=None):
    """Fused T-Net per-edge stage (model_utils.py:111-115): [P,C] -> [P,128]; needs C1 = 64, C2 = 128."""
    return _TnetEdge.apply(xp, W1, g1, b1, rm1, rv1, W2, g2, b2, rm2, rv2, graph, training, slope, momentum, eps, out_bounds)


def tnet_edge_supported(W1, W2, k):
    return W1.shape[0] == 64 and tuple(W2.shape) == (128, 64) and 1 <= k <= 128


# Deferred activations of chained layers (DeferredAct below): a Linear+BN+act(+dropout) layer whose consumers are all GEMM layers writes
# only its pre-BatchNorm output; the consumers apply scale / shift / activation / dropout while they stage their operand
# (gemm_split_kernel<.., XF, XD>, the f32 transform kernels, the thin streaming kernels).  On by default in the "bf16x6" and "fp32" product
# modes with fp32 activation storage; MLSP_DEFERRED_ACT=0 restores the materialised chain (an A/B switch, read once).
_DEFER_CHAINS = os.environ.get("MLSP_DEFERRED_ACT", "1") not in ("0", "")


# BatchNorm-backward sums of a deferred layer, produced by its consumers' dgrads (csrc/gemm.hip gemm_out_bs, thin.hip): MLSP_BWD_STATS_FUSED=0
# restores the streaming reduction pass (an A/B switch, read once).
_FUSE_BWD_STATS = os.environ.get("MLSP_BWD_STATS_FUSED", "1") not in ("0", "")


class BwdStats:
    """Hand-off between the backward passes of a deferred layer (the PRODUCER of a pre-BN matrix [M, ld]) and of its consumers.  The
    gradient w.r.t. the producer's activated output is written by the consumers' dgrads; each of them can apply the producer's activation
    derivative and dropout mask in that launch and leave the producer's per-panel column sums here, so the producer needs no reduction
    pass.  All or nothing: every consumer `promise`s its column range at forward time (with the number of 128-row panels its dgrad
    would write, 0 = it cannot); only if the promises tile [0, ld) exactly with one common panel count do the consumers do it
    (`agreed`), and the producer uses the sums only if every range was `deliver`ed in the same backward pass (`take`)."""
    __slots__ = ("ld", "promised", "part", "parts", "task", "delivered", "_ok")

    def __init__(self, ld):
        self.ld, self.promised, self.part, self.parts, self.task, self.delivered, self._ok = int(ld), [], None, 0, None, 0, None

    def promise(self, col, width, parts):
        self.promised.append((int(col), int(width), int(parts)))
        self._ok = None

    def agreed(self):
        if self._ok is None:
            pr = sorted(self.promised)
            ok = bool(pr) and pr[0][0] == 0 and pr[-1][0] + pr[-1][1] == self.ld and pr[0][2] > 0 and len({p[2] for p in pr}) == 1
            ok = ok and all(a[0] + a[1] == b[0] for a, b in zip(pr, pr[1:]))
            self._ok = ok
        return self._ok

    def buffer(self, device):
        """The partial rows of the current backward pass: [parts][2][ld] fp64 column sums (every column written by exactly one consumer),
        followed by [parts][ld] fp32 column maxima of |d'| (ABI v13: the producer's GEMMs bound their two-piece f16 products with them)."""
        task = torch._C._current_graph_task_id()
        if self.part is None or self.task != task:
            parts = self.promised[0][2]
            self.part = torch.empty((parts * 2 * self.ld + (parts * self.ld + 1) // 2,), dtype=torch.float64, device=device)
            self.parts, self.task, self.delivered = parts, task, 0
        self.delivered += 1
        return self.part

    def take(self):
        """producer side, three outcomes: (partial rows, panel count) when EVERY consumer delivered in this backward pass; None when none
        did (the incoming gradient is unmasked: the producer masks and reduces); (None, -1) when only SOME did -- a loss on a subset of
        the heads (PointDA/trainer.py:551-565): the delivered columns of the incoming gradient already carry the activation derivative
        and the dropout mask, the others are zero, the sums are incomplete -> the producer reduces without masking a second time."""
        if self.part is None or self.task != torch._C._current_graph_task_id() or not self.agreed():
            self.part = self.task = None
            return None
        part, self.part, self.task = self.part, None, None
        if self.delivered != len(self.promised):
            return None, -1
        return part, self.parts


class DeferredAct:
    """Output of a chained Linear+BN+act layer (pointmlp / multimlp with chain=True, fp32 storage) whose BatchNorm scale / shift,
    activation and dropout have NOT been applied yet.  `y`: columns [col, col + width) of the producer's pre-BN matrix [M, ld] -- an
    autograd tensor that STANDS FOR the activated output (gradients flowing into it are gradients w.r.t. the activated value);
    `bn_save` [4, ld]: the producer's scale | shift | mean | invstd; `spec`: ((width, negative-side factor, dropout on), ...) covering
    the columns of `y` in order (factor 0 = ReLU, 0.2 = LeakyReLU(0.2), 1 = no activation); `p`, `seed`: the producer's dropout stream.
    Legal consumers: pointmlp (a slice with ONE spec entry), multimlp (segments inside one spec entry each), split_columns_shared."""
    __slots__ = ("y", "bn_save", "ld", "col", "spec", "p", "seed", "stats")

    def __init__(self, y, bn_save, ld, col, spec, p, seed, stats=None):
        self.y, self.bn_save, self.ld, self.col, self.spec, self.p, self.seed = y, bn_save, int(ld), int(col), tuple(spec), float(p), int(seed)
        self.stats = stats                   # BwdStats of the producer (None: its backward reduces by itself)
        assert sum(w for w, _, _ in self.spec) == y.shape[1], (self.spec, y.shape)

    @property
    def shape(self):
        return self.y.shape

    @property
    def device(self):
        return self.y.device

    @property
    def dtype(self):
        return self.y.dtype

    def dim(self):
        return self.y.dim()

    def entry(self, c0, width):
        """(negative-side factor, dropout rate) of the columns [c0, c0 + width) of y -- they must lie inside one spec entry"""
        o = 0
        for w, fac, drop in self.spec:
            if o <= c0 and c0 + width <= o + w:
                return float(fac), (self.p if drop else 0.0)
            o += w
        raise ValueError("DeferredAct: columns [%d, %d) straddle activations %r" % (c0, c0 + width, self.spec))

    def desc(self, c0=0, width=None):
        """mlsp_defer_t fields of the columns [c0, c0 + width) of y: (bn_save tensor, ld, col, act, slope, p_drop, seed)"""
        width = self.y.shape[1] - c0 if width is None else width
        fac, p = self.entry(c0, width)
        return (self.bn_save, self.ld, self.col + c0, ACT_LRELU, fac, p, self.seed)

    def sub(self, y_slice, c0):
        """the DeferredAct of a column slice `y_slice` = y[:, c0 : c0 + w] (an autograd view of y)"""
        w, spec, o = y_slice.shape[1], [], 0
        for sw, fac, drop in self.spec:
            lo, hi = max(o, c0), min(o + sw, c0 + w)
            if hi > lo:
                spec.append((hi - lo, fac, drop))
            o += sw
        return DeferredAct(y_slice, self.bn_save, self.ld, self.col + c0, spec, self.p, self.seed, self.stats)


def _defer_struct(d):
    """ctypes mlsp_defer_t from DeferredAct.desc()"""
    bn, ld, col, act, slope, p, seed = d
    s = _lib.Defer()
    s.bn_save, s.ld, s.col, s.act, s.slope, s.p_drop, s.seed = bn.data_ptr(), ld, col, act, slope, p, seed
    return s


def _can_defer(M, dtype):
    return (_DEFER_CHAINS and activation_storage.current == "fp32" and gemm_precision.current in ("fp32", "bf16x6", "f16x3") and M > 32
            and dtype == torch.float32)


_stats_parts_cache = {}


def _stats_parts(key, query):
    """memoised mlsp_*_bwd_stats_parts answer (a pure function of the layer shape and the product mode)"""
    v = _stats_parts_cache.get(key)
    if v is None:
        v = _stats_parts_cache[key] = int(query())
    return v


class _PointMLP(Function):
    @staticmethod
    def forward(ctx, X, W, bias, gbias, gamma, beta, run_mean, run_var, rows_per_group, training, act, slope, p_drop, seed,
                momentum, eps, grad_accum=None, out_bf16=False, in_def=None, defer_out=False, grad_cols=None, in_stats=None, out_stats=None,
                x_bounds=None):
        lib = _lib.load()
        prec = ctx.prec = gemm_precision.code()     # the backward gets the same products, whenever it runs
        X = _rows(X, allow_bf16=True)
        ctx.grad_cols = grad_cols
        _lib.require_gpu(X, W)
        if W.stride(1) != 1:
            W = W.contiguous()
        M, Cin = X.shape
        Cout = W.shape[0]
        assert W.shape[1] == Cin, (W.shape, X.shape)
        dev = X.device
        has_bn = gamma is not None
        x_bf16 = X.dtype == torch.bfloat16
        mx = x_bf16 or out_bf16            # bf16 activation storage on either side of this layer (mlsp_pointmlp_*_mx)
        if mx and not (has_bn and lib.mlsp_pointmlp_mx_supported(M, Cin, Cout, X.stride(0), int(x_bf16), int(training), prec)):
            if x_bf16:
                raise RuntimeError("pointmlp: bf16 input on a layer the bf16-storage kernels do not cover (M=%d Cin=%d Cout=%d)" % (M, Cin, Cout))
            mx = out_bf16 = False
        assert not (mx and (in_def is not None or defer_out)), "bf16 activation storage and deferred activations are exclusive"
        assert not defer_out or has_bn
        odt = torch.bfloat16 if out_bf16 else torch.float32
        Z = None if defer_out else torch.empty((M, Cout), dtype=odt, device=dev)
        Y = torch.empty((M, Cout), dtype=odt, device=dev) if has_bn else None
        bn_save = torch.empty((4, Cout), dtype=torch.float32, device=dev) if has_bn else None
        if gbias is not None:
            gbias = gbias.contiguous()
            assert gbias.shape[1] == Cout and gbias.shape[0] * rows_per_group == M
        if bias is not None:
            bias = bias.contiguous()
        p = float(p_drop) if training else 0.0
        ws, wsn = _lib.workspace(dev, M, Cin, Cout)
        # a deferred layer with a per-cloud bias whose consumers leave its BatchNorm-backward sums: the backward may never form dY, and
        # then takes the per-cloud bias gradient from those sums and the clouds' column sums of Y -- which the forward has at hand
        ysum = (torch.empty((gbias.shape[0], Cout), dtype=torch.float32, device=dev)
                if (gbias is not None and defer_out and training and out_stats is not None and _FUSE_BWD_STATS and M > 32) else None)
        # mode "f16x3": bounds of the operands as they lie in memory, measured once by this call and read again by the backward (a deferred
        # input is bounded analytically; x_bounds: the caller shares X's with the other layers that read X)
        ctx.bounds = ((x_bounds if x_bounds is not None else OperandBounds(dev)) if in_def is None and not x_bf16 else None,
                      _weight_bounds(W, dev)) if prec == 3 and M > 32 else (None, None)
        ctx_offer = _offer_bounds(prec, (X, ctx.bounds[0]), (W, ctx.bounds[1]))
        ctx_offer.__enter__()
        try:
            if mx:
                _lib.check(lib.mlsp_pointmlp_fwd_mx(
                    X.data_ptr(), int(x_bf16), X.stride(0), M, Cin, W.data_ptr(), W.stride(0), Cout, _lib.ptr(bias), _lib.ptr(gbias),
                    int(rows_per_group), gamma.data_ptr(), beta.data_ptr(), _lib.ptr(run_mean), _lib.ptr(run_var), momentum, eps,
                    int(training), act, slope, p, seed, Y.data_ptr(), Z.data_ptr(), int(out_bf16), bn_save.data_ptr(), prec, ws, wsn,
                    _lib.stream()), "mlsp_pointmlp_fwd_mx")
            elif in_def is not None:
                ds = _defer_struct(in_def)
                _lib.check(lib.mlsp_pointmlp_fwd_chain_f32(
                    X.data_ptr(), X.stride(0), _lib._c.byref(ds), M, Cin, W.data_ptr(), W.stride(0), Cout,
                    _lib.ptr(bias), _lib.ptr(gbias), int(rows_per_group), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(run_mean),
                    _lib.ptr(run_var), momentum, eps, int(training), act, slope, p, seed, _lib.ptr(Y), _lib.ptr(Z), _lib.ptr(bn_save), _lib.ptr(ysum),
                    prec, ws, wsn, _lib.stream()), "mlsp_pointmlp_fwd_chain_f32")
            else:
                _lib.check(lib.mlsp_pointmlp_fwd_f32(
                    X.data_ptr(), X.stride(0), M, Cin, W.data_ptr(), W.stride(0), Cout, _lib.ptr(bias), _lib.ptr(gbias),
                    int(rows_per_group), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(run_mean), _lib.ptr(run_var), momentum, eps,
                    int(training), act, slope, p, seed, _lib.ptr(Y), _lib.ptr(Z), _lib.ptr(bn_save), _lib.ptr(ysum), prec, ws, wsn, _lib.stream()),
                    "mlsp_pointmlp_fwd_f32")
        finally:
            ctx_offer.__exit__(None, None, None)
        ctx.in_def = in_def
        # fused BatchNorm-backward sums (BwdStats): as a consumer, promise the producer this layer's input columns (with the panel count
        # the dgrad would write; 0 = it cannot); as a deferred producer, remember where the consumers will leave ours
        ctx.in_stats, ctx.out_stats = None, (out_stats if defer_out else None)
        if in_stats is not None and in_def is not None and not mx:
            parts = 0
            if _FUSE_BWD_STATS and ctx.needs_input_grad[0] and grad_accum is None:
                lddx = grad_cols[0].width if grad_cols is not None else Cin
                parts = _stats_parts(("p", M, Cin, Cout, W.stride(0), lddx, prec), lambda: lib.mlsp_pointmlp_bwd_stats_parts(M, Cin, Cout, W.stride(0), lddx, prec))
            in_stats.promise(in_def[2], Cin, parts)
            ctx.in_stats = in_stats
        ctx.save_for_backward(X, W, Y, bn_save, ysum)
        ctx.cfg = (has_bn, training, act, slope, p, seed, bias is not None, gbias.shape[0] if gbias is not None else 0,
                   int(rows_per_group))
        ctx.grad_accum = grad_accum
        ctx.mx = (mx, x_bf16, out_bf16)
        if bn_save is not None:
            ctx.mark_non_differentiable(bn_save)
        ctx.set_materialize_grads(False)       # (autograd would zero-fill a [4, Cout] gradient for bn_save on every backward: 11 fills a step)
        # a deferred layer hands out its pre-BN output: Y is both saved and returned (same storage, nobody writes to it)
        return (Y.view_as(Y) if defer_out else Z), bn_save

    @staticmethod
    @once_differentiable
    def backward(ctx, dZ, _dbn=None):
        if dZ is None:
            return (None,) * 24
        lib = _lib.load()
        X, W, Y, bn_save, ysum = ctx.saved_tensors
        has_bn, training, act, slope, p, seed, has_bias, G, rpg = ctx.cfg
        mx, x_bf16, out_bf16 = ctx.mx
        dZ = dZ.contiguous()
        if dZ.dtype != (torch.bfloat16 if out_bf16 else torch.float32):
            dZ = dZ.to(torch.bfloat16 if out_bf16 else torch.float32)
        dev = dZ.device
        M, Cin = X.shape
        Cout = W.shape[0]
        dX, accumulate, lddx = None, 0, Cin
        if ctx.needs_input_grad[0]:
            if ctx.grad_accum is not None and not x_bf16:
                dX, accumulate = ctx.grad_accum.claim((M, Cin), dev)
            elif ctx.grad_cols is not None:           # X is a column slice of a wider matrix: write into the same slice of ITS gradient
                acc, col = ctx.grad_cols
                dX = acc.claim(M, col, Cin, dev, X.dtype)
                lddx = dX.stride(0)
            else:
                dX = torch.empty((M, Cin), dtype=X.dtype, device=dev)
        dW = torch.empty((Cout, Cin), dtype=torch.float32, device=dev)
        dbias = torch.empty((Cout,), dtype=torch.float32, device=dev) if has_bias else None
        dgbias = torch.empty((G, Cout), dtype=torch.float32, device=dev) if G else None
        dgamma = torch.empty((Cout,), dtype=torch.float32, device=dev) if has_bn else None
        dbeta = torch.empty((Cout,), dtype=torch.float32, device=dev) if has_bn else None
        ws, wsn = _lib.workspace(dev, M, Cin, Cout)
        pre = ctx.out_stats.take() if ctx.out_stats is not None else None          # our own sums, left by the consumers' dgrads
        pre_ptr, pre_n = (pre[0].data_ptr() if pre[0] is not None else None, pre[1]) if pre is not None else (None, 0)
        ins = None
        if ctx.in_stats is not None and dX is not None and ctx.in_stats.agreed():
            ins = ctx.in_stats.buffer(dev)                                          # the producer's sums: this call's dgrad writes our columns
        ctx_offer = _offer_bounds(ctx.prec, (X, ctx.bounds[0]), (W, ctx.bounds[1]))
        ctx_offer.__enter__()
        try:
            if mx:
                _lib.check(lib.mlsp_pointmlp_bwd_mx(
                    dZ.data_ptr(), X.data_ptr(), int(x_bf16), X.stride(0), M, Cin, W.data_ptr(), W.stride(0), Cout, Y.data_ptr(),
                    int(out_bf16), bn_save.data_ptr(), int(training), act, slope, p, seed, G, rpg, _lib.ptr(dX), lddx, accumulate,
                    dW.data_ptr(), _lib.ptr(dbias), _lib.ptr(dgbias), dgamma.data_ptr(), dbeta.data_ptr(), ctx.prec, ws, wsn, _lib.stream()),
                    "mlsp_pointmlp_bwd_mx")
            elif ctx.in_def is not None:
                ds = _defer_struct(ctx.in_def)
                _lib.check(lib.mlsp_pointmlp_bwd_chain_f32(
                    dZ.data_ptr(), X.data_ptr(), X.stride(0), _lib._c.byref(ds), M, Cin, W.data_ptr(), W.stride(0),
                    Cout, _lib.ptr(Y), _lib.ptr(bn_save), int(has_bn), int(training), act, slope, p, seed, G, rpg, _lib.ptr(dX), lddx, accumulate,
                    dW.data_ptr(), _lib.ptr(dbias), _lib.ptr(dgbias), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ins), pre_ptr, pre_n, _lib.ptr(ysum),
                    ctx.prec, ws, wsn, _lib.stream()), "mlsp_pointmlp_bwd_chain_f32")
            else:
                _lib.check(lib.mlsp_pointmlp_bwd_f32(
                    dZ.data_ptr(), X.data_ptr(), X.stride(0), M, Cin, W.data_ptr(), W.stride(0), Cout, _lib.ptr(Y),
                    _lib.ptr(bn_save), int(has_bn), int(training), act, slope, p, seed, G, rpg, _lib.ptr(dX), lddx, accumulate, dW.data_ptr(),
                    _lib.ptr(dbias), _lib.ptr(dgbias), _lib.ptr(dgamma), _lib.ptr(dbeta), pre_ptr, pre_n, _lib.ptr(ysum), ctx.prec, ws, wsn,
                    _lib.stream()), "mlsp_pointmlp_bwd_f32")
        finally:
            ctx_offer.__exit__(None, None, None)
        return (dX, dW, dbias, dgbias, dgamma, dbeta) + (None,) * 18


def pointmlp(X, W, bias=None, gbias=None, gamma=None, beta=None, run_mean=None, run_var=None, rows_per_group=0,
             training=True, act=ACT_NONE, slope=0.2, p_drop=0.0, momentum=0.1, eps=1e-5, grad_accum=None, chain=False, grad_cols=None,
             defer=False, x_bounds=None):
    """Linear/1x1-conv (+bias, +per-group bias) [+ BatchNorm + act + dropout] on a [M,Cin] row matrix.
    `grad_accum`: the SharedInputGrad of a fan_out(X, n) whose alias this X is.  `chain=True`: the only consumer of the output is
    another pointmlp BN layer, so under activation_storage("bf16") Y / Z may be stored as bf16.  `grad_cols` = (SharedColumnGrad, first
    column): X is that column slice of a split_columns_shared() matrix.  `defer=True`: every consumer of the output is a GEMM layer
    (pointmlp / multimlp, with or without BatchNorm): under fp32 storage the result may be a DeferredAct (`chain=True` implies it).
    `x_bounds` (OperandBounds of X, optional): shared with the other layers that read X."""
    seed = _next_seed() if (training and p_drop > 0) else 0
    if isinstance(X, DeferredAct) and grad_cols is not None:
        assert grad_cols[0].width == X.ld and grad_cols[1] == X.col, "grad_cols of a deferred slice: its columns in the producer's matrix"
    if (not isinstance(X, DeferredAct) and X.dtype == torch.bfloat16 and not (
            gamma is not None and X.dim() == 2 and X.stride(1) == 1 and
            _lib.load().mlsp_pointmlp_mx_supported(X.shape[0], X.shape[1], W.shape[0], X.stride(0), 1, int(training), gemm_precision.code()))):
        # the producer stored its output as bf16 (its own shape allowed it) but THIS layer's shape is outside the bf16-storage
        # kernels (e.g. a consumer whose GEMM splits K): widen once and run the fp32 layer; autograd narrows the gradient again
        X = X.float()
    out_bf16 = bool(chain) and activation_storage.current == "bf16" and gamma is not None
    in_def = in_stats = None
    if isinstance(X, DeferredAct):
        in_def, in_stats, X = X.desc(), X.stats, X.y
    # fp32 storage: a chained layer leaves its BN + activation (+ dropout) to its consumers' GEMM operand loads
    defer_out = bool(chain or defer) and gamma is not None and _can_defer(X.shape[0], X.dtype)
    out_stats = BwdStats(W.shape[0]) if defer_out else None
    out, bn_save = _PointMLP.apply(X, W, bias, gbias, gamma, beta, run_mean, run_var, rows_per_group, training, act, slope, p_drop,
                                   seed, momentum, eps, grad_accum, out_bf16, in_def, defer_out, grad_cols, in_stats, out_stats, x_bounds)
    if defer_out:
        fac = 1.0 if act == ACT_NONE else 0.0 if act == ACT_RELU else float(slope)
        p = float(p_drop) if training else 0.0
        return DeferredAct(out, bn_save, out.shape[1], 0, ((out.shape[1], fac, p > 0),), p, seed, out_stats)
    return out


class _PointMLPColMax(Function):
    @staticmethod
    def forward(ctx, X, W, gamma, beta, run_mean, run_var, B, N, training, act, slope, momentum, eps, grad_accum=None, x_bounds=None):
        lib = _lib.load()
        prec = ctx.prec = gemm_precision.code()     # the backward gets the same products, whenever it runs
        ctx.grad_accum = grad_accum
        X = X.contiguous()
        _lib.require_gpu(X, W)
        if W.stride(1) != 1:
            W = W.contiguous()
        P, Cin = X.shape
        Cout = W.shape[0]
        assert P == B * N and W.shape[1] == Cin
        dev = X.device
        out = torch.empty((B, Cout), dtype=torch.float32, device=dev)
        ysel = torch.empty((B, Cout), dtype=torch.float32, device=dev)
        arg = torch.empty((B, Cout), dtype=torch.int32, device=dev)
        bn_save = torch.empty((4, Cout), dtype=torch.float32, device=dev)
        ws, wsn = _lib.workspace(dev, P, Cin, Cout)
        # (mode "f16x3": the operand bounds this call measures are kept for the backward, which reads X and W again)
        ctx.bounds = (x_bounds if x_bounds is not None else OperandBounds(dev), _weight_bounds(W, dev)) if prec == 3 else (None, None)
        with _offer_bounds(prec, (X, ctx.bounds[0]), (W, ctx.bounds[1])):
            _lib.check(lib.mlsp_pointmlp_colmax_fwd_f32(
                X.data_ptr(), X.stride(0), B, N, Cin, W.data_ptr(), W.stride(0), Cout, gamma.data_ptr(), beta.data_ptr(),
                _lib.ptr(run_mean), _lib.ptr(run_var), momentum, eps, int(training), act, slope, out.data_ptr(), ysel.data_ptr(),
                arg.data_ptr(), bn_save.data_ptr(), prec, ws, wsn, _lib.stream()), "mlsp_pointmlp_colmax_fwd_f32")
        if _sel_record is not None or _sel_forced is not None:
            _selection_hook(arg)
        ctx.save_for_backward(X, W, out, ysel, arg, bn_save)
        ctx.cfg = (B, N, training, act, slope)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dOut):
        lib = _lib.load()
        X, W, out, ysel, arg, bn_save = ctx.saved_tensors
        B, N, training, act, slope = ctx.cfg
        dOut = dOut.contiguous()
        dev = dOut.device
        P, Cin = X.shape
        Cout = W.shape[0]
        dX, accumulate = None, 0
        if ctx.needs_input_grad[0]:
            if ctx.grad_accum is not None:
                dX, accumulate = ctx.grad_accum.claim((P, Cin), dev)
            else:
                dX = torch.empty((P, Cin), dtype=torch.float32, device=dev)
        dW = torch.empty((Cout, Cin), dtype=torch.float32, device=dev)
        dgamma = torch.empty((Cout,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((Cout,), dtype=torch.float32, device=dev)
        ws, wsn = _lib.workspace(dev, max(P, Cout), Cin, max(Cout, Cin))
        with _offer_bounds(ctx.prec, (X, ctx.bounds[0]), (W, ctx.bounds[1])):
            _lib.check(lib.mlsp_pointmlp_colmax_bwd_f32(
                dOut.data_ptr(), X.data_ptr(), X.stride(0), B, N, Cin, W.data_ptr(), W.stride(0), Cout, out.data_ptr(),
                ysel.data_ptr(), arg.data_ptr(), bn_save.data_ptr(), int(training), act, slope, _lib.ptr(dX), accumulate, dW.data_ptr(),
                dgamma.data_ptr(), dbeta.data_ptr(), ctx.prec, ws, wsn, _lib.stream()), "mlsp_pointmlp_colmax_bwd_f32")
        return (dX, dW, dgamma, dbeta) + (None,) * 11


def pointmlp_colmax(X, W, gamma, beta, run_mean, run_var, B, N, training=True, act=ACT_LRELU, slope=0.2, momentum=0.1, eps=1e-5,
                    grad_accum=None, x_bounds=None):
    """conv (bias-free) + BN + act + max over the N rows of each of the B clouds: [B*N, Cin] -> [B, Cout]
    (Models.py:132-136; model_utils.py:116-117).  Closed-form backward through the Gram matrix (colmax.hip).
    `x_bounds` (OperandBounds of X, optional): shared with the other layers that read X."""
    return _PointMLPColMax.apply(X, W, gamma, beta, run_mean, run_var, B, N, training, act, slope, momentum, eps, grad_accum, x_bounds)


class _SegMax(Function):
    @staticmethod
    def forward(ctx, Z, k):
        lib = _lib.load()
        Z = Z.contiguous()
        _lib.require_gpu(Z)
        E, C = Z.shape
        P = E // k
        out = torch.empty((P, C), dtype=torch.float32, device=Z.device)
        argk = torch.empty((P, C), dtype=torch.uint8, device=Z.device)
        _lib.check(lib.mlsp_segmax_fwd_f32(Z.data_ptr(), P, k, C, out.data_ptr(), argk.data_ptr(), _lib.stream()),
                   "mlsp_segmax_fwd_f32")
        ctx.save_for_backward(argk)
        ctx.k = k
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dOut):
        lib = _lib.load()
        (argk,) = ctx.saved_tensors
        dOut = dOut.contiguous()
        P, C = dOut.shape
        dZ = torch.empty((P * ctx.k, C), dtype=torch.float32, device=dOut.device)
        _lib.check(lib.mlsp_segmax_bwd_f32(dOut.data_ptr(), argk.data_ptr(), P, ctx.k, C, dZ.data_ptr(), _lib.stream()),
                   "mlsp_segmax_bwd_f32")
        return dZ, None


def segmax(Z, k):
    """[P*k, C] edge-major -> [P, C]: max over each point's k edges (model_utils.py:114)."""
    return _SegMax.apply(Z, k)


class _PointMLPSegMax(Function):
    """Linear + BatchNorm + act + max over every k consecutive rows (mlsp_pointmlp_segmax_*_f32): the activated [M, Cout] tensor and its
    gradient are never materialised."""

    @staticmethod
    def forward(ctx, X, W, bias, gamma, beta, run_mean, run_var, k, training, act, slope, momentum, eps):
        lib = _lib.load()
        prec = ctx.prec = gemm_precision.code()     # the backward gets the same products, whenever it runs
        X = _rows(X)
        _lib.require_gpu(X, W)
        if W.stride(1) != 1:
            W = W.contiguous()
        M, Cin = X.shape
        Cout = W.shape[0]
        dev = X.device
        G = M // k
        Y = torch.empty((M, Cout), dtype=torch.float32, device=dev)
        out = torch.empty((G, Cout), dtype=torch.float32, device=dev)
        ysel = torch.empty((G, Cout), dtype=torch.float32, device=dev)
        argk = torch.empty((G, Cout), dtype=torch.uint8, device=dev)
        bn_save = torch.empty((4, Cout), dtype=torch.float32, device=dev)
        ws, wsn = _lib.workspace(dev, M, Cin, Cout)
        _lib.check(lib.mlsp_pointmlp_segmax_fwd_f32(X.data_ptr(), X.stride(0), M, Cin, W.data_ptr(), W.stride(0), Cout, _lib.ptr(bias),
                                                    gamma.data_ptr(), beta.data_ptr(), _lib.ptr(run_mean), _lib.ptr(run_var), float(momentum),
                                                    float(eps), int(training), int(act), float(slope), int(k), Y.data_ptr(), out.data_ptr(),
                                                    ysel.data_ptr(), argk.data_ptr(), bn_save.data_ptr(), prec, ws, wsn, _lib.stream()),
                   "mlsp_pointmlp_segmax_fwd_f32")
        if _sel_record is not None or _sel_forced is not None:
            forced = _sel_forced is not None
            _selection_hook(argk)              # test hooks (forced_selections): the slot the backward routes each group's gradient to
            if forced:                         # ... and the pre-BN value at that slot, which the backward's statistics read
                ysel.copy_(Y.view(G, int(k), Cout).gather(1, argk.long().unsqueeze(1)).squeeze(1))
        ctx.save_for_backward(X, W, Y, ysel, argk, bn_save)
        ctx.cfg = (int(k), bool(training), int(act), float(slope), bias is not None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dOut):
        lib = _lib.load()
        X, W, Y, ysel, argk, bn_save = ctx.saved_tensors
        k, training, act, slope, has_bias = ctx.cfg
        dOut = dOut.contiguous()
        M, Cin = X.shape
        Cout = W.shape[0]
        dev = dOut.device
        dX = torch.empty((M, Cin), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dW = torch.empty((Cout, Cin), dtype=torch.float32, device=dev)
        dbias = torch.empty((Cout,), dtype=torch.float32, device=dev) if has_bias else None
        dgamma = torch.empty((Cout,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((Cout,), dtype=torch.float32, device=dev)
        ws, wsn = _lib.workspace(dev, M, Cin, Cout)
        _lib.check(lib.mlsp_pointmlp_segmax_bwd_f32(dOut.data_ptr(), X.data_ptr(), X.stride(0), M, Cin, W.data_ptr(), W.stride(0), Cout,
                                                    Y.data_ptr(), ysel.data_ptr(), argk.data_ptr(), bn_save.data_ptr(), int(training), act, slope,
                                                    k, _lib.ptr(dX), Cin, dW.data_ptr(), _lib.ptr(dbias), dgamma.data_ptr(), dbeta.data_ptr(),
                                                    ctx.prec, ws, wsn, _lib.stream()), "mlsp_pointmlp_segmax_bwd_f32")
        return dX, dW, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None


def pointmlp_segmax_supported(M, Cout, k):
    return 1 <= k <= 255 and M % k == 0 and Cout % 4 == 0 and 256 % (Cout // 4) == 0 and Cout <= 1024


def pointmlp_segmax(X, W, k, bias=None, gamma=None, beta=None, run_mean=None, run_var=None, training=True, act=ACT_RELU, slope=0.2,
                    momentum=0.1, eps=1e-5):
    """max over every k consecutive rows of act(BN(X W^T + b)): [M, Cin] -> [M // k, Cout] without the activated [M, Cout] tensor."""
    return _PointMLPSegMax.apply(X, W, bias, gamma, beta, run_mean, run_var, k, training, act, slope, momentum, eps)


class _ColMax(Function):
    @staticmethod
    def forward(ctx, Z, B, N):
        lib = _lib.load()
        Z = Z.contiguous()
        _lib.require_gpu(Z)
        C = Z.shape[1]
        out = torch.empty((B, C), dtype=torch.float32, device=Z.device)
        arg = torch.empty((B, C), dtype=torch.int32, device=Z.device)
        _lib.check(lib.mlsp_colmax_fwd_f32(Z.data_ptr(), B, N, C, out.data_ptr(), arg.data_ptr(), _lib.stream()),
                   "mlsp_colmax_fwd_f32")
        ctx.save_for_backward(arg)
        ctx.dims = (B, N, C)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dOut):
        lib = _lib.load()
        (arg,) = ctx.saved_tensors
        B, N, C = ctx.dims
        dOut = dOut.contiguous()
        dZ = torch.empty((B * N, C), dtype=torch.float32, device=dOut.device)
        _lib.check(lib.mlsp_colmax_bwd_f32(dOut.data_ptr(), arg.data_ptr(), B, N, C, dZ.data_ptr(), _lib.stream()),
                   "mlsp_colmax_bwd_f32")
        return dZ, None, None


def colmax(Z, B, N):
    """[B*N, C] -> [B, C]: max over the points of each cloud (model_utils.py:117, Models.py:136)."""
    return _ColMax.apply(Z, B, N)


class _Transform3(Function):
    @staticmethod
    def forward(ctx, xp, T):
        lib = _lib.load()
        xp, T = xp.contiguous(), T.contiguous()
        _lib.require_gpu(xp, T)
        B = T.shape[0]
        N = xp.shape[0] // B
        out = torch.empty_like(xp)
        _lib.check(lib.mlsp_transform3_fwd_f32(xp.data_ptr(), T.data_ptr(), B, N, out.data_ptr(), _lib.stream()), "mlsp_transform3_fwd_f32")
        ctx.save_for_backward(xp, T)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        lib = _lib.load()
        xp, T = ctx.saved_tensors
        B = T.shape[0]
        N = xp.shape[0] // B
        dout = dout.contiguous()
        dx = torch.empty_like(xp) if ctx.needs_input_grad[0] else None
        dT = torch.empty_like(T)
        _lib.check(lib.mlsp_transform3_bwd_f32(xp.data_ptr(), T.data_ptr(), dout.data_ptr(), B, N, _lib.ptr(dx), dT.data_ptr(),
                                               _lib.stream()), "mlsp_transform3_bwd_f32")
        return dx, dT


def apply_transform(xp, T):
    """Models.py:113 `x = torch.matmul(T, x)` on point-major rows: xp [B*N,3], T [B,3,3] -> [B*N,3] with row p = T[b] @ x_p."""
    assert xp.shape[1] == 3 and tuple(T.shape[1:]) == (3, 3) and xp.dtype == torch.float32
    return _Transform3.apply(xp, T)


class _ComposeLinear(Function):
    @staticmethod
    def forward(ctx, Wa, ba, Wb, bb):
        lib = _lib.load()
        Wa, ba, Wb, bb = Wa.contiguous(), ba.contiguous(), Wb.contiguous(), bb.contiguous()
        _lib.require_gpu(Wa, ba, Wb, bb)
        Cm, Ci = Wa.shape
        Co = Wb.shape[0]
        W = torch.empty((Co, Ci), dtype=torch.float32, device=Wa.device)
        b = torch.empty((Co,), dtype=torch.float32, device=Wa.device)
        _lib.check(lib.mlsp_compose_linear_fwd_f32(Wa.data_ptr(), ba.data_ptr(), Wb.data_ptr(), bb.data_ptr(), Cm, Ci, Co, W.data_ptr(),
                                                   b.data_ptr(), _lib.stream()), "mlsp_compose_linear_fwd_f32")
        ctx.save_for_backward(Wa, ba, Wb)
        return W, b

    @staticmethod
    @once_differentiable
    def backward(ctx, dW, db):
        lib = _lib.load()
        Wa, ba, Wb = ctx.saved_tensors
        Cm, Ci = Wa.shape
        Co = Wb.shape[0]
        dW = dW.contiguous() if dW is not None else torch.zeros((Co, Ci), dtype=torch.float32, device=Wa.device)
        db = db.contiguous() if db is not None else torch.zeros((Co,), dtype=torch.float32, device=Wa.device)
        dWa, dba, dWb = torch.empty_like(Wa), torch.empty_like(ba), torch.empty_like(Wb)
        _lib.check(lib.mlsp_compose_linear_bwd_f32(dW.data_ptr(), db.data_ptr(), Wa.data_ptr(), ba.data_ptr(), Wb.data_ptr(), Cm, Ci, Co,
                                                   dWa.data_ptr(), dba.data_ptr(), dWb.data_ptr(), _lib.stream()), "mlsp_compose_linear_bwd_f32")
        return dWa, dba, dWb, db


def compose_linear(Wa, ba, Wb, bb):
    """conv_b(conv_a(f)) with nothing in between (PointSegDA/Models.py:176-182) as ONE linear map: -> (Wb Wa [Co,Ci], Wb ba + bb [Co]).
    One launch forward, one backward (the gradient of the composite goes back to all four parameters); fp32 fmaf chains."""
    assert Wa.dim() == 2 and Wb.dim() == 2 and Wb.shape[1] == Wa.shape[0] and ba.shape == (Wa.shape[0],) and bb.shape == (Wb.shape[0],)
    assert Wa.dtype == torch.float32 and Wb.dtype == torch.float32
    return _ComposeLinear.apply(Wa, ba, Wb, bb)


class _Chamfer(Function):
    @staticmethod
    def forward(ctx, pred, gold, mask, scale):
        lib = _lib.load()
        pred, gold, mask = pred.contiguous(), gold.contiguous().float(), mask.contiguous().float()
        _lib.require_gpu(pred, gold, mask)
        B, N, _ = pred.shape
        assert gold.shape == (B, 3, N) and mask.shape == (B, 3, N), (pred.shape, gold.shape, mask.shape)
        dev = pred.device
        per_cloud = torch.empty((B, 3), dtype=torch.float32, device=dev)
        argA = torch.empty((B, N), dtype=torch.int32, device=dev)
        argB = torch.empty((B, N), dtype=torch.int32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _lib.check(lib.mlsp_chamfer_masked_fwd_f32(pred.data_ptr(), gold.data_ptr(), mask.data_ptr(), B, N, scale,
                                                   per_cloud.data_ptr(), argA.data_ptr(), argB.data_ptr(), loss.data_ptr(),
                                                   _lib.stream()), "mlsp_chamfer_masked_fwd_f32")
        ctx.save_for_backward(pred, gold, mask, per_cloud, argA, argB)
        ctx.scale = scale
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        pred, gold, mask, per_cloud, argA, argB = ctx.saved_tensors
        B, N, _ = pred.shape
        g = g.contiguous().float()
        dpred = torch.empty_like(pred)
        _lib.check(lib.mlsp_chamfer_masked_bwd_f32(pred.data_ptr(), gold.data_ptr(), mask.data_ptr(), B, N, ctx.scale,
                                                   per_cloud.data_ptr(), argA.data_ptr(), argB.data_ptr(), g.data_ptr(),
                                                   dpred.data_ptr(), _lib.stream()), "mlsp_chamfer_masked_bwd_f32")
        return dpred, None, None, None


def chamfer_masked(pred, gold, mask, scale):
    """scale * sum_b (chamfer(gold->pred) + chamfer(pred->gold))_b / n_masked_b   (MLSP/mlsp.py:115-182)."""
    return _Chamfer.apply(pred, gold, mask, float(scale))


class _ChamferDir(Function):
    @staticmethod
    def forward(ctx, p1, p2, mask_cord):
        lib = _lib.load()
        p1, p2 = p1.contiguous().float(), p2.contiguous().float()
        mask_cord = mask_cord.contiguous().float()
        _lib.require_gpu(p1, p2, mask_cord)
        B, N, C = p1.shape
        assert C == 3 and p2.shape == (B, N, 3) and mask_cord.shape == (B, N), (p1.shape, p2.shape, mask_cord.shape)
        dev = p1.device
        per_cloud = torch.empty((B, 2), dtype=torch.float32, device=dev)
        arg = torch.empty((B, N), dtype=torch.int32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _lib.check(lib.mlsp_chamfer_dir_fwd_f32(p1.data_ptr(), p2.data_ptr(), mask_cord.data_ptr(), B, N, per_cloud.data_ptr(),
                                                arg.data_ptr(), loss.data_ptr(), _lib.stream()), "mlsp_chamfer_dir_fwd_f32")
        ctx.save_for_backward(p1, p2, mask_cord, per_cloud, arg)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        p1, p2, mask_cord, per_cloud, arg = ctx.saved_tensors
        B, N, _ = p1.shape
        g = g.contiguous().float()
        dp1 = torch.empty_like(p1) if ctx.needs_input_grad[0] else None
        dp2 = torch.empty_like(p2) if ctx.needs_input_grad[1] else None
        if dp1 is None and dp2 is None:
            return None, None, None
        _lib.check(lib.mlsp_chamfer_dir_bwd_f32(p1.data_ptr(), p2.data_ptr(), mask_cord.data_ptr(), B, N, per_cloud.data_ptr(),
                                                arg.data_ptr(), g.data_ptr(), _lib.ptr(dp1), _lib.ptr(dp2), _lib.stream()),
                   "mlsp_chamfer_dir_bwd_f32")
        return dp1, dp2, None


def chamfer_dir(p1, p2, mask_cord):
    """ONE direction of the masked Chamfer distance (MLSP/mlsp.py:115-153): p1, p2 [B,N,3], mask_cord [B,N] -> 0-dim loss;
    gradients flow to both point sets."""
    return _ChamferDir.apply(p1, p2, mask_cord)


class _NormalLoss(Function):
    @staticmethod
    def forward(ctx, pred, gt, w, weight):
        lib = _lib.load()
        pred = pred.contiguous()
        gt = gt.contiguous().float()
        _lib.require_gpu(pred, gt, w)
        P = pred.numel() // 3
        if w is not None:
            w = w.contiguous().float().reshape(-1)
            assert w.numel() == P
        out = torch.empty((2,), dtype=torch.float32, device=pred.device)
        ws, wsn = _lib.workspace(pred.device, 1, 1, 1)
        _lib.check(lib.mlsp_normal_loss_fwd_f32(pred.data_ptr(), gt.data_ptr(), _lib.ptr(w), P, weight, out.data_ptr(),
                                                ws, wsn, _lib.stream()), "mlsp_normal_loss_fwd_f32")
        ctx.save_for_backward(pred, gt, w, out)
        ctx.weight = weight
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        pred, gt, w, out = ctx.saved_tensors
        P = pred.numel() // 3
        g = g.contiguous().float()
        dpred = torch.empty_like(pred)
        _lib.check(lib.mlsp_normal_loss_bwd_f32(pred.data_ptr(), gt.data_ptr(), _lib.ptr(w), P, ctx.weight, out.data_ptr(),
                                                g.data_ptr(), dpred.data_ptr(), _lib.stream()), "mlsp_normal_loss_bwd_f32")
        return dpred, None, None, None


def normal_loss(pred, gt, w=None, weight=1.0):
    """-weight * sum w|cos(pred, gt)| / sum w   (MLSP/mlsp.py:275-287; PointDA/trainer.py:551-556)."""
    return _NormalLoss.apply(pred, gt, w, float(weight))


class _DensityTail(Function):
    @staticmethod
    def forward(ctx, logits, fc2w):
        lib = _lib.load()
        logits = logits.contiguous()
        fc2w = fc2w.contiguous().reshape(-1)
        _lib.require_gpu(logits, fc2w)
        P, nc = logits.shape
        pvec = torch.empty_like(logits)
        dens = torch.empty((P,), dtype=torch.float32, device=logits.device)
        _lib.check(lib.mlsp_density_tail_fwd_f32(logits.data_ptr(), fc2w.data_ptr(), P, nc, pvec.data_ptr(), dens.data_ptr(),
                                                 _lib.stream()), "mlsp_density_tail_fwd_f32")
        ctx.save_for_backward(pvec, fc2w)
        return pvec, dens

    @staticmethod
    @once_differentiable
    def backward(ctx, dp, dd):
        lib = _lib.load()
        pvec, fc2w = ctx.saved_tensors
        P, nc = pvec.shape
        dp = dp.contiguous() if dp is not None else None
        dd = dd.contiguous() if dd is not None else None
        dl = torch.empty_like(pvec)
        _lib.check(lib.mlsp_density_tail_bwd_f32(pvec.data_ptr(), fc2w.data_ptr(), _lib.ptr(dp), _lib.ptr(dd), P, nc,
                                                 dl.data_ptr(), _lib.stream()), "mlsp_density_tail_bwd_f32")
        return dl, None


def density_tail(logits, fc2w):
    """softmax + frozen expectation layer (PointDA/Models.py:281-285) -> (p_vec [P,nc], density [P])."""
    return _DensityTail.apply(logits, fc2w)


class _DensityLoss(Function):
    @staticmethod
    def forward(ctx, pvec, dens, tvec, target, mask, dweight):
        lib = _lib.load()
        pvec, dens = pvec.contiguous(), dens.contiguous()
        tvec, target = tvec.contiguous().float(), target.contiguous().float()
        _lib.require_gpu(pvec, dens, tvec, target, mask)
        P, nc = pvec.shape
        if mask is not None:
            mask = mask.contiguous().float().reshape(-1)
        out = torch.empty((3,), dtype=torch.float32, device=pvec.device)
        ws, wsn = _lib.workspace(pvec.device, 1, 1, 1)
        _lib.check(lib.mlsp_density_loss_fwd_f32(pvec.data_ptr(), dens.data_ptr(), tvec.data_ptr(), target.data_ptr(),
                                                 _lib.ptr(mask), P, nc, dweight, out.data_ptr(), ws, wsn, _lib.stream()),
                   "mlsp_density_loss_fwd_f32")
        ctx.save_for_backward(pvec, dens, tvec, target, mask, out)
        ctx.dweight = dweight
        return out[0], out[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, gkl, gmae):
        lib = _lib.load()
        pvec, dens, tvec, target, mask, out = ctx.saved_tensors
        P, nc = pvec.shape
        gkl = gkl.contiguous().float() if gkl is not None else None
        gmae = gmae.contiguous().float() if gmae is not None else None
        dp = torch.empty_like(pvec)
        dd = torch.empty_like(dens)
        _lib.check(lib.mlsp_density_loss_bwd_f32(pvec.data_ptr(), dens.data_ptr(), tvec.data_ptr(), target.data_ptr(),
                                                 _lib.ptr(mask), P, nc, ctx.dweight, out.data_ptr(), _lib.ptr(gkl),
                                                 _lib.ptr(gmae), dp.data_ptr(), dd.data_ptr(), _lib.stream()),
                   "mlsp_density_loss_bwd_f32")
        return dp, dd, None, None, None, None


def density_loss(pvec, dens, tvec, target, mask=None, dweight=1.0):
    """(kl, mae) of MLSP/mlsp.py:430-454."""
    return _DensityLoss.apply(pvec, dens, tvec, target, mask, float(dweight))


def gemm(A, B, ta=False, tb=False, bias=None):
    """Plain fp32 GEMM on the matrix cores: opA(A) @ opB(B) (+bias).  No autograd; used by tests."""
    lib = _lib.load()
    A, B = A.contiguous(), B.contiguous()
    _lib.require_gpu(A, B)
    M, K = (A.shape[1], A.shape[0]) if ta else A.shape
    N = B.shape[0] if tb else B.shape[1]
    C = torch.empty((M, N), dtype=torch.float32, device=A.device)
    ws, wsn = _lib.workspace(A.device, max(M, 1), max(K, 1), max(N, 1))
    _lib.check(lib.mlsp_gemm_f32(int(ta), int(tb), M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0),
                                 C.data_ptr(), N, _lib.ptr(bias), gemm_precision.code(), ws, wsn, _lib.stream()), "mlsp_gemm_f32")
    return C
