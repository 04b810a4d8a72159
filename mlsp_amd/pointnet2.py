"""PointNet++ set abstraction on MI355X (BASELINE.json configs[3]).

Mirrors the only in-tree statement of these semantics in the reference,
PointDA/hengshuang_transformer/pointnet_util.py: same function / class names, argument meaning, tensor layouts
([B,N,3] coordinates, [B,N,D] features, int64 indices) and state_dict keys (`mlp_convs.i.*`, `mlp_bns.i.*`), so code written
against that file runs unchanged.  Farthest point sampling, ball query and grouping are HIP kernels (csrc/sa.hip) with
bit-exact indices; the SA-MLP is the fused Linear+BN+ReLU kernel family over the B*S*nsample edge rows followed by the
per-centre max.  No CPU fallback: every entry point raises MlspLibraryError without the HIP library.

Not built (documented in DESIGN.md): `knn=True` grouping (argsort over a centre-to-point distance matrix),
PointNetSetAbstractionMsg / FeaturePropagation, gradients with respect to the coordinates.
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from . import functional as Fh
from .model_utils import _bn_buffers, flushing_forward


def _xyz_rows(xyz):
    assert xyz.dim() == 3 and xyz.shape[-1] >= 3 and xyz.dtype == torch.float32, (xyz.shape, xyz.dtype)
    x = xyz.detach()
    if x.stride(2) != 1 or x.stride(0) != x.shape[1] * x.stride(1):
        x = x.contiguous()
    return x


def index_points(points, idx):
    """pointnet_util.py:41-51: points [B,N,C], idx [B,S,[K]] -> [B,S,[K],C]  (torch gather: index glue, not a hot op)."""
    raw_size = idx.size()
    idx = idx.reshape(raw_size[0], -1).long()
    res = torch.gather(points, 1, idx[..., None].expand(-1, -1, points.size(-1)))
    return res.reshape(*raw_size, -1)


def farthest_point_sample(xyz, npoint, start=None):
    """pointnet_util.py:53-73.  xyz [B,N,3] -> sampled indices [B,npoint] (int64).  `start` [B] fixes the first sample
    (the reference draws it with torch.randint, :65; omitted -> drawn here the same way)."""
    lib = _lib.load()
    x = _xyz_rows(xyz)
    _lib.require_gpu(x)
    B, N, _ = x.shape
    if start is None:
        start = torch.randint(0, N, (B,), dtype=torch.long, device=x.device)
    st = start.to(device=x.device, dtype=torch.int32).contiguous()
    out = torch.empty((B, npoint), dtype=torch.int32, device=x.device)
    _lib.check(lib.mlsp_fps_f32(x.data_ptr(), x.stride(1), B, N, npoint, st.data_ptr(), out.data_ptr(), _lib.stream()),
               "mlsp_fps_f32")
    return out.long()


def query_ball_point(radius, nsample, xyz, new_xyz):
    """pointnet_util.py:76-96.  First `nsample` indices (ascending) of the points with !(d^2 > radius^2) around each centre,
    padded with the first one.  xyz [B,N,3], new_xyz [B,S,3] -> [B,S,nsample] int64."""
    lib = _lib.load()
    x, q = _xyz_rows(xyz), _xyz_rows(new_xyz)
    _lib.require_gpu(x, q)
    B, N, _ = x.shape
    S = q.shape[1]
    r2 = float(torch.tensor(radius ** 2, dtype=torch.float32))        # the reference compares in fp32 (:91)
    idx = torch.empty((B, S, nsample), dtype=torch.int32, device=x.device)
    _lib.check(lib.mlsp_ball_query_f32(x.data_ptr(), x.stride(1), q.data_ptr(), q.stride(1), B, N, S, r2, nsample, idx.data_ptr(),
                                       _lib.stream()), "mlsp_ball_query_f32")
    return idx.long()


class _Group(Function):
    """[xyz_j - new_xyz_i | feat_j] edge rows (pointnet_util.py:120-129); gradient flows to the features only."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feat, idx):
        lib = _lib.load()
        x, q = _xyz_rows(xyz), _xyz_rows(new_xyz)
        B, N, _ = x.shape
        S, ns = idx.shape[1], idx.shape[2]
        D = 0 if feat is None else feat.shape[-1]
        f = None if feat is None else feat.detach().reshape(B * N, D).contiguous()
        idx32 = idx.to(torch.int32).contiguous()
        _lib.require_gpu(x, q, idx32)
        G = torch.empty((B * S * ns, 3 + D), dtype=torch.float32, device=x.device)
        _lib.check(lib.mlsp_sa_group_fwd_f32(x.data_ptr(), x.stride(1), _lib.ptr(f), D, q.data_ptr(), q.stride(1), idx32.data_ptr(),
                                             B, N, S, ns, G.data_ptr(), _lib.stream()), "mlsp_sa_group_fwd_f32")
        ctx.save_for_backward(idx32)
        ctx.dims = (B, N, S, ns, D)
        return G

    @staticmethod
    @once_differentiable
    def backward(ctx, dG):
        lib = _lib.load()
        (idx32,) = ctx.saved_tensors
        B, N, S, ns, D = ctx.dims
        if D == 0 or not ctx.needs_input_grad[2]:
            return None, None, None, None
        dG = dG.contiguous()
        dev = dG.device
        rev_off = torch.empty((B * N + 1,), dtype=torch.int32, device=dev)
        rev_ent = torch.empty((B * S * ns,), dtype=torch.int32, device=dev)
        _lib.check(lib.mlsp_group_reverse(idx32.data_ptr(), B, S, N, ns, rev_off.data_ptr(), rev_ent.data_ptr(), _lib.stream()),
                   "mlsp_group_reverse")
        dfeat = torch.empty((B, N, D), dtype=torch.float32, device=dev)
        _lib.check(lib.mlsp_sa_group_bwd_f32(dG.data_ptr(), D, rev_off.data_ptr(), rev_ent.data_ptr(), B, N, S, ns, dfeat.data_ptr(),
                                             _lib.stream()), "mlsp_sa_group_bwd_f32")
        return None, None, dfeat, None


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, knn=False, fps_start=None):
    """pointnet_util.py:99-136: FPS centres, ball-query neighbourhoods, centred coordinates + features.
    Returns new_xyz [B,npoint,3], new_points [B,npoint,nsample,3+D] (+ grouped_xyz, fps_idx with returnfps)."""
    if knn:
        raise NotImplementedError("sample_and_group(knn=True) is not built on MI355X yet (DESIGN.md 6)")
    B, N, C = xyz.shape
    if xyz.requires_grad:
        raise NotImplementedError("gradients with respect to the coordinates are not built (the reference never needs them)")
    fps_idx = farthest_point_sample(xyz, npoint, start=fps_start)
    new_xyz = index_points(xyz, fps_idx)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    G = _Group.apply(xyz, new_xyz, points, idx)
    new_points = G.view(B, npoint, nsample, -1)
    if returnfps:
        return new_xyz, new_points, index_points(xyz, idx), fps_idx
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    """pointnet_util.py:139-156."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device)
    grouped_xyz = xyz.view(B, 1, N, C)
    new_points = torch.cat([grouped_xyz, points.view(B, 1, N, -1)], dim=-1) if points is not None else grouped_xyz
    return new_xyz, new_points


class PointNetSetAbstraction(nn.Module):
    """pointnet_util.py:159-196.  forward(xyz [B,N,3], points [B,N,D] | None) -> new_xyz [B,S,3], new_points [B,S,mlp[-1]].
    `fps_start` (attribute, optional [B] tensor) pins the first FPS sample for reproducible parity runs."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, knn=False):
        super(PointNetSetAbstraction, self).__init__()
        self.npoint = npoint
        self.radius = radius
        self.nsample = nsample
        self.knn = knn
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last_channel = out_channel
        self.group_all = group_all
        self.fps_start = None

    @flushing_forward
    def forward(self, xyz, points):
        B = xyz.shape[0]
        if self.group_all:
            new_xyz, new_points = sample_and_group_all(xyz, points)
            S, ns = 1, xyz.shape[1]
        else:
            new_xyz, new_points = sample_and_group(self.npoint, self.radius, self.nsample, xyz, points, knn=self.knn,
                                                   fps_start=self.fps_start)
            S, ns = self.npoint, self.nsample
        X = new_points.reshape(B * S * ns, new_points.shape[-1])          # edge-major rows (b, centre, slot)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            rm, rv = _bn_buffers(bn, self.training)
            X = Fh.pointmlp(X, conv.weight.view(conv.out_channels, conv.in_channels), bias=conv.bias, gamma=bn.weight,
                            beta=bn.bias, run_mean=rm, run_var=rv, training=self.training, act=Fh.ACT_RELU,
                            momentum=bn.momentum, eps=bn.eps)
        out = Fh.colmax(X, B, ns) if self.group_all else Fh.segmax(X, ns)  # max over the neighbourhood
        return new_xyz, out.view(B, S, -1)
