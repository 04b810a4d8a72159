"""PointNet++ set abstraction on MI355X (BASELINE.json configs[3]).

Mirrors the only in-tree statement of these semantics in the reference,
PointDA/hengshuang_transformer/pointnet_util.py: same function / class names, argument meaning, tensor layouts
([B,N,3] coordinates, [B,N,D] features, int64 indices) and state_dict keys (`mlp_convs.i.*`, `mlp_bns.i.*`), so code written
against that file runs unchanged.  Farthest point sampling, ball query and grouping are HIP kernels (csrc/sa.hip) with
bit-exact indices; the SA-MLP is the fused Linear+BN+ReLU kernel family over the B*S*nsample edge rows followed by the
per-centre max.  No CPU fallback: every entry point raises MlspLibraryError without the HIP library.

`knn=True` grouping, PointNetSetAbstractionMsg and PointNetFeaturePropagation run on the query-kNN kernel (ref != query,
mlsp_knn_query_f32) and the 3-NN interpolation kernels.  Not built: gradients with respect to the coordinates (the reference never
asks for them: coordinates are inputs).
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


def knn_point(k, xyz, new_xyz, return_dist=False):
    """`square_distance(new_xyz, xyz).argsort()[:, :, :k]` (pointnet_util.py:26-38,116-118) without the [B,S,N] matrix:
    xyz [B,N,3] reference points, new_xyz [B,S,3] queries -> idx [B,S,k] int64, nearest first (ties -> lower index)."""
    lib = _lib.load()
    x, q = _xyz_rows(xyz), _xyz_rows(new_xyz)
    _lib.require_gpu(x, q)
    B, N, C = x.shape
    S = q.shape[1]
    idx = torch.empty((B, S, k), dtype=torch.int32, device=x.device)
    dist = torch.empty((B, S, k), dtype=torch.float32, device=x.device) if return_dist else None
    _lib.check(lib.mlsp_knn_query_f32(x.data_ptr(), x.stride(1), N, q.data_ptr(), q.stride(1), S, B, min(C, 8), k, idx.data_ptr(),
                                      _lib.ptr(dist), _lib.stream()), "mlsp_knn_query_f32")
    return (idx.long(), dist) if return_dist else idx.long()


class _Interp3(Function):
    """Inverse-distance interpolation from the three nearest sampled points (pointnet_util.py:287-294)."""

    @staticmethod
    def forward(ctx, feat, idx32, dist):
        lib = _lib.load()
        feat = feat.contiguous()
        B, S, D = feat.shape
        N = idx32.shape[1]
        out = torch.empty((B, N, D), dtype=torch.float32, device=feat.device)
        _lib.check(lib.mlsp_interp3_fwd_f32(feat.data_ptr(), idx32.data_ptr(), dist.data_ptr(), B, N, S, D, out.data_ptr(), _lib.stream()),
                   "mlsp_interp3_fwd_f32")
        ctx.save_for_backward(idx32, dist)
        ctx.dims = (B, N, S, D)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        lib = _lib.load()
        idx32, dist = ctx.saved_tensors
        B, N, S, D = ctx.dims
        dout = dout.contiguous()
        dev = dout.device
        rev_off = torch.empty((B * S + 1,), dtype=torch.int32, device=dev)
        rev_ent = torch.empty((B * N * 3,), dtype=torch.int32, device=dev)
        _lib.check(lib.mlsp_group_reverse(idx32.data_ptr(), B, N, S, 3, rev_off.data_ptr(), rev_ent.data_ptr(), _lib.stream()),
                   "mlsp_group_reverse")
        dfeat = torch.empty((B, S, D), dtype=torch.float32, device=dev)
        _lib.check(lib.mlsp_interp3_bwd_f32(dout.data_ptr(), dist.data_ptr(), rev_off.data_ptr(), rev_ent.data_ptr(), B, N, S, D,
                                            dfeat.data_ptr(), _lib.stream()), "mlsp_interp3_bwd_f32")
        return dfeat, None, None


class _Group(Function):
    """[xyz_j - new_xyz_i | feat_j] edge rows (pointnet_util.py:120-129).  Gradients: features and both coordinate operands (the
    reference gets them through autograd of `grouped_xyz - new_xyz`, :122-124; the neighbour indices carry none)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feat, idx):
        lib = _lib.load()
        # (the reference subtracts every coordinate channel; its callers pass xyz [B, N, 3]: wider coordinates are not implemented)
        assert xyz.shape[-1] == 3 and new_xyz.shape[-1] == 3, "grouping takes 3-column coordinates"
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
        ctx.dims = (B, N, S, ns, D, xyz.shape[-1], new_xyz.shape[-1])
        return G

    @staticmethod
    @once_differentiable
    def backward(ctx, dG):
        lib = _lib.load()
        (idx32,) = ctx.saved_tensors
        B, N, S, ns, D, Cx, Cq = ctx.dims
        need_x, need_q, need_f = ctx.needs_input_grad[0], ctx.needs_input_grad[1], D > 0 and ctx.needs_input_grad[2]
        if not (need_x or need_q or need_f):
            return None, None, None, None
        dG = dG.contiguous()
        dev = dG.device
        dxyz = dq = dfeat = None
        if need_x or need_f:
            rev_off = torch.empty((B * N + 1,), dtype=torch.int32, device=dev)
            rev_ent = torch.empty((B * S * ns,), dtype=torch.int32, device=dev)
            _lib.check(lib.mlsp_group_reverse(idx32.data_ptr(), B, S, N, ns, rev_off.data_ptr(), rev_ent.data_ptr(), _lib.stream()),
                       "mlsp_group_reverse")
        if need_f:
            dfeat = torch.empty((B, N, D), dtype=torch.float32, device=dev)
            _lib.check(lib.mlsp_sa_group_bwd_f32(dG.data_ptr(), 3 + D, 3, D, rev_off.data_ptr(), rev_ent.data_ptr(), B, N, S, ns,
                                                 dfeat.data_ptr(), _lib.stream()), "mlsp_sa_group_bwd_f32")
        if need_x:
            # d xyz_j = sum of the coordinate columns over the groups j sits in: the same reverse-index gather on columns 0..2 of dG, in place
            dxyz = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
            _lib.check(lib.mlsp_sa_group_bwd_f32(dG.data_ptr(), 3 + D, 0, 3, rev_off.data_ptr(), rev_ent.data_ptr(), B, N, S, ns,
                                                 dxyz.data_ptr(), _lib.stream()), "mlsp_sa_group_bwd_f32")
        if need_q:
            dq = -dG[:, :3].reshape(B, S, ns, 3).sum(2)       # every slot of a group subtracts its centre
        return dxyz, dq, dfeat, None


class _SAFold(Function):
    """BatchNorm + ReLU of the folded first SA layer on the edges: Z[(b,i,s)] = relu(BN(u[b, idx[b,i,s]] - w[b,i]))  (mlsp_sa_fold_*_f32);
    u [B*N,C] per source point, w [B*S,C] per centre, idx int32 [B,S,ns].  Gradients: du, dw, dgamma, dbeta."""

    @staticmethod
    def forward(ctx, u, w, idx32, gamma, beta, run_mean, run_var, training, momentum, eps, dims):
        lib = _lib.load()
        B, N, S, ns = dims
        u, w = u.contiguous(), w.contiguous()
        _lib.require_gpu(u, w, idx32)
        C = u.shape[1]
        dev = u.device
        Z = torch.empty((B * S * ns, C), dtype=torch.float32, device=dev)
        bn_save = torch.empty((4, C), dtype=torch.float32, device=dev)
        ws, wsn = _lib.workspace(dev, B * S * ns, C, C)
        _lib.check(lib.mlsp_sa_fold_fwd_f32(u.data_ptr(), w.data_ptr(), idx32.data_ptr(), B, N, S, ns, C, gamma.data_ptr(), beta.data_ptr(),
                                            _lib.ptr(run_mean), _lib.ptr(run_var), float(momentum), float(eps), int(training), Z.data_ptr(),
                                            bn_save.data_ptr(), ws, wsn, _lib.stream()), "mlsp_sa_fold_fwd_f32")
        ctx.save_for_backward(u, w, idx32, bn_save)
        ctx.dims, ctx.training = dims, bool(training)
        return Z

    @staticmethod
    @once_differentiable
    def backward(ctx, dZ):
        lib = _lib.load()
        u, w, idx32, bn_save = ctx.saved_tensors
        B, N, S, ns = ctx.dims
        C = u.shape[1]
        dev = dZ.device
        dZ = dZ.contiguous()
        rev_off = torch.empty((B * N + 1,), dtype=torch.int32, device=dev)
        rev_cnt = torch.empty((B * N,), dtype=torch.int32, device=dev)
        rev_ent = torch.empty((B * S * ns,), dtype=torch.int32, device=dev)
        pad_cnt = torch.empty((B * S,), dtype=torch.int32, device=dev)
        # the padding slots of the ball-query groups (copies of the first hit, most of the slots at these radii) stay out of the lists
        _lib.check(lib.mlsp_group_reverse_compact(idx32.data_ptr(), B, S, N, ns, rev_off.data_ptr(), rev_cnt.data_ptr(), rev_ent.data_ptr(),
                                                  pad_cnt.data_ptr(), _lib.stream()), "mlsp_group_reverse_compact")
        du, dw = torch.empty_like(u), torch.empty_like(w)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
        ws, wsn = _lib.workspace(dev, B * S * ns, C, C)
        _lib.check(lib.mlsp_sa_fold_bwd_f32(dZ.data_ptr(), u.data_ptr(), w.data_ptr(), idx32.data_ptr(), rev_off.data_ptr(), rev_ent.data_ptr(),
                                            rev_cnt.data_ptr(), pad_cnt.data_ptr(), B, N, S, ns, C, bn_save.data_ptr(), int(ctx.training), du.data_ptr(), dw.data_ptr(),
                                            dgamma.data_ptr(), dbeta.data_ptr(), ws, wsn, _lib.stream()), "mlsp_sa_fold_bwd_f32")
        return du, dw, None, dgamma, dbeta, None, None, None, None, None, None


_FOLD_CHANNELS = (16, 32, 64, 128, 256)


def _fold_supported(conv, nsample):
    return conv.out_channels in _FOLD_CHANNELS and nsample <= 256


def _fold_first_layer(xyz, new_xyz, points, idx, W0, conv, bn, training):
    """First Conv2d(1x1) + BatchNorm2d + ReLU of a set-abstraction branch WITHOUT the grouped tensor:
    W0 [x_j - c_i ; f_j] + b = u_j - w_i,  u = [x | f] W0^T + b per source point, w = c W0[:, :3]^T per centre (W0 columns ordered
    [xyz | features], pointnet_util.py:124-129).  -> activated edge rows [B*S*ns, C]."""
    B, N, _ = xyz.shape
    S, ns = idx.shape[1], idx.shape[2]
    Wx = W0[:, :3]
    u = Fh.pointmlp(xyz.reshape(B * N, 3), Wx, bias=conv.bias)            # K = 3: the streaming thin-GEMM kernels
    if points is not None:
        # the feature half as its own GEMM with an aligned, interior-tile shape (K = D) instead of one ragged K = 3 + D product on the
        # predicated kernel: [x | f] W0^T = x Wx^T + f Wf^T
        D = points.shape[-1]
        u = u + Fh.pointmlp(points.reshape(B * N, D), W0[:, 3:].contiguous())
    w = Fh.pointmlp(new_xyz.reshape(B * S, 3).contiguous(), Wx)
    rm, rv = _bn_buffers(bn, training)
    idx32 = idx.to(torch.int32).contiguous()
    return _SAFold.apply(u, w, idx32, bn.weight, bn.bias, rm, rv, training, bn.momentum, bn.eps, (B, N, S, ns))


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, knn=False, fps_start=None):
    """pointnet_util.py:99-136: FPS centres, ball-query neighbourhoods, centred coordinates + features.
    Returns new_xyz [B,npoint,3], new_points [B,npoint,nsample,3+D] (+ grouped_xyz, fps_idx with returnfps)."""
    B, N, C = xyz.shape
    fps_idx = farthest_point_sample(xyz, npoint, start=fps_start)
    new_xyz = index_points(xyz, fps_idx)
    idx = knn_point(nsample, xyz, new_xyz) if knn else query_ball_point(radius, nsample, xyz, new_xyz)      # :116-120
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
        self.fold_first = True          # fold the first conv onto the points (no grouped tensor); False: the materialised path
        self.fuse_max = True            # last conv + BN + ReLU + neighbourhood max in one op (no activated edge tensor); False: separate max

    @flushing_forward
    def forward(self, xyz, points):
        B = xyz.shape[0]
        if self.group_all:
            new_xyz, new_points = sample_and_group_all(xyz, points)
            S, ns = 1, xyz.shape[1]
        elif self.fold_first and _fold_supported(self.mlp_convs[0], self.nsample) and len(self.mlp_convs) > 1:
            # sample_and_group (pointnet_util.py:99-136) without its output tensor: the first conv is folded onto the points (the
            # coordinates' gradient flows through the two K = 3 products u and w of _fold_first_layer, like the features')
            S, ns = self.npoint, self.nsample
            fps_idx = farthest_point_sample(xyz, S, start=self.fps_start)
            new_xyz = index_points(xyz, fps_idx)
            idx = knn_point(ns, xyz, new_xyz) if self.knn else query_ball_point(self.radius, ns, xyz, new_xyz)
            c0 = self.mlp_convs[0]
            X = _fold_first_layer(xyz, new_xyz, points, idx, c0.weight.view(c0.out_channels, c0.in_channels), c0, self.mlp_bns[0],
                                  self.training)
            out = _sa_mlp(X, self.mlp_convs[1:], self.mlp_bns[1:], self.training, max_over=ns if self.fuse_max else 0)
            return new_xyz, (out if self.fuse_max else _neigh_max(out, ns)).view(B, S, -1)
        else:
            new_xyz, new_points = sample_and_group(self.npoint, self.radius, self.nsample, xyz, points, knn=self.knn,
                                                   fps_start=self.fps_start)
            S, ns = self.npoint, self.nsample
        X = new_points.reshape(B * S * ns, new_points.shape[-1])          # edge-major rows (b, centre, slot)
        first_weight = None
        Kin = X.shape[1]
        if self.group_all and Kin % 128 != 0 and Kin > 32 and X.shape[0] % 128 == 0:
            # a ragged channel count (3 + D = 259) keeps the first layer's three GEMMs on the predicated kernel: zero-pad the input rows
            # and the weight columns to the next multiple of 128 (exact: the padding multiplies zeros)
            Kp = (Kin + 127) // 128 * 128
            X = torch.nn.functional.pad(X, (0, Kp - Kin))
            c0 = self.mlp_convs[0]
            first_weight = torch.nn.functional.pad(c0.weight.view(c0.out_channels, c0.in_channels), (0, Kp - Kin))
        if self.fuse_max and ns <= 255:
            out = _sa_mlp(X, self.mlp_convs, self.mlp_bns, self.training, first_weight=first_weight, max_over=ns)
        else:
            X = _sa_mlp(X, self.mlp_convs, self.mlp_bns, self.training, first_weight=first_weight)
            out = Fh.colmax(X, B, ns) if self.group_all else _neigh_max(X, ns)  # max over the neighbourhood
        return new_xyz, out.view(B, S, -1)


def _sa_mlp(X, convs, bns, training, first_weight=None, max_over=0):
    """Conv2d(1x1)+BN2d+ReLU stack of a set-abstraction branch on edge-major rows [B*S*ns, Cin].  max_over = k > 0: followed by the max
    over every k consecutive rows (the neighbourhood max, pointnet_util.py:195), fused into the last layer where the shape allows --
    returns [rows // k, C]."""
    n = len(convs)
    for i, (conv, bn) in enumerate(zip(convs, bns)):
        rm, rv = _bn_buffers(bn, training)
        W = first_weight if (i == 0 and first_weight is not None) else conv.weight.view(conv.out_channels, conv.in_channels)
        if max_over and i == n - 1 and Fh.pointmlp_segmax_supported(X.shape[0], conv.out_channels, max_over):
            return Fh.pointmlp_segmax(X, W, max_over, bias=conv.bias, gamma=bn.weight, beta=bn.bias, run_mean=rm, run_var=rv,
                                      training=training, act=Fh.ACT_RELU, momentum=bn.momentum, eps=bn.eps)
        X = Fh.pointmlp(X, W, bias=conv.bias, gamma=bn.weight, beta=bn.bias, run_mean=rm, run_var=rv, training=training,
                        act=Fh.ACT_RELU, momentum=bn.momentum, eps=bn.eps)
    return _neigh_max(X, max_over) if max_over else X


def _neigh_max(X, k):
    """max over every k consecutive rows: the uint8-slot segmax kernel up to k = 255, the int32-row column max beyond."""
    return Fh.segmax(X, k) if k <= 255 else Fh.colmax(X, X.shape[0] // k, k)


class PointNetSetAbstractionMsg(nn.Module):
    """pointnet_util.py:199-258: multi-scale grouping -- one (radius, nsample, MLP) branch per scale around the same FPS centres,
    outputs concatenated on the channel axis.  forward(xyz [B,N,3], points [B,N,D] | None, seed_idx=None) -> new_xyz [B,S,3],
    new_points [B,S,sum(mlp[-1])].  The reference feeds its first conv [features | centred xyz] (:246) while the grouping kernel
    emits [centred xyz | features]: the first layer's weight columns are permuted accordingly (same parameters, same result)."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list, knn=False):
        super(PointNetSetAbstractionMsg, self).__init__()
        self.npoint = npoint
        self.radius_list = radius_list
        self.nsample_list = nsample_list
        self.knn = knn
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        for i in range(len(mlp_list)):
            convs, bns = nn.ModuleList(), nn.ModuleList()
            last_channel = in_channel + 3
            for out_channel in mlp_list[i]:
                convs.append(nn.Conv2d(last_channel, out_channel, 1))
                bns.append(nn.BatchNorm2d(out_channel))
                last_channel = out_channel
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)
        self.fps_start = None
        self.fold_first = True

    @flushing_forward
    def forward(self, xyz, points, seed_idx=None):
        B, N, C = xyz.shape
        S = self.npoint
        fps_idx = farthest_point_sample(xyz, S, start=self.fps_start) if seed_idx is None else seed_idx
        new_xyz = index_points(xyz, fps_idx)
        outs = []
        for i, radius in enumerate(self.radius_list):
            K = self.nsample_list[i]
            idx = knn_point(K, xyz, new_xyz) if self.knn else query_ball_point(radius, K, xyz, new_xyz)
            c0 = self.conv_blocks[i][0]
            W0 = c0.weight.view(c0.out_channels, c0.in_channels)
            if points is not None:
                D = points.shape[-1]
                W0 = torch.cat((W0[:, D:], W0[:, :D]), dim=1)                        # reference column order: [feat | xyz]
            if self.fold_first and _fold_supported(c0, K) and len(self.conv_blocks[i]) > 1:
                X = _fold_first_layer(xyz, new_xyz, points, idx, W0, c0, self.bn_blocks[i][0], self.training)
                X = _sa_mlp(X, self.conv_blocks[i][1:], self.bn_blocks[i][1:], self.training, max_over=K)
            else:
                G = _Group.apply(xyz, new_xyz, points, idx)                          # rows [xyz_j - c_i | feat_j]
                X = _sa_mlp(G, self.conv_blocks[i], self.bn_blocks[i], self.training, first_weight=W0, max_over=K)
            outs.append(X.view(B, S, -1))
        return new_xyz, torch.cat(outs, dim=-1)


class PointNetFeaturePropagation(nn.Module):
    """pointnet_util.py:261-310.  forward(xyz1 [B,3,N], xyz2 [B,3,S], points1 [B,D1,N] | None, points2 [B,D2,S]) -> [B,mlp[-1],N]
    (channel-major in and out, as the reference: "this function swaps N and C")."""

    def __init__(self, in_channel, mlp):
        super(PointNetFeaturePropagation, self).__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last_channel = out_channel

    @flushing_forward
    def forward(self, xyz1, xyz2, points1, points2):
        x1 = xyz1.permute(0, 2, 1).contiguous()
        x2 = xyz2.permute(0, 2, 1).contiguous()
        p2 = points2.permute(0, 2, 1).contiguous()
        B, N, _ = x1.shape
        S = x2.shape[1]
        if S == 1:
            interp = p2.repeat(1, N, 1)
        else:
            if S < 3:
                raise ValueError("feature propagation needs at least three sampled points (or exactly one)")
            idx, dist = knn_point(3, x2, x1, return_dist=True)                       # three nearest sampled points of every point
            interp = _Interp3.apply(p2, idx.to(torch.int32).contiguous(), dist)
        new = torch.cat([points1.permute(0, 2, 1), interp], dim=-1) if points1 is not None else interp
        X = new.reshape(B * N, -1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            rm, rv = _bn_buffers(bn, self.training)
            X = Fh.pointmlp(X, conv.weight.view(conv.out_channels, conv.in_channels), bias=conv.bias, gamma=bn.weight, beta=bn.bias,
                            run_mean=rm, run_var=rv, training=self.training, act=Fh.ACT_RELU, momentum=bn.momentum, eps=bn.eps)
        return X.view(B, N, -1).permute(0, 2, 1)
