"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the PointNet++ set-abstraction row.

Restates, in plain torch on the CPU, the algorithm of the reference's
PointDA/hengshuang_transformer/pointnet_util.py -- farthest_point_sample (:53-73), query_ball_point (:76-96),
sample_and_group (:99-136) and PointNetSetAbstraction.forward (:181-196) -- as pure functions of explicit parameters.
Pinned to the reference itself: tests/golden/sa_*.npz hold inputs and outputs captured by importing that file in the
build container (tools/make_golden.py::sa_case); tests/test_oracle_golden.py checks this restatement against them.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import torch
import torch.nn.functional as F


def fps(xyz, npoint, start):
    """pointnet_util.py:53-73 with the random first index (:65) passed in.  xyz [B,N,3] -> [B,npoint] int64."""
    B, N, _ = xyz.shape
    out = torch.empty(B, npoint, dtype=torch.long)
    running = torch.full((B, N), 1e10)
    cur = start.clone().long()
    rows = torch.arange(B)
    for i in range(npoint):
        out[:, i] = cur
        c = xyz[rows, cur].unsqueeze(1)                       # [B,1,3]
        d = ((xyz - c) ** 2).sum(-1)                          # :69 same expression, same order
        running = torch.minimum(running, d)
        cur = running.argmax(-1)                              # :71 torch.max(...)[1]: first maximum on the CPU
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    """pointnet_util.py:76-96: ascending indices of the points with !(d2 > r^2), first nsample, padded with the first."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    d2 = ((new_xyz[:, :, None] - xyz[:, None]) ** 2).sum(-1)  # square_distance :22-38
    inside = ~(d2 > radius ** 2)
    rank = inside.long().cumsum(-1) - 1                       # position of every inside point in index order
    idx = torch.zeros(B, S, nsample, dtype=torch.long)
    filled = torch.zeros(B, S, nsample, dtype=torch.bool)
    b, s, j = torch.nonzero(inside & (rank < nsample), as_tuple=True)
    idx[b, s, rank[b, s, j]] = j
    filled[b, s, rank[b, s, j]] = True
    first = idx[:, :, :1].expand(-1, -1, nsample)
    return torch.where(filled, idx, first)


def gather_rows(points, idx):
    """index_points :41-51."""
    B = points.shape[0]
    flat = idx.reshape(B, -1)
    return torch.gather(points, 1, flat[..., None].expand(-1, -1, points.shape[-1])).reshape(*idx.shape, -1)


def sa_forward(params, buffers, cfg, xyz, points, start, training=True, momentum=0.1, eps=1e-5, sel=None, sel_out=None, groups=None):
    """PointNetSetAbstraction.forward :181-196.  params: {"mlp_convs.i.weight/bias", "mlp_bns.i.weight/bias"};
    buffers: {"mlp_bns.i.running_mean/var"}.  Returns new_xyz, new_points, fps_idx, group_idx, new_buffers.
    Test hooks for gradient pins (as oracle/ref_cpu.py's ctx.sel): sel_out (a list) receives the neighbourhood max's arg-max slots
    [B, S, C]; sel (such a tensor) FORCES the pooling to those slots (a gather), so that two precisions route every gradient alike;
    groups = (fps_idx, group_idx) forces the sampling and grouping (a float64 run decides a handful of radius-boundary neighbours
    differently from fp32: the discrete choices of the fp32 run are then imposed on it)."""
    B, N, _ = xyz.shape
    new_buffers = dict(buffers)
    if cfg["group_all"]:
        new_xyz = torch.zeros(B, 1, 3)
        grouped = xyz.view(B, 1, N, 3)
        feat = torch.cat([grouped, points.view(B, 1, N, -1)], -1) if points is not None else grouped
        fps_idx = gidx = None
    else:
        fps_idx = fps(xyz, cfg["npoint"], start) if groups is None else groups[0]
        new_xyz = gather_rows(xyz, fps_idx)
        gidx = ball_query(cfg["radius"], cfg["nsample"], xyz, new_xyz) if groups is None else groups[1]
        centred = gather_rows(xyz, gidx) - new_xyz[:, :, None]
        feat = torch.cat([centred, gather_rows(points, gidx)], -1) if points is not None else centred
    h = feat.permute(0, 3, 2, 1)                              # [B, C, nsample, S]
    for i in range(len(cfg["mlp"])):
        w, b = params["mlp_convs.%d.weight" % i], params["mlp_convs.%d.bias" % i]
        h = F.conv2d(h, w, b)
        rm = new_buffers["mlp_bns.%d.running_mean" % i].clone()
        rv = new_buffers["mlp_bns.%d.running_var" % i].clone()
        h = F.batch_norm(h, rm, rv, params["mlp_bns.%d.weight" % i], params["mlp_bns.%d.bias" % i], training, momentum, eps)
        new_buffers["mlp_bns.%d.running_mean" % i], new_buffers["mlp_bns.%d.running_var" % i] = rm, rv
        h = F.relu(h)
    if sel is not None:
        out = torch.gather(h, 2, sel.permute(0, 2, 1).unsqueeze(2)).squeeze(2).transpose(1, 2)
    else:
        val, arg = h.max(2)
        out = val.transpose(1, 2)
        if sel_out is not None:
            sel_out.append(arg.permute(0, 2, 1).contiguous())
    return new_xyz, out, fps_idx, gidx, new_buffers


# --------------------------------------------------------------------------- round 2: knn grouping, multi-scale grouping, feature propagation
def square_distance(src, dst):
    """pointnet_util.py:22-38: -2 src.dst^T + |src|^2 + |dst|^2, in that association."""
    d = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    d = d + torch.sum(src ** 2, -1).unsqueeze(-1)
    return d + torch.sum(dst ** 2, -1).unsqueeze(1)


def knn_group(k, xyz, new_xyz):
    """pointnet_util.py:116-118 (knn=True): the k nearest points of every centre, nearest first (ties -> lower index)."""
    d = square_distance(new_xyz, xyz)
    return torch.sort(d, dim=-1, stable=True)[1][:, :, :k]


def _conv_bn_relu(h, params, new_buffers, conv_key, bn_key, training, momentum, eps, conv):
    h = conv(h, params[conv_key + ".weight"], params[conv_key + ".bias"])
    rm, rv = new_buffers[bn_key + ".running_mean"].clone(), new_buffers[bn_key + ".running_var"].clone()
    h = F.batch_norm(h, rm, rv, params[bn_key + ".weight"], params[bn_key + ".bias"], training, momentum, eps)
    new_buffers[bn_key + ".running_mean"], new_buffers[bn_key + ".running_var"] = rm, rv
    return F.relu(h)


def sa_msg_forward(params, buffers, cfg, xyz, points, fps_idx, training=True, momentum=0.1, eps=1e-5):
    """PointNetSetAbstractionMsg.forward (pointnet_util.py:220-258) with the FPS indices passed in (`seed_idx`).
    cfg: {"radius_list", "nsample_list", "mlp_list", "knn"}.  Returns new_xyz, new_points [B,S,sum], group indices, new_buffers."""
    B, N, C = xyz.shape
    S = fps_idx.shape[1]
    new_buffers = dict(buffers)
    new_xyz = gather_rows(xyz, fps_idx)
    outs, gidxs = [], []
    for i, radius in enumerate(cfg["radius_list"]):
        K = cfg["nsample_list"][i]
        gidx = knn_group(K, xyz, new_xyz) if cfg.get("knn") else ball_query(radius, K, xyz, new_xyz)
        gidxs.append(gidx)
        gx = gather_rows(xyz, gidx) - new_xyz.view(B, S, 1, C)
        g = torch.cat([gather_rows(points, gidx), gx], -1) if points is not None else gx       # features first (:246)
        h = g.permute(0, 3, 2, 1)
        for j in range(len(cfg["mlp_list"][i])):
            h = _conv_bn_relu(h, params, new_buffers, "conv_blocks.%d.%d" % (i, j), "bn_blocks.%d.%d" % (i, j), training, momentum, eps, F.conv2d)
        outs.append(h.max(2)[0])
    return new_xyz, torch.cat(outs, 1).transpose(1, 2), gidxs, new_buffers


def fp_forward(params, buffers, mlp, xyz1, xyz2, points1, points2, training=True, momentum=0.1, eps=1e-5):
    """PointNetFeaturePropagation.forward (pointnet_util.py:272-310); channel-major inputs/outputs like the reference."""
    x1, x2, p2 = xyz1.permute(0, 2, 1), xyz2.permute(0, 2, 1), points2.permute(0, 2, 1)
    B, N, _ = x1.shape
    S = x2.shape[1]
    new_buffers = dict(buffers)
    if S == 1:
        interp = p2.repeat(1, N, 1)
    else:
        d, idx = torch.sort(square_distance(x1, x2), dim=-1, stable=True)
        d, idx = d[:, :, :3], idx[:, :, :3]
        recip = 1.0 / (d + 1e-8)
        w = recip / recip.sum(2, keepdim=True)
        interp = (gather_rows(p2, idx) * w.view(B, N, 3, 1)).sum(2)
    h = torch.cat([points1.permute(0, 2, 1), interp], -1) if points1 is not None else interp
    h = h.permute(0, 2, 1)
    for i in range(len(mlp)):
        h = _conv_bn_relu(h, params, new_buffers, "mlp_convs.%d" % i, "mlp_bns.%d" % i, training, momentum, eps, F.conv1d)
    return h, new_buffers
