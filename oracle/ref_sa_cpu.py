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


def sa_forward(params, buffers, cfg, xyz, points, start, training=True, momentum=0.1, eps=1e-5):
    """PointNetSetAbstraction.forward :181-196.  params: {"mlp_convs.i.weight/bias", "mlp_bns.i.weight/bias"};
    buffers: {"mlp_bns.i.running_mean/var"}.  Returns new_xyz, new_points, fps_idx, group_idx, new_buffers."""
    B, N, _ = xyz.shape
    new_buffers = dict(buffers)
    if cfg["group_all"]:
        new_xyz = torch.zeros(B, 1, 3)
        grouped = xyz.view(B, 1, N, 3)
        feat = torch.cat([grouped, points.view(B, 1, N, -1)], -1) if points is not None else grouped
        fps_idx = gidx = None
    else:
        fps_idx = fps(xyz, cfg["npoint"], start)
        new_xyz = gather_rows(xyz, fps_idx)
        gidx = ball_query(cfg["radius"], cfg["nsample"], xyz, new_xyz)
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
    out = h.max(2)[0].transpose(1, 2)
    return new_xyz, out, fps_idx, gidx, new_buffers
