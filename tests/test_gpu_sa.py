"""PointNet++ set abstraction on the GPU (mlsp_amd/pointnet2.py + csrc/sa.hip) against the reference's golden vectors and
the CPU oracle: FPS and ball-query indices bit-exact, layer outputs within 1e-3, gradients and BN buffers close."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_sa_cpu as sa
from test_sa_oracle_cpu import GOLD, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_fps_and_ball_query_bit_exact_config3_scale(dev):
    from mlsp_amd import pointnet2 as p2
    g = dict(np.load(os.path.join(GOLD, "sa_idx_s3_B2_N2048.npz")))
    xyz = torch.from_numpy(g["xyz"]).to(dev)
    fps_idx = p2.farthest_point_sample(xyz, g["fps_idx"].shape[1], start=torch.from_numpy(g["start"]))
    assert np.array_equal(fps_idx.cpu().numpy(), g["fps_idx"].astype(np.int64))
    gidx = p2.query_ball_point(float(g["radius"]), int(g["nsample"]), xyz, p2.index_points(xyz, fps_idx))
    assert np.array_equal(gidx.cpu().numpy(), g["group_idx"].astype(np.int64))


@pytest.mark.parametrize("B,N,S,radius,ns", [(3, 100, 20, 0.5, 8), (2, 1024, 256, 0.15, 64), (1, 4096, 1024, 0.1, 32), (5, 70, 70, 3.0, 40)])
def test_fps_and_ball_query_vs_oracle(dev, B, N, S, radius, ns):
    from mlsp_amd import pointnet2 as p2
    g = torch.Generator().manual_seed(N + S)
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    if N == 70:
        xyz[:, 10:30] = xyz[:, 10:11]                       # duplicates: equal distances, ties resolved by index
    start = torch.randint(0, N, (B,), generator=g)
    want = sa.fps(xyz, S, start)
    got = p2.farthest_point_sample(xyz.to(dev), S, start=start).cpu()
    assert torch.equal(got, want)
    new_xyz = sa.gather_rows(xyz, want)
    want_idx = sa.ball_query(radius, ns, xyz, new_xyz)
    got_idx = p2.query_ball_point(radius, ns, xyz.to(dev), new_xyz.to(dev)).cpu()
    assert torch.equal(got_idx, want_idx)


@pytest.mark.parametrize("name", ["sa_s0_B4_N256.npz", "sa_s1_B3_N200_nofeat.npz", "sa_s2_B4_N128_all.npz"])
def test_set_abstraction_layer_vs_reference_golden(dev, name):
    from mlsp_amd import pointnet2 as p2
    g, cfg, params, buffers = load_case(name)
    layer = p2.PointNetSetAbstraction(cfg["npoint"] or None, cfg["radius"] or None, cfg["nsample"] or None, 3 + cfg["D"], cfg["mlp"],
                                      cfg["group_all"])
    sd = {k[len("state_before/"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("state_before/")}
    layer.load_state_dict(sd)                               # same state_dict keys as the reference module
    layer.to(dev).train()
    xyz = torch.from_numpy(g["xyz"]).to(dev)
    points = torch.from_numpy(g["points"]).to(dev).requires_grad_(True) if cfg["D"] else None
    if "start" in g:
        layer.fps_start = torch.from_numpy(g["start"])
    new_xyz, out = layer(xyz, points)
    np.testing.assert_allclose(new_xyz.cpu().numpy(), g["new_xyz"], rtol=0, atol=0)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["new_points"], rtol=1e-3, atol=1e-3)     # north-star tolerance
    assert np.abs(out.detach().cpu().numpy() - g["new_points"]).max() < 2e-4
    (out * torch.from_numpy(g["wgt"]).to(dev)).sum().backward()
    if cfg["D"]:
        err = np.abs(points.grad.cpu().numpy() - g["d_points"]).max() / (np.abs(g["d_points"]).max() + 1e-9)
        assert err < 2e-3, err
    for k, p in layer.named_parameters():
        want = g["grad/" + k]
        if "mlp_convs" in k and k.endswith("bias"):
            continue                                        # a bias in front of a batch-stat BN: analytically zero gradient
        err = np.abs(p.grad.cpu().numpy() - want).max() / (np.abs(want).max() + 1e-9)
        assert err < 5e-3, (k, err)
    after = layer.state_dict()
    for k, v in after.items():
        np.testing.assert_allclose(v.cpu().numpy(), g["state_after/" + k], rtol=1e-4, atol=1e-5, err_msg=k)


def test_set_abstraction_stack_config3_shape(dev):
    """BASELINE.json configs[3] shape (B=32, N=2048): two stacked SA layers + a group-all layer run end to end; the result
    is invariant to the order of the features' rows inside each neighbourhood max (recomputed from the grouped tensor)."""
    from mlsp_amd import pointnet2 as p2
    torch.manual_seed(0)
    B, N = 32, 2048
    xyz = (torch.rand(B, N, 3) * 2 - 1).to(dev)
    sa1 = p2.PointNetSetAbstraction(512, 0.2, 32, 3, [64, 64, 128], False).to(dev)
    sa2 = p2.PointNetSetAbstraction(128, 0.4, 64, 128 + 3, [128, 128, 256], False).to(dev)
    sa3 = p2.PointNetSetAbstraction(None, None, None, 256 + 3, [256, 512, 1024], True).to(dev)
    l1_xyz, l1 = sa1(xyz, None)
    l2_xyz, l2 = sa2(l1_xyz, l1)
    _, l3 = sa3(l2_xyz, l2)
    assert l1.shape == (B, 512, 128) and l2.shape == (B, 128, 256) and l3.shape == (B, 1, 1024)
    l3.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in sa1.parameters())
    # every sampled centre is its own first in-radius neighbour candidate; FPS indices are unique
    fps_idx = p2.farthest_point_sample(xyz, 512, start=torch.zeros(B, dtype=torch.long))
    assert all(len(set(r.tolist())) == 512 for r in fps_idx.cpu()[:4])


def test_set_abstraction_stack_config3_vs_oracle(dev):
    """BASELINE.json configs[3] AT ITS STATED SIZE (B = 32, N = 2048): the three-layer stack (FPS + ball query + grouping + SA-MLP, then a
    group-all layer) on the HIP path against oracle/ref_sa_cpu.py (pinned to pointnet_util.py by tests/golden/sa_*.npz) on the same
    parameters and cloud: sampled centres identical (FPS / ball-query indices are bit-exact), outputs within 1e-3 of the fp32 oracle.
    Gradients: the loss reaches the weights through three neighbourhood / global max-poolings, i.e. through 32 x 1024 + ... arg-max
    ROUTES.  Two fp32 evaluations whose forward values differ in the sixth digit (measured, tools/sa_diag.py: HIP 3e-6 .. 9e-6 relative
    L2 from the float64 oracle per layer, the fp32 oracle 6e-7 .. 2e-6 -- the folded first conv u_j - w_i rounds 4x coarser than
    W (x_j - c_i), both 100x inside the 1e-3 contract) route a handful of near-tied maxima differently, and every re-routed maximum
    moves ~0.5 % of a gradient's norm: the fp32 oracle happens to re-route none at this size (its last conv's gradient sits 5e-6 from
    float64), the HIP path about ten (2e-2).  The reference fixtures at N <= 256 (no near ties) hold every gradient to 5e-3
    (test_set_abstraction_layer_vs_reference_golden); here the bar is 6e-2 relative L2 against the float64 oracle: a wrong formula
    would miss by O(1), and the BatchNorm gradients that do not pass through a re-routed maximum (last layer) must agree to 1e-4."""
    from mlsp_amd import pointnet2 as p2
    torch.manual_seed(7)
    B, N = 32, 2048
    xyz = torch.rand(B, N, 3) * 2 - 1
    cfgs = [dict(npoint=512, radius=0.2, nsample=32, D=0, mlp=[64, 64, 128], group_all=False),
            dict(npoint=128, radius=0.4, nsample=64, D=128, mlp=[128, 128, 256], group_all=False),
            dict(npoint=None, radius=None, nsample=None, D=256, mlp=[256, 512, 1024], group_all=True)]
    layers = [p2.PointNetSetAbstraction(c["npoint"], c["radius"], c["nsample"], 3 + c["D"], c["mlp"], c["group_all"]) for c in cfgs]
    starts = [torch.randint(0, N, (B,)), torch.randint(0, 512, (B,)), None]
    with torch.no_grad():
        for l in layers:                                       # off-default BatchNorm parameters (both signs of the scale)
            for bn in l.mlp_bns:
                bn.weight.mul_(1.0 + 0.3 * torch.randn_like(bn.weight))
                bn.weight[torch.rand_like(bn.weight) < 0.2] *= -1
                bn.bias.add_(0.2 * torch.randn_like(bn.bias))
    wgt = torch.randn(B, 1, 1024)

    def oracle(dtype, fps_idx=None):
        grads, x, f, bufs = [], xyz.to(dtype), None, []
        prs = []
        for l, c, st in zip(layers, cfgs, starts):
            pr = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in l.named_parameters()}
            bf = {k: v.detach().clone().to(dtype) for k, v in l.named_buffers() if not k.endswith("num_batches_tracked")}
            x, f, _, _, nb = sa.sa_forward(pr, bf, c, x, f, st)
            prs.append(pr)
            bufs.append(nb)
        (f * wgt.to(dtype)).sum().backward()
        return f.detach(), [{k: v.grad.double().numpy() for k, v in pr.items()} for pr in prs], bufs

    want32, g32, obufs = oracle(torch.float32)
    _, g64, _ = oracle(torch.float64)
    # HIP path
    gx, gf = xyz.to(dev), None
    for l, st in zip(layers, starts):
        l.to(dev).train()
        if st is not None:
            l.fps_start = st
        gx, gf = l(gx, gf)
    np.testing.assert_allclose(gf.detach().cpu().numpy(), want32.numpy(), rtol=1e-3, atol=1e-3)
    (gf * wgt.to(dev)).sum().backward()
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
    report, bad = {}, []
    for li, (l, nb) in enumerate(zip(layers, obufs)):
        for k, p in l.named_parameters():
            if "mlp_convs" in k and k.endswith("bias"):
                continue                                       # analytically zero in front of a batch-statistics BatchNorm
            hip, yard = rel(p.grad.double().cpu().numpy(), g64[li][k]), rel(g32[li][k], g64[li][k])
            report["sa%d.%s" % (li + 1, k)] = "%.1e | %.1e" % (hip, yard)
            if not hip < (1e-4 if (li == 2 and k.startswith("mlp_bns.2")) else max(6e-2, 3.0 * yard)):
                bad.append((li, k, hip, yard))
        for k, v in l.named_buffers():
            if not k.endswith("num_batches_tracked"):
                np.testing.assert_allclose(v.cpu().numpy(), nb[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    print("configs[3] stack, gradient rel-L2 vs float64 (HIP fp32 | oracle fp32):", report)
    assert not bad, bad


def test_set_abstraction_stack_gradients_forced_selections(dev):
    """The arithmetic of the stack's backward, with NOTHING discrete left to differ (round-4 verdict item 4c): FPS / ball-query indices are
    bit-exact, and the arg-max slot of every neighbourhood max is forced to the float64 oracle's on the HIP path and on the fp32 oracle
    (the yardstick) alike -- so no near-tied maximum is re-routed and every parameter gradient must sit within max(5e-3, 3 x yardstick)
    relative L2 of float64 (the free-running test above needs 6e-2 at B = 32, N = 2048).  What is left are the ReLU kinks: see the second
    yardstick below.  The kernels themselves, one layer at a time under the same forcing, sit at 3e-7 .. 2e-6 -- below the fp32 oracle's
    5e-7 .. 5e-6 (tools/r5/sa_layer_debug.py)."""
    from mlsp_amd import functional as Fh, pointnet2 as p2
    torch.manual_seed(11)
    B, N = 8, 1024
    xyz = torch.rand(B, N, 3) * 2 - 1
    cfgs = [dict(npoint=256, radius=0.25, nsample=32, D=0, mlp=[64, 64, 128], group_all=False),
            dict(npoint=64, radius=0.5, nsample=64, D=128, mlp=[128, 128, 256], group_all=False),
            dict(npoint=None, radius=None, nsample=None, D=256, mlp=[256, 512, 1024], group_all=True)]
    layers = [p2.PointNetSetAbstraction(c["npoint"], c["radius"], c["nsample"], 3 + c["D"], c["mlp"], c["group_all"]) for c in cfgs]
    starts = [torch.randint(0, N, (B,)), torch.randint(0, 256, (B,)), None]
    with torch.no_grad():
        for l in layers:
            for bn in l.mlp_bns:
                bn.weight.mul_(1.0 + 0.3 * torch.randn_like(bn.weight))
                bn.weight[torch.rand_like(bn.weight) < 0.2] *= -1
                bn.bias.add_(0.2 * torch.randn_like(bn.bias))
    wgt = torch.randn(B, 1, 1024)

    def oracle(dtype, sels=None, forced_groups=None, noise=0.0, noise_seed=0):
        gen = torch.Generator().manual_seed(noise_seed)
        x, f, prs, rec, groups = xyz.detach().clone().to(dtype), None, [], [], []
        for li, (l, c, st) in enumerate(zip(layers, cfgs, starts)):
            pr = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in l.named_parameters()}
            bf = {k: v.detach().clone().to(dtype) for k, v in l.named_buffers() if not k.endswith("num_batches_tracked")}
            fg = None if forced_groups is None or forced_groups[li][1] is None else forced_groups[li]
            x, f, fidx, gidx, _ = sa.sa_forward(pr, bf, c, x, f, st, sel=None if sels is None else sels[li], sel_out=rec if sels is None else None,
                                                groups=fg)
            if noise:                                          # a forward error of relative size `noise` on every layer's output
                f = f * (1.0 + noise * torch.randn(f.shape, generator=gen, dtype=dtype))
            prs.append(pr)
            groups.append((fidx, gidx))
        (f * wgt.to(dtype)).sum().backward()
        return f.detach(), [{k: v.grad.double().numpy() for k, v in pr.items()} for pr in prs], rec, groups

    # the discrete choices of the fp32 run (sampled centres, ball-query groups: bit-exact on the HIP path, tests above) are imposed on the
    # float64 run: a handful of radius-boundary neighbours fall the other way in float64, which is another network
    _, _, _, groups32 = oracle(torch.float32)
    out64, g64, sel64, groups = oracle(torch.float64, forced_groups=groups32)
    groups = [g[1] for g in groups]
    # A ball-query group is padded with copies of its first hit (pointnet_util.py:84-86): identical rows, an exact tie that torch.max may
    # give to any of them.  The HIP path's convention is the FIRST slot (its reverse index leaves the padding slots out: sa.hip), so the
    # selections are expressed that way -- the same function: a padding slot and slot 0 carry the same value and the same source point.
    for li, gidx in enumerate(groups):
        if gidx is not None:
            pad = gidx == gidx[..., :1]                                          # [B, S, ns]: slot holds the group's first hit
            sel64[li] = torch.where(torch.gather(pad, 2, sel64[li]), torch.zeros_like(sel64[li]), sel64[li])
    _, g32, _, _ = oracle(torch.float32, sel64, forced_groups=groups32)
    # Second yardstick: the float64 oracle itself with a relative forward error of 1e-5 on every layer's output -- the HIP path's measured
    # forward accuracy (3e-6 .. 9e-6 relative L2 per layer against float64, tools/r5/sa_forced_debug.py).  The loss reaches the weights
    # through ONE row per (cloud, channel) of the group-all layer, so a single ReLU whose pre-activation lies within that error of zero
    # (one unit at 1.9e-8 for this seed) moves every upstream gradient by ~1e-2 when it flips: such units set the floor for ANY
    # implementation whose forward is not bit-identical to float64 (the fp32 oracle happens to keep this one's sign).
    gn = [oracle(torch.float64, sel64, forced_groups=groups32, noise=1e-5, noise_seed=s)[1] for s in (1, 2)]
    gx, gf = xyz.to(dev), None
    with Fh.forced_selections([s.reshape(-1, s.shape[-1]) for s in sel64]):
        for l, st in zip(layers, starts):
            l.to(dev).train()
            if st is not None:
                l.fps_start = st
            gx, gf = l(gx, gf)
    np.testing.assert_allclose(gf.detach().cpu().numpy(), out64.numpy(), rtol=1e-3, atol=1e-3)
    (gf * wgt.to(dev)).sum().backward()
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
    report, bad = {}, []
    for li, l in enumerate(layers):
        for k, p in l.named_parameters():
            if "mlp_convs" in k and k.endswith("bias"):
                continue                                       # analytically zero in front of a batch-statistics BatchNorm
            hip, yard = rel(p.grad.double().cpu().numpy(), g64[li][k]), rel(g32[li][k], g64[li][k])
            yard_eps = max(rel(g[li][k], g64[li][k]) for g in gn)
            report["sa%d.%s" % (li + 1, k)] = "%.1e | %.1e | %.1e" % (hip, yard, yard_eps)
            if not hip < max(5e-3, 3.0 * yard, 3.0 * yard_eps):
                bad.append((li, k, hip, yard, yard_eps))
    print("SA stack, forced selections, gradient rel-L2 vs float64 (HIP fp32 | oracle fp32 | float64 with 1e-5 forward error):", report)
    assert not bad, bad


# ----------------------------------------------------------------------------- round 4: gradients with respect to the coordinates
@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("B,N,S,radius,ns,D,mlp", [(3, 256, 48, 0.45, 16, 5, [32, 64]), (2, 200, 40, 0.5, 12, 0, [16, 32, 64])])
def test_set_abstraction_coordinate_gradients_vs_oracle(dev, fold, B, N, S, radius, ns, D, mlp):
    """d loss / d xyz of a set-abstraction layer (pointnet_util.py:120-124: `grouped_xyz - new_xyz` is differentiable in both operands, the
    sampled centres are rows of xyz; FPS / ball-query indices carry no gradient) on both paths -- the grouped tensor (_Group) and the
    folded first layer (the two K = 3 products) -- against the autograd of oracle/ref_sa_cpu.sa_forward in float64 on the same indices."""
    from mlsp_amd import pointnet2 as p2
    torch.manual_seed(3)
    xyz = torch.rand(B, N, 3) * 2 - 1
    feat = torch.randn(B, N, D) if D else None
    layer = p2.PointNetSetAbstraction(S, radius, ns, 3 + D, mlp, False)
    with torch.no_grad():
        for bn in layer.mlp_bns:
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    start = torch.arange(B) % N
    cfg = dict(npoint=S, radius=radius, nsample=ns, D=D, mlp=mlp, group_all=False)
    wgt = torch.randn(B, S, mlp[-1])

    def oracle(dtype):
        pr = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in layer.named_parameters()}
        bf = {k: v.detach().clone().to(dtype) for k, v in layer.named_buffers() if not k.endswith("num_batches_tracked")}
        x = xyz.detach().clone().to(dtype).requires_grad_(True)
        f = feat.detach().clone().to(dtype).requires_grad_(True) if D else None
        _, out, fps_idx, gidx, _ = sa.sa_forward(pr, bf, cfg, x, f, start)
        (out * wgt.to(dtype)).sum().backward()
        return out.detach(), x.grad, (f.grad if D else None), fps_idx, gidx

    o64, dx64, df64, fps64, gidx64 = oracle(torch.float64)
    o32, dx32, _, fps32, gidx32 = oracle(torch.float32)
    layer = layer.to(dev).train()
    layer.fold_first, layer.fps_start = fold, start
    if not fold:
        layer.fuse_max = False
    gx = xyz.detach().clone().to(dev).requires_grad_(True)
    gf = feat.detach().clone().to(dev).requires_grad_(True) if D else None
    new_xyz, out = layer(gx, gf)
    assert torch.equal(fps32, fps64) and torch.equal(gidx32, gidx64)            # the float64 oracle groups the same points
    np.testing.assert_allclose(out.detach().cpu().numpy(), o32.numpy(), rtol=1e-3, atol=1e-3)
    (out * wgt.to(dev)).sum().backward()
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
    hip, yard = rel(gx.grad.double().cpu().numpy(), dx64.numpy()), rel(dx32.double().numpy(), dx64.numpy())
    print("d xyz rel-L2 vs float64: HIP %.2e | fp32 oracle %.2e (fold_first=%s)" % (hip, yard, fold))
    assert hip < max(2e-3, 5.0 * yard), (hip, yard)
    assert gx.grad.abs().max().item() > 0
    if D:
        assert rel(gf.grad.double().cpu().numpy(), df64.numpy()) < 2e-3


def test_group_coordinate_gradients_closed_form(dev):
    """_Group alone: d/d xyz and d/d new_xyz of sum(W * [xyz_j - c_i | f_j]) are a scatter of W's coordinate columns onto the grouped points
    and minus their per-group sums (pointnet_util.py:120-124)."""
    from mlsp_amd import pointnet2 as p2
    torch.manual_seed(5)
    B, N, S, ns, D = 2, 120, 30, 9, 4
    xyz = (torch.rand(B, N, 3) * 2 - 1).to(dev).requires_grad_(True)
    ctr = (torch.rand(B, S, 3) * 2 - 1).to(dev).requires_grad_(True)
    feat = torch.randn(B, N, D).to(dev).requires_grad_(True)
    idx = torch.randint(0, N, (B, S, ns)).to(dev)
    W = torch.randn(B * S * ns, 3 + D).to(dev)
    (p2._Group.apply(xyz, ctr, feat, idx) * W).sum().backward()
    x2, c2, f2 = (t.detach().clone().requires_grad_(True) for t in (xyz, ctr, feat))
    G = torch.cat([p2.index_points(x2, idx) - c2[:, :, None], p2.index_points(f2, idx)], -1).reshape(B * S * ns, 3 + D)
    (G * W).sum().backward()
    for got, want in ((xyz.grad, x2.grad), (ctr.grad, c2.grad), (feat.grad, f2.grad)):
        assert (got - want).abs().max().item() < 1e-4 * (want.abs().max().item() + 1.0)


# ----------------------------------------------------------------------------- round 2: knn grouping, multi-scale grouping, feature propagation
from test_sa_oracle_cpu import split_state


def _load_into(layer, g, dev):
    params, buffers = split_state(g)
    sd = dict(params)
    sd.update(buffers)
    for k in layer.state_dict():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
    layer.load_state_dict(sd, strict=True)                 # the reference's state_dict keys
    return layer.to(dev).train()


def _check_layer(g, layer, tol_grad=5e-3):
    for k, p in layer.named_parameters():
        want = g["grad/" + k]
        if "conv" in k and k.endswith(".bias"):    # a bias in front of a batch-statistics BatchNorm: analytically zero (the reference holds rounding noise)
            assert np.abs(want).max() < 1e-3 and p.grad.abs().max().item() < 1e-3, k
            continue
        err = np.linalg.norm(p.grad.cpu().numpy() - want) / (np.linalg.norm(want) + 1e-12)
        assert err < tol_grad, (k, err)
    st = layer.state_dict()
    for k in st:
        if "running" in k:
            np.testing.assert_allclose(st[k].cpu().numpy(), g["state_after/" + k], rtol=1e-3, atol=1e-5, err_msg=k)


def test_knn_point_bit_exact_vs_oracle(dev):
    from mlsp_amd import pointnet2 as p2
    g = torch.Generator().manual_seed(21)
    for B, N, S, k in [(2, 300, 48, 12), (1, 2048, 512, 32), (3, 64, 64, 40), (2, 500, 7, 3)]:
        xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
        q = xyz[:, torch.randperm(N, generator=g)[:S]] + (0.0 if k == 12 else 0.05) * torch.randn(B, S, 3, generator=g)
        want = sa.knn_group(k, xyz, q)
        got, dist = p2.knn_point(k, xyz.to(dev), q.to(dev), return_dist=True)
        d = sa.square_distance(q, xyz)
        gap = torch.sort(d, dim=-1)[0]
        safe = ((gap[:, :, 1:k + 1] - gap[:, :, :k]).min(-1)[0] > 5e-6) if N > k else torch.ones(B, S, dtype=torch.bool)
        assert torch.equal(got.cpu()[safe], want[safe]) and safe.float().mean() > 0.8     # rows whose rank gaps exceed fp32 rounding
        np.testing.assert_allclose(dist.cpu().numpy(), torch.gather(d, 2, got.cpu()).numpy(), rtol=0, atol=2e-6)


def test_sa_knn_grouping_vs_reference_golden(dev):
    from mlsp_amd import pointnet2 as p2
    g = dict(np.load(os.path.join(GOLD, "sa_knn_s11_B3_N300.npz")))
    layer = _load_into(p2.PointNetSetAbstraction(48, 0.3, 12, 3 + 5, [16, 32], False, knn=True), g, dev)
    layer.fps_start = torch.from_numpy(g["start"])
    xyz = torch.from_numpy(g["xyz"]).to(dev)
    pts = torch.from_numpy(g["points"]).to(dev).requires_grad_(True)
    assert np.array_equal(p2.knn_point(12, xyz, p2.index_points(xyz, torch.from_numpy(g["fps_idx"]).to(dev))).cpu().numpy(), g["group_idx"])
    new_xyz, out = layer(xyz, pts)
    np.testing.assert_allclose(new_xyz.cpu().numpy(), g["new_xyz"], rtol=0, atol=0)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["new_points"], rtol=1e-3, atol=2e-4)
    (out * torch.from_numpy(g["wgt"]).to(dev)).sum().backward()
    np.testing.assert_allclose(pts.grad.cpu().numpy(), g["d_points"], rtol=1e-3, atol=1e-4)
    _check_layer(g, layer)


@pytest.mark.parametrize("name,knn,D", [("sa_msg_s12_B2_N256.npz", False, 6), ("sa_msgknn_s12_B2_N256.npz", True, 0)])
def test_sa_msg_vs_reference_golden(dev, name, knn, D):
    from mlsp_amd import pointnet2 as p2
    g = dict(np.load(os.path.join(GOLD, name)))
    layer = _load_into(p2.PointNetSetAbstractionMsg(32, [0.2, 0.45], [8, 16], D, [[16, 24], [16, 32]], knn=knn), g, dev)
    xyz = torch.from_numpy(g["xyz"]).to(dev)
    pts = torch.from_numpy(g["points"]).to(dev).requires_grad_(True) if D else None
    new_xyz, out = layer(xyz, pts, seed_idx=torch.from_numpy(g["fps_idx"]).to(dev))
    np.testing.assert_allclose(new_xyz.cpu().numpy(), g["new_xyz"], rtol=0, atol=0)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["new_points"], rtol=1e-3, atol=2e-4)
    (out * torch.from_numpy(g["wgt"]).to(dev)).sum().backward()
    if D:
        np.testing.assert_allclose(pts.grad.cpu().numpy(), g["d_points"], rtol=1e-3, atol=1e-4)
    _check_layer(g, layer)


@pytest.mark.parametrize("name", ["sa_fp_s13_B2_N200.npz", "sa_fp1_s13_B2_N200.npz"])
def test_feature_propagation_vs_reference_golden(dev, name):
    from mlsp_amd import pointnet2 as p2
    g = dict(np.load(os.path.join(GOLD, name)))
    layer = _load_into(p2.PointNetFeaturePropagation(4 + 7, [16, 12]), g, dev)
    p1 = torch.from_numpy(g["points1"]).to(dev).requires_grad_(True)
    pp2 = torch.from_numpy(g["points2"]).to(dev).requires_grad_(True)
    out = layer(torch.from_numpy(g["xyz1"]).to(dev), torch.from_numpy(g["xyz2"]).to(dev), p1, pp2)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], rtol=1e-3, atol=1e-3)
    (out * torch.from_numpy(g["wgt"]).to(dev)).sum().backward()
    np.testing.assert_allclose(p1.grad.cpu().numpy(), g["d_points1"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(pp2.grad.cpu().numpy(), g["d_points2"], rtol=2e-3, atol=1e-3)
    _check_layer(g, layer, tol_grad=1e-2)


# ----------------------------------------------------------------------------- folded first layer (no grouped tensor)
@pytest.mark.parametrize("B,N,S,radius,ns,D,mlp,knn,train", [(4, 256, 64, 0.4, 16, 0, [64, 64, 128], False, True),
                                                              (3, 300, 50, 0.5, 24, 13, [128, 64], False, True),
                                                              (2, 200, 40, 0.5, 8, 5, [32, 48], True, True),
                                                              (2, 256, 64, 0.4, 16, 6, [64, 32], False, False)])
def test_folded_first_sa_layer_equals_grouped_path(dev, B, N, S, radius, ns, D, mlp, knn, train):
    """W [x_j - c_i ; f_j] + b = u_j - w_i on the points (mlsp_sa_fold_*_f32, `fold_first`) against the materialised grouped tensor +
    edge GEMM of the same module: outputs, feature / parameter gradients and BatchNorm buffers (training and eval mode)."""
    from mlsp_amd import pointnet2 as p2
    import copy
    torch.manual_seed(7)
    xyz = (torch.rand(B, N, 3) * 2 - 1).to(dev)
    feat = torch.randn(B, N, D).to(dev) if D else None
    a = p2.PointNetSetAbstraction(S, radius, ns, 3 + D, mlp, False, knn=knn).to(dev)
    for bn in a.mlp_bns:                                   # non-trivial affine parameters and buffers (negative scales included)
        bn.weight.data.uniform_(-1.0, 1.5); bn.bias.data.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
    b = copy.deepcopy(a)
    a.fold_first, b.fold_first = True, False
    b.fuse_max = False                                     # path b: grouped tensor, edge GEMMs, activated tensor, separate max
    start = torch.arange(B) % N
    a.fps_start = b.fps_start = start
    a.train(train); b.train(train)
    fa = feat.clone().requires_grad_(True) if D else None
    fb = feat.clone().requires_grad_(True) if D else None
    xa, oa = a(xyz, fa)
    xb, ob = b(xyz, fb)
    assert torch.equal(xa, xb)
    scale = ob.abs().max().item() + 1e-9
    assert (oa - ob).abs().max().item() / scale < 2e-5, (oa - ob).abs().max().item() / scale
    wgt = torch.randn_like(oa)
    (oa * wgt).sum().backward()
    (ob * wgt).sum().backward()
    if D:
        err = (fa.grad - fb.grad).abs().max().item() / (fb.grad.abs().max().item() + 1e-9)
        assert err < 2e-4, err
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if k.startswith("mlp_convs") and k.endswith("bias") and train:
            continue                                       # analytically zero in front of a batch-statistics BatchNorm
        err = (pa.grad - pb.grad).abs().max().item() / (pb.grad.abs().max().item() + 1e-9)
        assert err < 5e-4, (k, err)
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        np.testing.assert_allclose(va.cpu().numpy(), vb.cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
