"""The set-abstraction oracle (oracle/ref_sa_cpu.py) against the reference's own outputs (tests/golden/sa_*.npz, captured by
tools/make_golden.py sa from PointDA/hengshuang_transformer/pointnet_util.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_sa_cpu as sa

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    g = dict(np.load(os.path.join(GOLD, name)))
    cfgv = g["cfg"]
    cfg = {"npoint": int(cfgv[0]), "nsample": int(cfgv[1]), "group_all": bool(cfgv[2]), "D": int(cfgv[3]),
           "mlp": [int(v) for v in cfgv[4:]], "radius": float(g["radius"])}
    params = {k[len("state_before/"):]: torch.from_numpy(v) for k, v in g.items()
              if k.startswith("state_before/") and "running" not in k and "num_batches" not in k}
    buffers = {k[len("state_before/"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("state_before/") and "running" in k}
    return g, cfg, params, buffers


@pytest.mark.parametrize("name", ["sa_s0_B4_N256.npz", "sa_s1_B3_N200_nofeat.npz", "sa_s2_B4_N128_all.npz"])
def test_sa_oracle_matches_reference(name):
    g, cfg, params, buffers = load_case(name)
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xyz = torch.from_numpy(g["xyz"])
    points = torch.from_numpy(g["points"]).requires_grad_(True) if cfg["D"] else None
    start = torch.from_numpy(g["start"]) if "start" in g else None
    new_xyz, out, fps_idx, gidx, nb = sa.sa_forward(params, buffers, cfg, xyz, points, start)
    if not cfg["group_all"]:
        assert np.array_equal(fps_idx.numpy(), g["fps_idx"])
        assert np.array_equal(gidx.numpy(), g["group_idx"])
    np.testing.assert_allclose(new_xyz.numpy(), g["new_xyz"], rtol=0, atol=0)
    np.testing.assert_allclose(out.detach().numpy(), g["new_points"], rtol=1e-5, atol=1e-5)
    (out * torch.from_numpy(g["wgt"])).sum().backward()
    if cfg["D"]:
        np.testing.assert_allclose(points.grad.numpy(), g["d_points"], rtol=1e-4, atol=1e-5)
    for k, v in params.items():
        np.testing.assert_allclose(v.grad.numpy(), g["grad/" + k], rtol=1e-3, atol=2e-4, err_msg=k)
    for k, v in nb.items():
        np.testing.assert_allclose(v.numpy(), g["state_after/" + k], rtol=1e-5, atol=1e-6, err_msg=k)
    # the forced-selection hook of the gradient pins (tests/test_gpu_sa.py): forcing the pooling to its own recorded slots is the identity,
    # for the output and for every gradient -- so the pinned oracle and the forced one are the same function
    rec = []
    p2 = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    _, out_a, _, _, _ = sa.sa_forward(p2, buffers, cfg, xyz, points.detach() if cfg["D"] else None, start, sel_out=rec)
    p3 = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    _, out_b, _, _, _ = sa.sa_forward(p3, buffers, cfg, xyz, points.detach() if cfg["D"] else None, start, sel=rec[0])
    assert torch.equal(out_a, out_b) and torch.equal(out_a, out.detach())
    (out_b * torch.from_numpy(g["wgt"])).sum().backward()
    for k, v in params.items():
        np.testing.assert_allclose(p3[k].grad.numpy(), v.grad.numpy(), rtol=1e-6, atol=1e-7, err_msg=k)


def test_sa_oracle_indices_at_config3_scale():
    g = dict(np.load(os.path.join(GOLD, "sa_idx_s3_B2_N2048.npz")))
    xyz = torch.from_numpy(g["xyz"])
    fps_idx = sa.fps(xyz, g["fps_idx"].shape[1], torch.from_numpy(g["start"]))
    assert np.array_equal(fps_idx.numpy(), g["fps_idx"].astype(np.int64))
    gidx = sa.ball_query(float(g["radius"]), int(g["nsample"]), xyz, sa.gather_rows(xyz, fps_idx))
    assert np.array_equal(gidx.numpy(), g["group_idx"].astype(np.int64))


# ----------------------------------------------------------------------------- round 2: knn grouping, multi-scale grouping, feature propagation
def split_state(g):
    """(params requiring grad, BatchNorm buffers at their initial values) from a fixture's state_after/* entries: backward does not
    change parameters, and the running statistics start from the nn.BatchNorm defaults (the generator only perturbs parameters)."""
    params = {k[len("state_after/"):]: torch.from_numpy(v).clone() for k, v in g.items()
              if k.startswith("state_after/") and "running" not in k and "num_batches" not in k}
    buffers = {}
    for k, v in g.items():
        if k.startswith("state_after/") and "running_mean" in k:
            buffers[k[len("state_after/"):]] = torch.zeros(v.shape)
        elif k.startswith("state_after/") and "running_var" in k:
            buffers[k[len("state_after/"):]] = torch.ones(v.shape)
    return params, buffers


def check_grads_and_buffers(g, params, nb, rtol=1e-3, atol=2e-4):
    for k, v in params.items():
        np.testing.assert_allclose(v.grad.numpy(), g["grad/" + k], rtol=rtol, atol=atol, err_msg=k)
    for k, v in nb.items():
        np.testing.assert_allclose(v.numpy(), g["state_after/" + k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_sa_knn_grouping_oracle_matches_reference():
    g = dict(np.load(os.path.join(GOLD, "sa_knn_s11_B3_N300.npz")))
    xyz = torch.from_numpy(g["xyz"])
    fps_idx = sa.fps(xyz, 48, torch.from_numpy(g["start"]))
    assert np.array_equal(fps_idx.numpy(), g["fps_idx"])
    gidx = sa.knn_group(12, xyz, sa.gather_rows(xyz, fps_idx))
    assert np.array_equal(gidx.numpy(), g["group_idx"])


@pytest.mark.parametrize("name,knn,D", [("sa_msg_s12_B2_N256.npz", False, 6), ("sa_msgknn_s12_B2_N256.npz", True, 0)])
def test_sa_msg_oracle_matches_reference(name, knn, D):
    g = dict(np.load(os.path.join(GOLD, name)))
    params, buffers = split_state(g)
    params = {k: v.requires_grad_(True) for k, v in params.items()}
    cfg = {"radius_list": [0.2, 0.45], "nsample_list": [8, 16], "mlp_list": [[16, 24], [16, 32]], "knn": knn}
    points = torch.from_numpy(g["points"]).requires_grad_(True) if D else None
    new_xyz, out, _, nb = sa.sa_msg_forward(params, buffers, cfg, torch.from_numpy(g["xyz"]), points, torch.from_numpy(g["fps_idx"]))
    np.testing.assert_allclose(new_xyz.numpy(), g["new_xyz"], rtol=0, atol=0)
    np.testing.assert_allclose(out.detach().numpy(), g["new_points"], rtol=1e-5, atol=1e-5)
    (out * torch.from_numpy(g["wgt"])).sum().backward()
    if D:
        np.testing.assert_allclose(points.grad.numpy(), g["d_points"], rtol=1e-4, atol=1e-5)
    check_grads_and_buffers(g, params, nb)


@pytest.mark.parametrize("name", ["sa_fp_s13_B2_N200.npz", "sa_fp1_s13_B2_N200.npz"])
def test_feature_propagation_oracle_matches_reference(name):
    g = dict(np.load(os.path.join(GOLD, name)))
    params, buffers = split_state(g)
    params = {k: v.requires_grad_(True) for k, v in params.items()}
    p1 = torch.from_numpy(g["points1"]).requires_grad_(True)
    p2 = torch.from_numpy(g["points2"]).requires_grad_(True)
    out, nb = sa.fp_forward(params, buffers, [16, 12], torch.from_numpy(g["xyz1"]), torch.from_numpy(g["xyz2"]), p1, p2)
    # the interpolation weights 1 / (d + 1e-8) amplify the rounding of d = -2ab + a^2 + b^2 for near-coincident points (d ~ 1e-4)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-4, atol=1e-4)
    (out * torch.from_numpy(g["wgt"])).sum().backward()
    np.testing.assert_allclose(p1.grad.numpy(), g["d_points1"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(p2.grad.numpy(), g["d_points2"], rtol=1e-3, atol=1e-4)
    check_grads_and_buffers(g, params, nb, rtol=2e-3, atol=5e-4)
