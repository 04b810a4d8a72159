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


def test_sa_oracle_indices_at_config3_scale():
    g = dict(np.load(os.path.join(GOLD, "sa_idx_s3_B2_N2048.npz")))
    xyz = torch.from_numpy(g["xyz"])
    fps_idx = sa.fps(xyz, g["fps_idx"].shape[1], torch.from_numpy(g["start"]))
    assert np.array_equal(fps_idx.numpy(), g["fps_idx"].astype(np.int64))
    gidx = sa.ball_query(float(g["radius"]), int(g["nsample"]), xyz, sa.gather_rows(xyz, fps_idx))
    assert np.array_equal(gidx.numpy(), g["group_idx"].astype(np.int64))
