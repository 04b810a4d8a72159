"""On-device input corruption and PCM mixing against the reference's golden vectors and the numpy oracle."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import ref_corrupt_np as oc
from test_corrupt_oracle_cpu import GOLD, check_deform_against_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_region_assignment_bit_exact_vs_reference(dev):
    from mlsp_amd import pc_utils
    g = dict(np.load(os.path.join(GOLD, "deform_s5_B6_N1024.npz")))
    got = pc_utils.assign_region_to_point(torch.from_numpy(g["X"]).to(dev), dev)
    assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), g["regions"])
    assert np.allclose(pc_utils.region_mean(3), g["lookup"])


@pytest.mark.parametrize("groups", [1, 3])
def test_deform_input_vs_reference_and_oracle(dev, groups):
    from mlsp_amd import mlsp
    g = dict(np.load(os.path.join(GOLD, "deform_s5_B6_N1024.npz")))
    X = torch.from_numpy(g["X"]).to(dev)
    noise = torch.randn(X.shape, generator=torch.Generator().manual_seed(groups))
    Xd, mask = mlsp.deform_input(X, torch.from_numpy(g["lookup"]), 'volume_based_voxels', dev, groups,
                                 region_ids=g["perm_g%d" % groups], noise=noise)
    assert Xd.data_ptr() == X.data_ptr()                       # in place, like the reference
    check_deform_against_reference(g["X"], Xd.cpu().numpy(), mask.cpu().numpy(), g, groups)
    want, wmask = oc.deform(g["X"], g["lookup"], g["perm_g%d" % groups], noise.numpy(), groups)
    assert np.array_equal(mask.cpu().numpy(), wmask)
    np.testing.assert_allclose(Xd.cpu().numpy(), want, rtol=0, atol=1e-7)
    # default random draws: still a valid deformation
    X2 = torch.from_numpy(g["X"]).to(dev)
    _, m2 = mlsp.deform_input(X2, torch.from_numpy(g["lookup"]), 'volume_based_voxels', dev)
    assert (m2[:, 0].sum(1) >= 40).all() and (m2[:, :3].amax(1) == m2[:, :3].amin(1)).all()


def test_pcm_mix_shapes_vs_reference(dev):
    from mlsp_amd import PCM
    g = dict(np.load(os.path.join(GOLD, "pcm_s3_B5_N256.npz")))
    args = types.SimpleNamespace(cuda=True, mixup_params=1.0, DefRec_weight=0.5)
    rng = {"index": torch.from_numpy(g["index"]), "lam": float(g["lam"]), "start_a": torch.from_numpy(g["start_a"]),
           "start_b": torch.from_numpy(g["start_b"]), "points_perm": torch.from_numpy(g["points_perm"])}
    mixed, (Ya, Yb, lam) = PCM.mix_shapes(args, torch.from_numpy(g["X"]).to(dev), torch.from_numpy(g["Y"]).to(dev), rng=rng)
    assert np.array_equal(mixed.cpu().numpy(), g["mixed"])
    assert np.array_equal(Ya.cpu().numpy(), g["Ya"]) and np.array_equal(Yb.cpu().numpy(), g["Yb"]) and lam == float(g["lam"])
    # unseeded draws: a permutation of sampled columns of the two parents
    mixed2, (_, _, lam2) = PCM.mix_shapes(args, torch.from_numpy(g["X"]).to(dev), torch.from_numpy(g["Y"]).to(dev))
    assert mixed2.shape == mixed.shape and 0.0 <= lam2 <= 1.0


def test_pc_utils_fps_vs_oracle(dev):
    from mlsp_amd import pc_utils
    gen = torch.Generator().manual_seed(2)
    xyz = torch.rand(3, 3, 500, generator=gen) * 2 - 1
    start = torch.tensor([0, 499, 123])
    idx, vals = pc_utils.farthest_point_sample(None, xyz.to(dev), 77, start=start)
    widx, wvals = oc.fps(xyz.numpy(), 77, start.numpy())
    assert np.array_equal(idx.cpu().numpy(), widx) and np.array_equal(vals.cpu().numpy(), wvals)


def test_scan_input_vs_reference(dev):
    """mlsp.scan_input with the reference's own random draws (pixel size, rotation angles): identical survivors and mask."""
    from mlsp_amd import mlsp
    g = dict(np.load(os.path.join(GOLD, "scan_s21_B5_N512.npz")))
    X = torch.from_numpy(g["X"]).to(dev)
    Xs, mask = mlsp.scan_input(X, dev, pixel_size=float(g["pixel_size"]), angles=g["angles"])
    assert Xs.data_ptr() == X.data_ptr()
    assert np.array_equal(mask.cpu().numpy(), g["mask"]) and np.array_equal(Xs.cpu().numpy(), g["X_out"])
    # unseeded: one survivor per occupied cell, everything else zeroed
    X2 = torch.from_numpy(g["X"]).to(dev)
    Xs2, m2 = mlsp.scan_input(X2, dev)
    kept = (m2[:, :, 0] == 0)
    assert (kept.sum(1) > 100).all() and (Xs2[~kept] == 0).all() and torch.equal(Xs2[kept], torch.from_numpy(g["X"]).to(dev)[kept])


def test_deform_input_volume_based_radius(dev):
    """mlsp.deform_input(..., 'volume_based_radius') (MLSP/mlsp.py:33-36 -> pc_utils.collapse_to_point) on device: with the reference's
    recorded picks and Gaussian draws the mask is identical and the deformed cloud equal to fp32 rounding; with its own draws the
    picked point is a candidate (>= 20 points within 0.5) and exactly the points within 0.5 of it move."""
    from mlsp_amd import mlsp
    g = dict(np.load(os.path.join(GOLD, "collapse_s9_B5_N512.npz")))
    X = torch.from_numpy(g["X"]).to(dev)
    Xd, mask = mlsp.deform_input(X.clone(), None, 'volume_based_radius', dev, choice=g["choice"], noise=torch.from_numpy(g["noise"]))
    assert np.array_equal(mask.cpu().numpy(), g["mask"])
    np.testing.assert_allclose(Xd.cpu().numpy(), g["X_out"], rtol=0, atol=3e-7)
    # the trainer's calling convention: a permuted (non-contiguous) view, default draws
    Xv = torch.from_numpy(g["X"]).to(dev).permute(0, 2, 1).contiguous().permute(0, 2, 1)
    assert not Xv.is_contiguous()
    orig = Xv.clone()
    X2, m2 = mlsp.deform_input(Xv, None, 'volume_based_radius', dev)
    assert X2.data_ptr() == Xv.data_ptr()
    moved = (X2 != orig).any(1)
    assert torch.equal(moved, m2[:, 0] > 0) and (m2[:, 0].sum(1) >= 20).all()
    _, _, cand = oc.collapse_to_point(g["X"], -np.ones(5, np.int32), g["noise"])
    for b in range(5):                                   # the moved set is the 0.5-ball of ONE candidate point of the original cloud
        o = orig[b].cpu().numpy()
        d2 = ((o[:, :, None] - o[:, None, :]) ** 2).sum(0)
        sets = {tuple(np.nonzero(d2[p] <= 0.25 + 1e-6)[0]) for p in np.nonzero(cand[b])[0]} | \
               {tuple(np.nonzero(d2[p] <= 0.25 - 1e-6)[0]) for p in np.nonzero(cand[b])[0]}
        assert tuple(np.nonzero(moved[b].cpu().numpy())[0]) in sets
