"""oracle/ref_corrupt_np.py against the reference's own outputs (tests/golden/deform_*.npz, pcm_*.npz).  CPU only."""
import os

import numpy as np

from oracle import ref_corrupt_np as oc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def check_deform_against_reference(X_in, X_out, mask, g, groups):
    """Everything deterministic must be identical: the mask, the untouched coordinates; the replaced points are Gaussian
    around the chosen voxel centres (numpy's RNG stream cannot be reproduced): checked statistically."""
    ref_mask, ref_out = g["mask_g%d" % groups], g["X_out_g%d" % groups]
    assert np.array_equal(mask, ref_mask)
    keep = ref_mask[:, 0] == 0
    for b in range(X_in.shape[0]):
        assert np.array_equal(X_out[b][:, keep[b]], ref_out[b][:, keep[b]])
        hit = ~keep[b]
        assert hit.sum() >= 40
        regs = np.unique(g["regions"][b][hit])
        assert len(regs) <= groups
        for r in regs:
            sel = hit & (g["regions"][b] == r)
            d = X_out[b][:, sel] - g["lookup"][r][:, None]
            assert np.abs(d.mean(1)).max() < 4 * np.sqrt(0.001 / sel.sum()) + 1e-3
            assert abs(d.std() - np.sqrt(0.001)) < 0.3 * np.sqrt(0.001)


def test_region_assignment_bit_exact():
    g = dict(np.load(os.path.join(GOLD, "deform_s5_B6_N1024.npz")))
    assert np.array_equal(oc.assign_region(g["X"]), g["regions"])


def test_deform_oracle_vs_reference():
    g = dict(np.load(os.path.join(GOLD, "deform_s5_B6_N1024.npz")))
    rs = np.random.RandomState(0)
    for groups in (1, 3):
        noise = rs.randn(*g["X"].shape).astype(np.float32)
        X_out, mask = oc.deform(g["X"], g["lookup"], g["perm_g%d" % groups], noise, groups)
        check_deform_against_reference(g["X"], X_out, mask, g, groups)


def test_pcm_oracle_vs_reference():
    g = dict(np.load(os.path.join(GOLD, "pcm_s3_B5_N256.npz")))
    mixed = oc.mix_shapes(g["X"], g["index"], float(g["lam"]), g["start_a"], g["start_b"], g["points_perm"])
    assert np.array_equal(mixed, g["mixed"])


def test_scan_oracle_vs_reference():
    g = dict(np.load(os.path.join(GOLD, "scan_s21_B5_N512.npz")))
    out, mask = oc.scan(g["X"], float(g["pixel_size"]), g["angles"])
    assert np.array_equal(mask, g["mask"]) and np.array_equal(out, g["X_out"])


def test_collapse_to_point_oracle_vs_reference():
    """deform_input(..., 'volume_based_radius'): oracle/ref_corrupt_np.collapse_to_point with the reference's recorded draws reproduces
    the reference's mask exactly and its deformed cloud to fp32 rounding (tests/golden/collapse_*.npz, tools/make_golden.py radius)."""
    g = dict(np.load(os.path.join(GOLD, "collapse_s9_B5_N512.npz")))
    X, mask, cand = oc.collapse_to_point(g["X"], g["choice"], g["noise"])
    assert np.array_equal(mask, g["mask"])
    np.testing.assert_allclose(X, g["X_out"], rtol=0, atol=2e-7)
    assert all(cand[b, g["choice"][b]] for b in range(len(g["choice"])))       # the reference only picks candidates
    keep = g["mask"] == 0
    assert np.array_equal(X[keep], g["X"][keep])
