"""on-device label generators vs the numpy restatement (parity unpinned: no python-pcl) and closed forms."""
import numpy as np
import pytest
import torch

from oracle import knn_canon, labels_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _cloud(seed, B, N):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, N, 3, generator=g) * 2 - 1
    return x / x.norm(dim=-1, keepdim=True).clamp_min(1e-6) * torch.rand(B, N, 1, generator=g) ** (1 / 3)   # unit ball


@pytest.mark.parametrize("B,N,radius", [(3, 512, 0.2), (2, 1024, 0.135), (1, 300, 0.5)])
def test_cal_density_vs_numpy(dev, B, N, radius):
    from mlsp_amd import labels
    x = _cloud(1, B, N)
    cls, row = labels.cal_density(x.to(dev), radius=radius, num_cls=16, pergroup=2)
    wc, wr = labels_np.cal_density(x.numpy(), radius, 16, 2)
    assert cls.shape == (B, N, 16) and row.shape == (B, N)
    # a pair whose squared distance sits within an ulp of r^2 may fall on either side: allow a handful of +-1 counts
    diff = np.abs(row - wr)
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3
    same = diff == 0
    np.testing.assert_array_equal(cls[same], wc[same])
    np.testing.assert_allclose(cls.sum(-1), 1.0)
    # point 0 of every cloud is never counted (mlsp.py:254) -> its own count excludes itself
    full = (torch.cdist(x[0], x[0]) ** 2 < radius ** 2).sum(1).numpy()
    assert row[0, 0] == min(min(full[0], 100) - 1, 30)


def test_normals_vs_numpy_and_closed_forms(dev):
    from mlsp_amd import labels
    x = _cloud(2, 2, 1024)
    n = labels.estimate_normals(x.to(dev), near=20).cpu().numpy()
    np.testing.assert_allclose(np.linalg.norm(n, axis=-1), 1.0, atol=1e-5)
    for b in range(2):
        idx = knn_canon.knn_point_major(x[b:b + 1], 20)[0]
        want, w = labels_np.knn_normals(x[b].numpy(), idx)
        cos = np.abs((n[b] * want).sum(-1))
        well = (w[:, 1] - w[:, 0]) > 1e-3 * w[:, 2]           # smallest eigenvalue separated -> direction well defined
        assert well.mean() > 0.9 and cos[well].min() > 1 - 1e-4
        assert ((n[b] * -x[b].numpy()).sum(-1) >= -1e-6).all()   # oriented towards the origin
    # plane z = 0.3: normal = (0,0,-1) (towards the origin from above)
    g = torch.Generator().manual_seed(3)
    pl = torch.rand(1, 600, 3, generator=g) * 2 - 1
    pl[..., 2] = 0.3
    npl = labels.estimate_normals(pl.to(dev), near=20).cpu().numpy()[0]
    np.testing.assert_allclose(npl, np.tile([0, 0, -1.0], (600, 1)), atol=1e-4)
    # sphere of radius 1: normal = -p (radial, towards the centre)
    sp = torch.randn(1, 4096, 3, generator=g)
    sp = sp / sp.norm(dim=-1, keepdim=True)
    nsp = labels.estimate_normals(sp.to(dev), near=20).cpu().numpy()[0]
    assert ((nsp * -sp[0].numpy()).sum(-1) > 0.97).mean() > 0.99
