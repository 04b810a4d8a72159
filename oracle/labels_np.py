"""ORACLE -- test infrastructure, NOT the product.  PARITY UNPINNED.

numpy restatement of the two python-pcl calls behind the reference's label generators (MLSP/mlsp.py:240-272,
PointDA/trainer.py:173-188).  python-pcl (conda `sirokujira`, unpinned, README.md:24-25) is not in the reference tree
and not installable here, so nothing from the reference pins these functions; they restate the published algorithms:
  * FLANN/PCL radius search: neighbours with squared L2 distance < radius^2 (the query itself included), at most K,
    nearest first; the reference then counts `(ind != 0).sum(1)`, i.e. every returned index except 0.
  * PCL NormalEstimation with KSearch(k): covariance of the k nearest points (self included), eigenvector of the
    smallest eigenvalue, flipped so that n . (viewpoint - p) >= 0 with viewpoint (0,0,0).
"""
import numpy as np


def radius_count(pts, radius, K=100):
    """pts [N,3] float32 -> int64 [N]"""
    p = pts.astype(np.float32)
    d = p[:, None, :] - p[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    r2 = np.float32(radius) * np.float32(radius)
    within = d2 < r2
    cnt = np.minimum(within.sum(1), K) - within[:, 0].astype(np.int64)
    return np.maximum(cnt, 0)


def cal_density(batch_pts, radius, num_cls, pergroup=2, shift=0, K=100):
    """MLSP/mlsp.py:240-272 with radius_count in place of the pcl call."""
    cls_all, row_all = [], []
    for pts in batch_pts:
        row = radius_count(np.asarray(pts), radius, K) - shift
        row[row < 0] = 0
        row[row > (num_cls - 1) * pergroup] = (num_cls - 1) * pergroup
        c1 = np.floor(row / pergroup).astype(np.int32)
        c2 = np.ceil(row / pergroup).astype(np.int32)
        eye = np.identity(num_cls)
        cls_all.append((eye[c1] + eye[c2]) / 2.0)
        row_all.append(row)
    return np.array(cls_all), np.array(row_all)


def knn_normals(pts, idx):
    """pts [N,3], idx [N,k] (self included) -> unit normals [N,3] oriented towards the origin."""
    p = pts.astype(np.float64)
    nb = p[idx]                                       # [N,k,3]
    c = nb - nb.mean(1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", c, c) / idx.shape[1]
    w, v = np.linalg.eigh(cov)
    n = v[:, :, 0]
    flip = (n * (-p)).sum(1) < 0
    n[flip] = -n[flip]
    return n.astype(np.float32), w
