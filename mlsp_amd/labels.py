"""On-device label generators of the MLSP target branch.

The reference computes these per cloud on the CPU through python-pcl inside the training step
(`mlsp.cal_density`, MLSP/mlsp.py:240-272; `kSearchNormalEstimation`, PointDA/trainer.py:173-188).  python-pcl is an
unpinned third-party package that is not part of the reference tree: PARITY IS UNPINNED.  The functions keep the
reference's names, arguments and return types; the arithmetic restates the published PCL/FLANN algorithms
(mlsp_amd/csrc/labels.hip) and is pinned by tests to a numpy restatement and to closed-form shapes.
"""
import numpy as np
import torch

from . import _lib
from . import functional as Fh


def radius_count(batch_pts, radius, K=100):
    """batch_pts [B,N,3] (cuda) -> int32 [B,N]: neighbours within `radius` (self included), at most K, not counting the
    cloud's point 0 (the reference's `(ind != 0).sum(1)`, mlsp.py:254)."""
    lib = _lib.load()
    x = batch_pts.detach().float().contiguous()
    _lib.require_gpu(x)
    B, N, C = x.shape
    out = torch.empty((B, N), dtype=torch.int32, device=x.device)
    _lib.check(lib.mlsp_radius_count_f32(x.data_ptr(), C, B, N, float(radius), int(K), out.data_ptr(), _lib.stream()),
               "mlsp_radius_count_f32")
    return out


def cal_density_gpu(batch_pts, radius, num_cls, pergroup=2, shift=0, K=100):
    """Device-resident form of cal_density: (soft label [B,N,num_cls] float32, count [B,N] float32) as cuda tensors."""
    row = radius_count(batch_pts, radius, K).float() - shift
    row = row.clamp(min=0, max=(num_cls - 1) * pergroup)
    c1 = torch.floor(row / pergroup).long()
    c2 = torch.ceil(row / pergroup).long()
    eye = torch.eye(num_cls, device=row.device)
    return (eye[c1] + eye[c2]) / 2.0, row


def cal_density(batch_pts, radius, num_cls, pergroup=2, shift=0, K=100):
    """Drop-in for mlsp.cal_density (MLSP/mlsp.py:240-272): returns (np.ndarray [B,N,num_cls], np.ndarray [B,N])."""
    cls, row = cal_density_gpu(batch_pts, radius, num_cls, pergroup, shift, K)
    return cls.double().cpu().numpy(), row.cpu().numpy().astype(np.int64)


def estimate_normals(batch_pts, near=20):
    """batch_pts [B,N,3] (cuda) -> unit normals [B,N,3]: PCA of each point's `near` nearest neighbours (self included),
    oriented towards the origin -- the batched device form of kSearchNormalEstimation (PointDA/trainer.py:173-188)."""
    lib = _lib.load()
    x = batch_pts.detach().float().contiguous()
    _lib.require_gpu(x)
    B, N, C = x.shape
    near = int(near)
    g = Fh.knn_graph(x.view(B * N, C)[:, :3], B, N, near, need_reverse=False)
    out = torch.empty((B, N, 3), dtype=torch.float32, device=x.device)
    _lib.check(lib.mlsp_knn_normals_f32(x.data_ptr(), C, g.idx.data_ptr(), B, N, near, out.data_ptr(), _lib.stream()),
               "mlsp_knn_normals_f32")
    return out
