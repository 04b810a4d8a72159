"""Minimal stand-in for python-pcl so that the reference trainers' `import pcl` and their two uses of it keep working
on a ROCm box without the (unpinned, conda-only) package:

    cloud = pcl.PointCloud(); cloud.from_array(pts)                               (PointDA/trainer.py:525-528, mlsp.py:246-250)
    ne = cloud.make_NormalEstimation(); ne.set_SearchMethod(cloud.make_kdtree()); ne.set_KSearch(k); ne.compute().to_array()
    kdtree = cloud.make_kdtree_flann(); ind, sqdist = kdtree.radius_search_for_cloud(cloud, radius, K)

Normals run on the GPU through mlsp_amd.labels (parity unpinned); the radius search is a plain torch
implementation kept only for API completeness -- `mlsp_amd.mlsp.cal_density` never goes through it.
"""
import numpy as np
import torch


class _Normals:
    def __init__(self, arr):
        self._a = arr
        self.size = arr.shape[0]

    def __getitem__(self, i):
        return self._a[i]

    def to_array(self):
        return self._a


class _NormalEstimation:
    def __init__(self, cloud):
        self.cloud, self.k, self.radius = cloud, 20, None

    def set_SearchMethod(self, tree):
        pass

    def set_KSearch(self, k):
        self.k = int(k)

    def set_RadiusSearch(self, r):
        self.radius = float(r)

    def compute(self):
        from mlsp_amd import labels
        if self.radius is not None:
            raise NotImplementedError("only KSearch normal estimation is provided (PointDA/trainer.py:173-188)")
        pts = torch.from_numpy(self.cloud._pts).cuda().unsqueeze(0)
        n = labels.estimate_normals(pts, near=self.k)[0].cpu().numpy()
        return _Normals(np.concatenate([n, np.zeros((n.shape[0], 1), np.float32)], axis=1))   # [nx, ny, nz, curvature=0]


class _KdTreeFlann:
    def __init__(self, cloud):
        self.cloud = cloud

    def radius_search_for_cloud(self, query, radius, K):
        ref = torch.from_numpy(self.cloud._pts)
        q = torch.from_numpy(query._pts)
        d2 = torch.cdist(q, ref) ** 2
        d2m = torch.where(d2 < radius * radius, d2, torch.full_like(d2, float("inf")))
        val, ind = d2m.topk(min(K, ref.shape[0]), dim=1, largest=False)
        ok = torch.isfinite(val)
        return torch.where(ok, ind, torch.zeros_like(ind)).numpy(), torch.where(ok, val, torch.zeros_like(val)).numpy()


class PointCloud:
    def __init__(self, pts=None):
        self._pts = np.zeros((0, 3), np.float32) if pts is None else np.asarray(pts, np.float32)

    def from_array(self, arr):
        self._pts = np.ascontiguousarray(arr, dtype=np.float32)

    def to_array(self):
        return self._pts

    @property
    def size(self):
        return self._pts.shape[0]

    def make_NormalEstimation(self):
        return _NormalEstimation(self)

    def make_kdtree(self):
        return object()

    def make_kdtree_flann(self):
        return _KdTreeFlann(self)
