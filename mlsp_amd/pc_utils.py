"""Device versions of the point-cloud helpers the MLSP input corruption and PCM mixing lean on.

Mirrors utils/pc_utils.py of the reference for the functions on that path: `region_mean` (:13-29),
`assign_region_to_point` (:33-73), `farthest_point_sample(args, xyz, npoint)` (:137-161); same names, argument order and
return values.  The reference implements them as Python loops of tensor ops (27 region passes; `npoint` FPS iterations);
here each is one HIP kernel (csrc/corrupt.hip, csrc/sa.hip).  No CPU fallback.
"""
import numpy as np
import torch

from . import _lib

NREGIONS = 3           # utils/pc_utils.py:10
MIN_POINTS = 20
RADIUS = 0.5


def region_mean(num_regions):
    """utils/pc_utils.py:13-29: centres of the num_regions^3 voxels of [-1,1]^3 in the reference's order (host table)."""
    n = num_regions
    d = 2 / n
    ax = 1 - d * (np.arange(n - 1, -1, -1) + 0.5)
    return np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)


def assign_region_to_point(X, device=None):
    """utils/pc_utils.py:33-73.  X [B,C,N] -> int64 [B,N] voxel id of every point (0 for points on a voxel face)."""
    lib = _lib.load()
    x = X.detach()
    if not x.is_contiguous():
        x = x.contiguous()
    _lib.require_gpu(x)
    B, C, N = x.shape
    n = NREGIONS
    d = 2 / n
    thr = torch.tensor([-1 + i * d for i in range(n + 1)], dtype=torch.float64).to(torch.float32).to(x.device)   # torch rounds the scalars
    clip = float(torch.tensor(0.99999999, dtype=torch.float32))
    Y = torch.empty((B, N), dtype=torch.int32, device=x.device)
    _lib.check(lib.mlsp_region_assign_f32(x.data_ptr(), B, C, N, thr.data_ptr(), n, clip, Y.data_ptr(), _lib.stream()),
               "mlsp_region_assign_f32")
    return Y.long()


def farthest_point_sample(args, xyz, npoint, start=None):
    """utils/pc_utils.py:137-161.  xyz [B,C,N] -> (indices [B,npoint] int64, values [B,C,npoint]).  `start` [B] pins the first
    sample (the reference draws it with torch.randint, :150)."""
    from .pointnet2 import farthest_point_sample as fps_rows
    B, C, N = xyz.shape
    if npoint == 0:
        return torch.zeros(B, 0, dtype=torch.long, device=xyz.device), torch.zeros(B, C, 0, device=xyz.device)
    rows = xyz.detach().transpose(1, 2).contiguous()             # [B,N,C] point-major for the kernel
    idx = fps_rows(rows, npoint, start=start)
    vals = torch.gather(xyz, 2, idx[:, None, :].expand(-1, C, -1))
    return idx, vals
