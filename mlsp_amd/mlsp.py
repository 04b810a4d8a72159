"""Mirror of the tensor part of the reference's MLSP/mlsp.py (losses of the three heads) on the HIP path.

Same function names, argument order and return conventions (0-dim tensors supporting .item() and
.backward()) as MLSP/mlsp.py:115-238, 275-287, 430-454.  `deform_input` (MLSP/mlsp.py:10-51) runs
on device for DefRec_dist == 'volume_based_voxels'; `scan_input` (:54-89) is one kernel per batch with the float64 arithmetic of the numpy original.
`cal_density` is provided on device by mlsp_amd/labels.py (f-1, parity unpinned: python-pcl is third-party).
"""
import numpy as np
import torch

from . import _lib
from . import pc_utils

from . import functional as Fh
from .labels import cal_density, cal_density_gpu, estimate_normals   # noqa: F401  (mlsp.py:240-272)

DefRec_SCALER = 20.0   # MLSP/mlsp.py:7


def deform_input(X, lookup, DefRec_dist='volume_based_voxels', device='cuda:0', groups=1, region_ids=None, noise=None, choice=None):
    """MLSP/mlsp.py:10-51.  X [B,C,N] is deformed IN PLACE (as in the reference) and returned with the 0/1 mask [B,C,N].
    Per cloud the first `groups` voxels of a random visiting order that hold >= 40 points collapse to a Gaussian blob
    (std sqrt(0.001), pc_utils.draw_from_gaussian) around the voxel centre `lookup[i]`.
    The random inputs may be injected for reproducible runs: `region_ids` (a permutation of the 27 voxel ids; default
    np.random.permutation like :27) and `noise` [B,3,N] standard normal (default torch.randn on the device)."""
    if DefRec_dist not in ('volume_based_voxels', 'volume_based_radius'):
        raise ValueError("DefRec_dist must be 'volume_based_voxels' or 'volume_based_radius'")
    lib = _lib.load()
    _lib.require_gpu(X)
    # every call site of the trainer passes `data.permute(0, 2, 1)` of a [B,N,3] loader tensor (PointDA/trainer.py:381-387,
    # 428-431, 538-541): a non-contiguous view.  The kernels work on a dense [B,C,N] copy; the in-place contract of the
    # reference (it writes into X and returns it) is kept by copying the result back through the view.
    X_arg = X
    if not X.is_contiguous() or X.dtype != torch.float32:
        X = X.float().contiguous()
    B, C, N = X.shape
    if DefRec_dist == 'volume_based_radius':
        # pc_utils.collapse_to_point per cloud (utils/pc_utils.py:76-111): a random point with >= MIN_POINTS points inside RADIUS
        # attracts everything within RADIUS of it; `choice` [B] pins the picked points (default: uniform among the candidates)
        if C != 3:
            raise ValueError("'volume_based_radius' needs [B,3,N] clouds (the reference assigns a 3-row block)")
        if noise is None:
            noise = torch.randn(B, 3, N, device=X.device)
        scaled = (noise.to(X.device, torch.float32) * float(np.sqrt(0.001))).contiguous()
        ch = None if choice is None else torch.as_tensor(np.asarray(choice), dtype=torch.int32).to(X.device)
        u = torch.rand(B, device=X.device)
        mask = torch.empty_like(X)
        chosen = torch.empty(B, dtype=torch.int32, device=X.device)
        _lib.check(lib.mlsp_collapse_to_point_f32(X.data_ptr(), B, N, _lib.ptr(ch), u.data_ptr(), scaled.data_ptr(),
                                                  float(np.float32(pc_utils.RADIUS ** 2)), pc_utils.MIN_POINTS, mask.data_ptr(),
                                                  chosen.data_ptr(), _lib.stream()), "mlsp_collapse_to_point_f32")
        if X is not X_arg:
            X_arg.copy_(X)
        return X_arg, mask
    n = pc_utils.NREGIONS
    regions = pc_utils.assign_region_to_point(X, device).to(torch.int32)
    if region_ids is None:
        region_ids = np.random.permutation(n ** 3)
    order = torch.as_tensor(np.asarray(region_ids), dtype=torch.int32).to(X.device)
    if noise is None:
        noise = torch.randn(B, 3, N, device=X.device)
    scaled = (noise.to(X.device, torch.float32) * float(np.sqrt(0.001))).contiguous()
    look = lookup.to(X.device, torch.float32).contiguous()
    mask = torch.empty_like(X)
    _lib.check(lib.mlsp_deform_regions_f32(X.data_ptr(), B, C, N, regions.data_ptr(), order.data_ptr(), n ** 3, look.data_ptr(),
                                           scaled.data_ptr(), 40, groups, mask.data_ptr(), _lib.stream()), "mlsp_deform_regions_f32")
    if X is not X_arg:
        X_arg.copy_(X)
        mask = mask.to(X_arg.dtype)
    return X_arg, mask


def rotation_matrix_3d(angles):
    """rotate_point_cloud_3d (MLSP/mlsp.py:91-112): the float64 rotation matrix of three angles (host, numpy like the reference)."""
    c, s_ = np.cos(angles), np.sin(angles)
    r1 = np.array([[c[0], 0, s_[0]], [0, 1, 0], [-s_[0], 0, c[0]]])
    r2 = np.array([[1, 0, 0], [0, c[1], -s_[1]], [0, s_[1], c[1]]])
    r3 = np.array([[c[2], -s_[2], 0], [s_[2], c[2], 0], [0, 0, 1]])
    return np.matmul(np.matmul(r1, r2), r3)


def scan_input(X, device, pixel_size=0.07, angles=None):
    """MLSP/mlsp.py:54-64 (+ p_scan :66-89).  X [B,N,C]: single-view "scan" of every cloud -- only the points visible along +x of
    a random rotation survive (the rest become zero); mask [B,N,C] is 0 on the xyz of the survivors, 1 elsewhere.  As in the
    reference the `pixel_size` argument is overwritten by random.uniform(0.045, 0.075) unless `angles` [B,3] is given, which
    pins both random inputs (the rotation angles and the pixel size passed in) for reproducible runs."""
    import random
    lib = _lib.load()
    _lib.require_gpu(X)
    if angles is None:
        pixel_size = random.uniform(0.045, 0.075)
        angles = np.stack([np.random.rand(3) * 2 * np.pi for _ in range(X.shape[0])])
    B, N, C = X.shape
    R = np.stack([rotation_matrix_3d(np.asarray(a, dtype=np.float64)) for a in np.asarray(angles)]).reshape(B, 9)
    Rd = torch.from_numpy(np.ascontiguousarray(R)).to(X.device)
    xin = X.detach().float().contiguous()
    Xs, mask = torch.empty_like(xin), torch.empty_like(xin)
    _lib.check(lib.mlsp_scan_select_f32(xin.data_ptr(), B, N, C, Rd.data_ptr(), int(2 / pixel_size), Xs.data_ptr(), mask.data_ptr(),
                                        _lib.stream()), "mlsp_scan_select_f32")
    if X.dtype == torch.float32:
        X.copy_(Xs)                                   # the reference overwrites its argument too (:61)
        return X, mask
    return Xs, mask


def chamfer_distance(p1, p2, mask):
    """MLSP/mlsp.py:115-153.  p1, p2, mask [B,N,3] (reconstruction_loss passes the permuted clouds): for every MASKED point of p1
    the squared distance to its nearest masked point of p2 (unmasked columns carry a +100 penalty), averaged over the masked
    points of each cloud and summed over the batch.  One direction of the fused kernel behind reconstruction_loss; differentiable
    in p1 and p2."""
    assert p1.size(0) == p2.size(0) and p1.size(1) == p2.size(1)
    return Fh.chamfer_dir(p1, p2, mask[:, :, 0])


def reconstruction_loss(pred, gold, mask):
    """MLSP/mlsp.py:156-182.  pred [B,N,3]; gold, mask [B,3,N] -> mean-batch symmetric masked Chamfer."""
    return Fh.chamfer_masked(pred, gold, mask, 1.0 / pred.size(0))


def calc_loss(args, logits, labels, mask):
    """MLSP/mlsp.py:222-229."""
    pred = logits['DefRec']
    return Fh.chamfer_masked(pred, labels, mask, args.DefRec_weight * DefRec_SCALER / pred.size(0))


def calc_scan_loss(args, logits, labels, mask):
    """MLSP/mlsp.py:231-238."""
    pred = logits['Rec_scan']
    return Fh.chamfer_masked(pred, labels, mask, args.Scan_Rec_weight * DefRec_SCALER / pred.size(0))


def normal_prediction_loss(pred, gt, weight=None):
    """MLSP/mlsp.py:275-283; `weight` [B,N] gives the masked form of PointDA/trainer.py:551-556."""
    return Fh.normal_loss(pred, gt, weight, 1.0)


def calc_normal_loss(args, prediction, labels):
    """MLSP/mlsp.py:285-287."""
    return Fh.normal_loss(prediction, labels, None, args.normal_pred_weight)


def calc_masked_normal_loss(args, prediction, labels, mask_cord):
    """The loss the trainer inlines at PointDA/trainer.py:551-556 (mask_cord = mask*26+1)."""
    return Fh.normal_loss(prediction, labels, mask_cord, args.normal_pred_weight)


def densityloss(args, logits, target, target_vec, mask=None):
    """MLSP/mlsp.py:430-454 -> (kl, mae)."""
    return Fh.density_loss(logits['density'], logits['density_mse'], target_vec, target, mask, args.Density_weight)
