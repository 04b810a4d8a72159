"""Mirror of the tensor part of the reference's MLSP/mlsp.py (losses of the three heads) on the HIP path.

Same function names, argument order and return conventions (0-dim tensors supporting .item() and
.backward()) as MLSP/mlsp.py:115-238, 275-287, 430-454.  Host-side input corruption and the pcl-based
corruption of that file (deform_input, scan_input) is out of this path's scope;
`cal_density` is provided on device by mlsp_amd/labels.py (f-1, parity unpinned: python-pcl is third-party).
"""
import torch

from . import functional as Fh
from .labels import cal_density, cal_density_gpu, estimate_normals   # noqa: F401  (mlsp.py:240-272)

DefRec_SCALER = 20.0   # MLSP/mlsp.py:7


def chamfer_distance(p1, p2, mask):
    """MLSP/mlsp.py:115-153 is only ever called in symmetric pairs by reconstruction_loss; the fused
    kernel computes both directions at once, so the one-directional entry is not provided."""
    raise NotImplementedError("use reconstruction_loss(pred, gold, mask): both directions are fused in one kernel")


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
