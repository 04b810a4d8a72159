"""Mirror of the reference's MLSP/PCM.py (point-cloud mixup of the source branch) with the farthest point
sampling on device (mlsp_amd.pc_utils.farthest_point_sample -> csrc/sa.hip).  Same names, arguments and returns."""
import numpy as np
import torch

from . import pc_utils


def mix_shapes(args, X, Y, rng=None):
    """MLSP/PCM.py:6-38.  X [B,C,N], Y [B] -> mixed_X, (Y_a, Y_b, lam).  `rng` (optional dict with 'index', 'lam', 'start_a',
    'start_b', 'points_perm') replaces the random draws for reproducible runs."""
    rng = rng or {}
    batch_size, _, num_points = X.size()
    dev = X.device
    index = rng["index"].to(dev) if "index" in rng else torch.randperm(batch_size).to(dev)
    if "lam" in rng:
        lam = rng["lam"]
    else:
        lam = np.random.beta(args.mixup_params, args.mixup_params) if args.mixup_params > 0 else 1.0
    num_pts_a = round(lam * num_points)
    num_pts_b = num_points - round(lam * num_points)
    _, pts_vals_a = pc_utils.farthest_point_sample(args, X, num_pts_a, start=rng.get("start_a"))
    _, pts_vals_b = pc_utils.farthest_point_sample(args, X[index, :], num_pts_b, start=rng.get("start_b"))
    mixed_X = torch.cat((pts_vals_a, pts_vals_b), 2)
    points_perm = rng["points_perm"].to(dev) if "points_perm" in rng else torch.randperm(num_points).to(dev)
    mixed_X = mixed_X[:, :, points_perm]
    return mixed_X, (Y.clone(), Y[index].clone(), lam)


def mix_shapes_segmentation(args, X, Y, rng=None):
    """MLSP/PCM.py:40-73 (PointSegDA/trainer.py:306).  X [B,C,N], Y [B,N] per-point labels -> mixed_X [B,C,N], mixed_Y [B,N]:
    the labels follow their points through both FPS selections and the shared point permutation."""
    rng = rng or {}
    batch_size, _, num_points = X.size()
    dev = X.device
    index = rng["index"].to(dev) if "index" in rng else torch.randperm(batch_size).to(dev)
    if "lam" in rng:
        lam = rng["lam"]
    else:
        lam = np.random.beta(args.mixup_params, args.mixup_params) if args.mixup_params > 0 else 1.0
    num_pts_a = round(lam * num_points)
    num_pts_b = num_points - round(lam * num_points)
    idx_a, pts_vals_a = pc_utils.farthest_point_sample(args, X, num_pts_a, start=rng.get("start_a"))
    idx_b, pts_vals_b = pc_utils.farthest_point_sample(args, X[index, :], num_pts_b, start=rng.get("start_b"))
    mixed_X = torch.cat((pts_vals_a, pts_vals_b), 2)
    mixed_Y = torch.cat((Y.gather(1, idx_a), Y[index, :].gather(1, idx_b)), 1)
    points_perm = rng["points_perm"].to(dev) if "points_perm" in rng else torch.randperm(num_points).to(dev)
    return mixed_X[:, :, points_perm], mixed_Y[:, points_perm]


def calc_loss(args, logits, mixup_vals, criterion):
    """MLSP/PCM.py:76-89."""
    Y_a, Y_b, lam = mixup_vals
    loss = lam * criterion(logits['cls'], Y_a) + (1 - lam) * criterion(logits['cls'], Y_b)
    loss *= (1 - args.DefRec_weight)
    return loss


def calc_loss_ptrans(args, logits, mixup_vals, criterion):
    """MLSP/PCM.py:91-104: the mixup loss on raw logits (no DefRec re-weighting)."""
    Y_a, Y_b, lam = mixup_vals
    return lam * criterion(logits, Y_a) + (1 - lam) * criterion(logits, Y_b)
