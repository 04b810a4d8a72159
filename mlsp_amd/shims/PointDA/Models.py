"""Shim for the reference import path (PointDA/trainer.py:14 `from PointDA.Models import PointNet, DGCNN`)."""
from mlsp_amd.Models import *            # noqa: F401,F403
from mlsp_amd.Models import DGCNN, RegionReconstruction, Normal_prediction, Density_prediction, K  # noqa: F401


class PointNet:                           # `--model pointnet` is outside the accelerated path
    def __init__(self, *a, **k):
        raise NotImplementedError("PointNet is not part of the MI355X hot path; use the reference's own module")
