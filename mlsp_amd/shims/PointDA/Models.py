"""Shim for the reference import path (PointDA/trainer.py:14 `from PointDA.Models import PointNet, DGCNN`)."""
from mlsp_amd.Models import *            # noqa: F401,F403
from mlsp_amd.Models import PointNet, DGCNN, RegionReconstruction, Normal_prediction, Density_prediction, K  # noqa: F401
