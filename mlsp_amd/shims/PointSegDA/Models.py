"""Shim for the reference import path of the segmentation variant (PointSegDA/trainer.py imports `Models`)."""
from mlsp_amd.seg_models import *            # noqa: F401,F403
from mlsp_amd.seg_models import DGCNN_DefRec, segmentation, DeformationReconstruction, Normal_prediction, Density_prediction, K  # noqa: F401
