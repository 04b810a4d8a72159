"""Shim for MLSP/PCM.py: mixup with the farthest point sampling on device (mlsp_amd.PCM)."""
from mlsp_amd.PCM import mix_shapes, mix_shapes_segmentation, calc_loss, calc_loss_ptrans   # noqa: F401
