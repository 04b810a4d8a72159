"""Shim for MLSP/PCM.py: mixup with the farthest point sampling on device (mlsp_amd.PCM)."""
from mlsp_amd.PCM import mix_shapes, calc_loss   # noqa: F401
