"""Shim for MLSP/mlsp.py: tensor losses and on-device input corruption from mlsp_amd.mlsp (HIP)."""
from mlsp_amd.mlsp import *              # noqa: F401,F403
from mlsp_amd.mlsp import (DefRec_SCALER, deform_input, scan_input, reconstruction_loss, calc_loss, calc_scan_loss, normal_prediction_loss,  # noqa: F401
                           calc_normal_loss, calc_masked_normal_loss, densityloss)
