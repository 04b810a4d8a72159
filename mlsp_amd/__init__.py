"""mlsp_amd -- MI355X-native (gfx950, HIP) implementation of the data-parallel hot path of
VITA-Group/MLSP: DGCNN encoder forward/backward + the three masked-local-structure heads and
their losses, behind the reference's own Python API (PointDA/Models.py, PointDA/model_utils.py,
MLSP/mlsp.py).  Compute runs in hand-written HIP kernels reached through the C ABI of
include/mlsp_hip.h (mlsp_amd/libmlsp_hip.so); there is no CPU fallback."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
