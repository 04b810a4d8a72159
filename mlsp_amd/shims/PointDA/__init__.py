"""Import shim: `from PointDA.Models import DGCNN` / `from PointDA.model_utils import ...` resolve to mlsp_amd."""
