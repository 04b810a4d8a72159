"""Shim for `from model_utils import ...` / `from PointDA.model_utils import ...` (PointDA/Models.py:4,10)."""
from mlsp_amd.model_utils import *       # noqa: F401,F403
from mlsp_amd.model_utils import knn, get_graph_feature, conv_2d, fc_layer, transform_net, classifier, density_classifier  # noqa: F401
