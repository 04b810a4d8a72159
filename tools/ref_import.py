"""Import the upstream reference (VITA-Group/MLSP, mounted read-only at /root/reference)
in THIS container so golden vectors can be captured.  Never runs on the GPU box
(/root/reference does not exist there) and is never imported by the product path.

The reference's top-of-file imports pull in CUDA-only / absent pip packages that the
DGCNN hot path does not use (knn_cuda, pointnet2_ops, pcl, timm, termcolor, ...); they
are replaced by empty stub modules (SURVEY.md section 8c recipe).
"""
import sys
import types

REF_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    class KNN:  # knn_cuda.KNN(k, transpose_mode) -- only used by the transformer family
        def __init__(self, k=1, transpose_mode=False):
            self.k = k

    _stub("knn_cuda", KNN=KNN)
    timm = _stub("timm")
    models = _stub("timm.models")
    layers = _stub("timm.models.layers", DropPath=object, trunc_normal_=lambda *a, **k: None)
    timm.models = models
    models.layers = layers
    _stub("timm.scheduler", CosineLRScheduler=object)
    p2 = _stub("pointnet2_ops")
    p2.pointnet2_utils = _stub("pointnet2_ops.pointnet2_utils")
    _stub("pcl")
    _stub("termcolor", colored=lambda s, *a, **k: s)
    _stub("easydict", EasyDict=dict)
    _stub("h5py")


def import_reference():
    """Returns (Models, model_utils, mlsp) modules of the reference."""
    install_stubs()
    for p in (REF_ROOT + "/PointDA", REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import warnings
    warnings.filterwarnings("ignore")
    from PointDA import Models as RefModels  # noqa
    from PointDA import model_utils as ref_mu  # noqa
    from MLSP import mlsp as ref_mlsp  # noqa
    return RefModels, ref_mu, ref_mlsp


def ref_args(dropout=0.5, cuda=False):
    import argparse
    return argparse.Namespace(num_class=10, dropout=dropout, model="dgcnn", encoder_type=None,
                              cuda=cuda, density_num_class=16, pergroup=2.0,
                              DefRec_weight=0.5, normal_pred_weight=0.5, Density_weight=0.5)
