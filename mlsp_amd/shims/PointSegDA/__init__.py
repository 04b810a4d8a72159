"""Import shim: `from PointSegDA.Models import DGCNN_DefRec` resolves to mlsp_amd.seg_models."""
