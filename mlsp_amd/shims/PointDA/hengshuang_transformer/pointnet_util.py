"""Drop-in for PointDA/hengshuang_transformer/pointnet_util.py (set abstraction, multi-scale grouping, feature propagation) backed by
the MI355X kernels."""
from mlsp_amd.pointnet2 import (farthest_point_sample, index_points, query_ball_point, knn_point, sample_and_group,  # noqa: F401
                                sample_and_group_all, PointNetSetAbstraction, PointNetSetAbstractionMsg, PointNetFeaturePropagation)
