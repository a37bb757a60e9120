"""Drop-in for the reference's ``utilities`` package (utilities/__init__.py:1-9):
``ICP``, ``voxel_downsample``, ``rotation_search``, ``OccupancyGrid2D`` and
``PoseGraph2D`` run on the MI355X through libicpmi.so.  ``feature_based_alignment``
(off by default in the reference, unseeded RANSAC) is outside the accelerated path
and raises NotImplementedError."""
from . import features, icp, mapping, pose_graph  # noqa: F401

ICP, voxel_downsample = icp.ICP, icp.voxel_downsample
rotation_search, feature_based_alignment = features.rotation_search, features.feature_based_alignment
OccupancyGrid2D = mapping.OccupancyGrid2D
PoseGraph2D = pose_graph.PoseGraph2D
pose_matrix_to_vec, pose_vec_to_matrix = pose_graph.pose_matrix_to_vec, pose_graph.pose_vec_to_matrix
relative_transform_vec = pose_graph.relative_transform_vec

__all__ = ["ICP", "voxel_downsample", "rotation_search", "feature_based_alignment", "OccupancyGrid2D", "PoseGraph2D",
           "pose_matrix_to_vec", "pose_vec_to_matrix", "relative_transform_vec"]
