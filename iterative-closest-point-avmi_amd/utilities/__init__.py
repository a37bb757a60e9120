"""Drop-in for the reference's ``utilities`` package (utilities/__init__.py:1-9):
``ICP``, ``voxel_downsample``, ``rotation_search``, ``OccupancyGrid2D`` and
``PoseGraph2D`` run on the MI355X through libicpmi.so.  ``feature_based_alignment``
(off by default in the reference, unseeded RANSAC) is outside the accelerated path
and raises NotImplementedError."""
from .icp import ICP, voxel_downsample  # noqa: F401
from .features import feature_based_alignment, rotation_search  # noqa: F401
from .mapping import OccupancyGrid2D  # noqa: F401
from .pose_graph import (  # noqa: F401
    PoseGraph2D,
    pose_matrix_to_vec,
    pose_vec_to_matrix,
    relative_transform_vec,
)
