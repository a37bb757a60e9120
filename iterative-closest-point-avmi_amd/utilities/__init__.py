"""Drop-in for the reference's ``utilities`` package (utilities/__init__.py:1-9),
hot-path part: ``ICP``, ``voxel_downsample`` and ``OccupancyGrid2D`` run on the
MI355X through libicpmi.so.  ``features`` / ``pose_graph`` are outside the
accelerated path and are not re-implemented here."""
from .icp import ICP, voxel_downsample  # noqa: F401
from .features import rotation_search  # noqa: F401
from .mapping import OccupancyGrid2D  # noqa: F401
