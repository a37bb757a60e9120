"""icpmi — MI355X-native per-scan ICP + occupancy-mapping core.

Host side of libicpmi.so: ``icpmi.batch`` (batched scan-pair ICP on one GPU),
``icpmi.dist`` (the same batch sharded over the GPUs of a node), ``icpmi.synth``
(synthetic scans).  The drop-in modules with the reference's own names live in
the sibling package ``utilities`` (``utilities.icp``, ``utilities.mapping``).
"""
from ._lib import IcpmiError, build, lib  # noqa: F401

__version__ = "0.1"
