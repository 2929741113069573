"""Shim for the names imported at pointstowood/src/model.py:3,5."""
from oracle.ops import knn_interpolate, radius, voxel_grid, knn, global_max_pool  # noqa: F401
from .conv import MessagePassing  # noqa: F401


class PointNetConv(MessagePassing):
    """Placeholder: model.py:5 imports this name and model.py:6 immediately shadows it
    with the reference's own src.pointnet.PointNetConv."""
