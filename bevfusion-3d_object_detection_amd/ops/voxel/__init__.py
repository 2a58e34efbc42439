from . import voxel_layer
from .scatter_points import DynamicScatter, dynamic_scatter
from .voxelize import Voxelization, voxelization

__all__ = ["Voxelization", "voxelization", "dynamic_scatter", "DynamicScatter", "voxel_layer"]
