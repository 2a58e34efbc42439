"""Voxel operators: hard / dynamic voxelization and dynamic scatter (same public names as the reference's
bevfusion/ops/voxel package), backed by csrc/voxelize.hip and csrc/scatter.hip."""
from . import voxel_layer  # the C-ABI backed replacement of the reference's pybind module of the same name
from .scatter_points import DynamicScatter as DynamicScatter
from .scatter_points import dynamic_scatter as dynamic_scatter
from .voxelize import Voxelization as Voxelization
from .voxelize import voxelization as voxelization

__all__ = ["voxel_layer", "Voxelization", "voxelization", "DynamicScatter", "dynamic_scatter"]
