"""Host-side mirror of projects/BEVFusion/bevfusion/ops/__init__.py (same public names)."""
from .bev_pool import bev_pool, bev_pool_ext
from .voxel import DynamicScatter, Voxelization, dynamic_scatter, voxel_layer, voxelization

__all__ = ["bev_pool", "bev_pool_ext", "Voxelization", "voxelization", "dynamic_scatter", "DynamicScatter",
           "voxel_layer"]
