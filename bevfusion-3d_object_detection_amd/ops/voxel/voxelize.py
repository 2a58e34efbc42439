"""Host mirror of BF/ops/voxel/voxelize.py: `Voxelization` module and `voxelization` function."""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn.modules.utils import _pair

from .voxel_layer import dynamic_voxelize, hard_voxelize


class _Voxelization(Function):
    """points [N, >=3] -> (voxels [M,P,F], coors [M,3] xyz, num_points [M]) or, when
    max_points == -1 or max_voxels == -1, per-point coors [N,3] (reference: voxelize.py:13-74)."""

    @staticmethod
    def forward(ctx, points, voxel_size, coors_range, max_points=35, max_voxels=20000, deterministic=True):
        if max_points == -1 or max_voxels == -1:
            coors = points.new_zeros(size=(points.size(0), 3), dtype=torch.int)
            dynamic_voxelize(points, coors, voxel_size, coors_range, 3)
            return coors
        # caller-allocated, zero-filled outputs: the op's contract (voxelize.py:51-53)
        voxels = points.new_zeros(size=(max_voxels, max_points, points.size(1)))
        coors = points.new_zeros(size=(max_voxels, 3), dtype=torch.int)
        num_points_per_voxel = points.new_zeros(size=(max_voxels,), dtype=torch.int)
        voxel_num = hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range,
                                  max_points, max_voxels, 3, deterministic)
        return voxels[:voxel_num], coors[:voxel_num], num_points_per_voxel[:voxel_num]


voxelization = _Voxelization.apply


class Voxelization(nn.Module):
    """Same constructor and attributes as the reference module (voxelize.py:80-121):
    `max_voxels` may be an int or a (train, test) pair; coordinate order is (x, y, z)."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, deterministic=True):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else _pair(max_voxels)
        self.deterministic = deterministic
        pcr = torch.tensor(point_cloud_range, dtype=torch.float32)
        vs = torch.tensor(voxel_size, dtype=torch.float32)
        grid_size = torch.round((pcr[3:] - pcr[:3]) / vs).long()
        self.grid_size = grid_size
        self.pcd_shape = [*grid_size[:2], 1]

    def forward(self, input):
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        return voxelization(input, self.voxel_size, self.point_cloud_range, self.max_num_points, max_voxels,
                            self.deterministic)

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, "
                f"point_cloud_range={self.point_cloud_range}, max_num_points={self.max_num_points}, "
                f"max_voxels={self.max_voxels}, deterministic={self.deterministic})")
