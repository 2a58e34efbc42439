"""Host mirror of BF/ops/voxel/scatter_points.py: `DynamicScatter` and `dynamic_scatter`."""
import torch
from torch import nn
from torch.autograd import Function

from .voxel_layer import dynamic_point_to_voxel_backward, dynamic_point_to_voxel_forward


class _dynamic_scatter(Function):
    """feats [N,C], coors [N,3] -> (voxel_feats [M,C], voxel_coors [M,3]); reduce in
    {'max','sum','mean'} (reference: scatter_points.py:8-49)."""

    @staticmethod
    def forward(ctx, feats, coors, reduce_type="max"):
        voxel_feats, voxel_coors, point2voxel_map, voxel_points_count = dynamic_point_to_voxel_forward(
            feats, coors, reduce_type)
        ctx.reduce_type = reduce_type
        ctx.save_for_backward(feats, voxel_feats, point2voxel_map, voxel_points_count)
        ctx.mark_non_differentiable(voxel_coors)
        return voxel_feats, voxel_coors

    @staticmethod
    def backward(ctx, grad_voxel_feats, grad_voxel_coors=None):
        feats, voxel_feats, point2voxel_map, voxel_points_count = ctx.saved_tensors
        grad_feats = torch.zeros_like(feats)
        dynamic_point_to_voxel_backward(grad_feats, grad_voxel_feats.contiguous(), feats, voxel_feats,
                                        point2voxel_map, voxel_points_count, ctx.reduce_type)
        return grad_feats, None, None


dynamic_scatter = _dynamic_scatter.apply


class DynamicScatter(nn.Module):
    """Same constructor as the reference (scatter_points.py:54-76).  With batched coors
    [N,4]=(b,x,y,z) every sample is reduced separately and the batch id is put back in front."""

    def __init__(self, voxel_size, point_cloud_range, average_points: bool):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.average_points = average_points

    def forward_single(self, points, coors):
        reduce = "mean" if self.average_points else "max"
        return dynamic_scatter(points.contiguous(), coors.contiguous(), reduce)

    def forward(self, points, coors):
        if coors.size(-1) == 3:
            return self.forward_single(points, coors)
        batch_size = int(coors[-1, 0]) + 1
        voxels, voxel_coors = [], []
        for i in range(batch_size):
            inds = torch.where(coors[:, 0] == i)
            voxel, voxel_coor = self.forward_single(points[inds], coors[inds][:, 1:])
            voxel_coors.append(nn.functional.pad(voxel_coor, (1, 0), mode="constant", value=i))
            voxels.append(voxel)
        return torch.cat(voxels, dim=0), torch.cat(voxel_coors, dim=0)

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, "
                f"point_cloud_range={self.point_cloud_range}, average_points={self.average_points})")
