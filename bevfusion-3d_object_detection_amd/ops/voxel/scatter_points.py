"""Host mirror of BF/ops/voxel/scatter_points.py: the `DynamicScatter` module and the `dynamic_scatter`
autograd function (max / mean / sum reduction of point features into their voxels)."""
import torch
from torch import nn
from torch.autograd import Function

from . import voxel_layer


class _dynamic_scatter(Function):
    """feats [N,C] + coors [N,3] -> (voxel_feats [M,C], voxel_coors [M,3]); voxels come out in ascending
    (c0,c1,c2) order; rows with a negative coordinate are dropped (reference: scatter_points.py:8-49)."""

    @staticmethod
    def forward(ctx, feats, coors, reduce_type="max"):
        reduced, out_coors, point2voxel, counts = voxel_layer.dynamic_point_to_voxel_forward(feats, coors, reduce_type)
        ctx.reduce_type = reduce_type
        ctx.save_for_backward(feats, reduced, point2voxel, counts)
        ctx.mark_non_differentiable(out_coors)
        return reduced, out_coors

    @staticmethod
    def backward(ctx, grad_reduced, grad_coors=None):
        feats, reduced, point2voxel, counts = ctx.saved_tensors
        grad_feats = torch.zeros_like(feats)
        voxel_layer.dynamic_point_to_voxel_backward(grad_feats, grad_reduced.contiguous(), feats, reduced, point2voxel,
                                                    counts, ctx.reduce_type)
        return grad_feats, None, None


dynamic_scatter = _dynamic_scatter.apply


class DynamicScatter(nn.Module):
    """Constructor arguments as the reference (scatter_points.py:54-76): `average_points` selects mean
    instead of max.  Coordinates with a leading batch column are reduced sample by sample."""

    def __init__(self, voxel_size, point_cloud_range, average_points: bool):
        super().__init__()
        self.voxel_size, self.point_cloud_range, self.average_points = voxel_size, point_cloud_range, average_points

    @property
    def _reduce(self):
        return "mean" if self.average_points else "max"

    def forward_single(self, points, coors):
        return dynamic_scatter(points.contiguous(), coors.contiguous(), self._reduce)

    def forward(self, points, coors):
        if coors.size(-1) == 3:
            return self.forward_single(points, coors)
        per_sample = []
        for b in range(int(coors[-1, 0]) + 1):  # batch ids are ascending; the last row holds the largest
            sel = torch.nonzero(coors[:, 0] == b, as_tuple=False).flatten()
            feats_b, coors_b = self.forward_single(points[sel], coors[sel, 1:])
            per_sample.append((feats_b, nn.functional.pad(coors_b, (1, 0), mode="constant", value=b)))
        return torch.cat([f for f, _ in per_sample], dim=0), torch.cat([c for _, c in per_sample], dim=0)

    def __repr__(self):
        return "%s(voxel_size=%s, point_cloud_range=%s, average_points=%s)" % (
            type(self).__name__, self.voxel_size, self.point_cloud_range, self.average_points)
