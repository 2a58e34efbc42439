"""GPU parity: csrc/scatter.hip (DynamicScatter) vs the oracle.  The reference has no CPU binding and no
test for this op (voxelization.h:118,139) -> parity unpinned by fixtures; integers bit-exact, sums in
point order so fp32 features are bit-exact with the oracle too."""
import numpy as np
import pytest
import torch

import bevfusion_amd  # noqa: F401
import oracle
from bevfusion_amd import synthetic
from bevfusion_amd.ops import DynamicScatter, dynamic_scatter
from bevfusion_amd.ops.voxel import voxel_layer

pytestmark = pytest.mark.gpu
NUSC = synthetic.NUSC


def _case(n=30000, seed=3):
    pts = synthetic.uniform_points(n, seed=seed, rng_range=(-10, -10, -2, 10, 10, 2), margin=1.0)
    coors = oracle.dynamic_voxelize(pts, [0.5, 0.5, 0.5], [-10, -10, -2, 10, 10, 2])
    return pts, coors


@pytest.mark.parametrize("reduce_type", ["sum", "mean", "max"])
def test_forward_backward_vs_oracle(dev, reduce_type):
    pts, coors = _case()
    vf, vc, p2v, cnt = oracle.dynamic_scatter_fwd(pts, coors, reduce_type)
    feats = torch.from_numpy(pts).to(dev)
    got = voxel_layer.dynamic_point_to_voxel_forward(feats, torch.from_numpy(coors).to(dev), reduce_type)
    assert np.array_equal(got[1].cpu().numpy(), vc)            # sorted unique coordinates
    assert np.array_equal(got[2].cpu().numpy(), p2v)           # inverse map, -1 for dropped points
    assert np.array_equal(got[3].cpu().numpy(), cnt)
    assert (p2v == -1).sum() > 0
    assert np.array_equal(got[0].cpu().numpy(), vf)            # same summation order -> bit-exact
    g = np.random.default_rng(1).standard_normal(vf.shape).astype(np.float32)
    want = oracle.dynamic_scatter_bwd(g, pts, vf, p2v, cnt, reduce_type)
    grad = torch.zeros_like(feats)
    voxel_layer.dynamic_point_to_voxel_backward(grad, torch.from_numpy(g).to(dev), feats, got[0], got[2], got[3],
                                                reduce_type)
    assert np.array_equal(grad.cpu().numpy(), want)


def test_autograd_module_and_batched(dev):
    pts, coors = _case(8000, seed=5)
    feats = torch.from_numpy(pts).to(dev).requires_grad_(True)
    c = torch.from_numpy(coors).to(dev)
    vf, vc = dynamic_scatter(feats, c, "mean")
    vf.sum().backward()
    valid = (coors >= 0).all(1)
    assert torch.allclose(feats.grad[torch.from_numpy(valid).to(dev)].sum(0), torch.full((5,), float(vf.shape[0]), device=dev), rtol=1e-4)
    assert not feats.grad[torch.from_numpy(~valid).to(dev)].any()
    # batched coors (b, x, y, z): per-sample reduce, batch id put back
    ds = DynamicScatter([0.5, 0.5, 0.5], [-10, -10, -2, 10, 10, 2], average_points=False)
    b = torch.cat([torch.zeros(4000, 1, dtype=torch.int32), torch.ones(4000, 1, dtype=torch.int32)]).to(dev)
    feats2, coors2 = ds(feats.detach(), torch.cat([b, c], 1))
    assert coors2.shape[1] == 4 and set(coors2[:, 0].unique().tolist()) == {0, 1}
    v0, c0 = ds(feats.detach()[:4000], c[:4000])
    n0 = int((coors2[:, 0] == 0).sum())
    assert torch.equal(feats2[:n0], v0) and torch.equal(coors2[:n0, 1:], c0)


def test_edge_cases(dev):
    e = voxel_layer.dynamic_point_to_voxel_forward(torch.zeros(0, 4, device=dev), torch.zeros(0, 3, dtype=torch.int32, device=dev), "max")
    assert e[0].shape == (0, 4) and e[2].shape == (0,)
    allbad = voxel_layer.dynamic_point_to_voxel_forward(torch.ones(10, 4, device=dev), -torch.ones(10, 3, dtype=torch.int32, device=dev), "sum")
    assert allbad[0].shape == (0, 4) and (allbad[2] == -1).all()
    with pytest.raises(RuntimeError, match="do not support reduce type"):
        voxel_layer.dynamic_point_to_voxel_forward(torch.ones(2, 4, device=dev), torch.zeros(2, 3, dtype=torch.int32, device=dev), "min")


def test_voxel_mean(dev):
    """feats.sum(1) / sizes (BF/bevfusion.py:251-253) fused in one kernel."""
    from bevfusion_amd.bevfusion import voxel_mean
    pts = synthetic.lidar_sweep(20000, seed=9)
    vox, coors, num = oracle.hard_voxelize(pts, NUSC["voxel_size"], NUSC["point_cloud_range"], 10, 120000)
    want = oracle.voxel_mean(vox, num)
    got = voxel_mean(torch.from_numpy(vox).to(dev), torch.from_numpy(num).to(dev))
    assert np.array_equal(got.cpu().numpy(), want)
    ref = torch.from_numpy(vox).sum(1) / torch.from_numpy(num).float().view(-1, 1)
    assert np.allclose(want, ref.numpy(), rtol=1e-6, atol=1e-6)
