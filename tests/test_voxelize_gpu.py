"""GPU parity: csrc/voxelize.hip through the C ABI vs the reference-derived goldens and the oracle.
Bit-exact (indices, counts and the gathered point rows)."""
import numpy as np
import pytest
import torch

import bevfusion_amd  # noqa: F401
import oracle
from bevfusion_amd import synthetic
from bevfusion_amd.ops import Voxelization
from bevfusion_amd.ops.voxel import voxel_layer

from util import sha

pytestmark = pytest.mark.gpu
NUSC = synthetic.NUSC


def _hard(dev, pts, vs, rng, max_points, max_voxels):
    p = torch.from_numpy(pts).to(dev)
    voxels = p.new_zeros((max_voxels, max_points, p.shape[1]))
    coors = p.new_zeros((max_voxels, 3), dtype=torch.int)
    num = p.new_zeros((max_voxels,), dtype=torch.int)
    m = voxel_layer.hard_voxelize(p, voxels, coors, num, vs, rng, max_points, max_voxels, 3, True)
    # rows beyond voxel_num must stay zero (caller-zeroed contract)
    assert not voxels[m:].any() and not num[m:].any()
    return voxels[:m].cpu().numpy(), coors[:m].cpu().numpy(), num[:m].cpu().numpy()


def test_known_answer(dev, golden_vox):
    """The reference's known answer (test_voxel_generator.py:7-20) through the HIP path."""
    vox, coors, num = _hard(dev, golden_vox["kat_points"], [5, 5, 1], [0, 0, 0, 20, 40, 4], 5, 20)
    assert np.array_equal(coors[:, ::-1], np.array([[2, 0, 0], [3, 0, 0], [0, 0, 0], [1, 0, 0]]))
    assert np.array_equal(num, np.array([5, 5, 5, 3]))
    assert np.array_equal(vox, golden_vox["kat_voxels"])


def test_hard_vs_reference_goldens(dev, golden_vox):
    pts = synthetic.lidar_sweep(40000, seed=1000)
    rng = [-40.0, -40.0, -40.0, 40.0, 40.0, 40.0]
    vox, coors, num = _hard(dev, pts, [1.0, 1.0, 1.0], rng, 10, 20000)
    assert np.array_equal(coors, golden_vox["cubic_coors"])
    assert np.array_equal(num, golden_vox["cubic_num"])
    assert sha(vox) == str(golden_vox["cubic_voxels_sha"])
    vox, coors, num = _hard(dev, pts, [0.5, 0.5, 0.5], rng, 3, 3000)  # both caps binding
    assert np.array_equal(coors, golden_vox["cap_coors"])
    assert np.array_equal(num, golden_vox["cap_num"])
    assert sha(vox) == str(golden_vox["cap_voxels_sha"])
    upts = synthetic.uniform_points(20000, seed=7, rng_range=(-20, -20, -20, 20, 20, 20), margin=2.0)
    vox, coors, num = _hard(dev, upts, [0.5, 0.5, 0.5], [-20, -20, -20, 20, 20, 20], 10, 30000)
    assert np.array_equal(coors, golden_vox["uni_coors"])
    assert np.array_equal(num, golden_vox["uni_num"])
    assert sha(vox) == str(golden_vox["uni_voxels_sha"])


def test_dynamic_vs_reference_goldens(dev, golden_vox):
    for pts, key in ((synthetic.lidar_sweep(40000, seed=1000), "dyn_nusc_coors"),
                     (synthetic.uniform_points(40000, seed=11), "dyn_uni_coors")):
        p = torch.from_numpy(pts).to(dev)
        coors = p.new_zeros((p.shape[0], 3), dtype=torch.int)
        voxel_layer.dynamic_voxelize(p, coors, NUSC["voxel_size"], NUSC["point_cloud_range"], 3)
        assert np.array_equal(coors.cpu().numpy(), golden_vox[key])


@pytest.mark.parametrize("seed,n", [(1001, 40000), (1002, 34688), (1003, 250000)])
def test_hard_nuscenes_grid_vs_oracle(dev, seed, n):
    """The real nuScenes grid 1440x1440x40 (the reference's own CPU binary cannot run it)."""
    pts = synthetic.lidar_sweep(n, seed=seed)
    want = oracle.hard_voxelize(pts, NUSC["voxel_size"], NUSC["point_cloud_range"], 10, 120000)
    got = _hard(dev, pts, NUSC["voxel_size"], NUSC["point_cloud_range"], 10, 120000)
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)


def test_edge_cases(dev):
    vs, rng = [1.0, 1.0, 1.0], [0, 0, 0, 8, 8, 8]
    # empty input
    vox, coors, num = _hard(dev, np.zeros((0, 4), np.float32), vs, rng, 5, 10)
    assert vox.shape == (0, 5, 4)
    # all points in ONE voxel (rank cascade under maximal contention), max_points binding
    pts = np.zeros((5000, 4), np.float32)
    pts[:, :3] = 3.5
    pts[:, 3] = np.arange(5000)
    vox, coors, num = _hard(dev, pts, vs, rng, 7, 10)
    assert coors.tolist() == [[3, 3, 3]] and num.tolist() == [7]
    assert np.array_equal(vox[0, :, 3], np.arange(7, dtype=np.float32))
    # all points out of range / NaN / inf
    bad = np.full((100, 4), np.nan, np.float32)
    bad[::2] = np.inf
    bad[1::4] = -1e30
    vox, coors, num = _hard(dev, bad, vs, rng, 5, 10)
    assert vox.shape[0] == 0
    # boundary values: exactly on min (inside) and exactly on max (outside)
    edge = np.array([[0, 0, 0, 1], [8, 0, 0, 2], [7.9999995, 7.9999995, 7.9999995, 3], [-0.0, 0.0, 0.0, 4]], np.float32)
    want = oracle.hard_voxelize(edge, vs, rng, 5, 10)
    got = _hard(dev, edge, vs, rng, 5, 10)
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)
    # max_voxels smaller than the number of distinct voxels, random order
    r = np.random.default_rng(3)
    pts = r.uniform(0, 8, (4000, 4)).astype(np.float32)
    want = oracle.hard_voxelize(pts, vs, rng, 3, 100)
    got = _hard(dev, pts, vs, rng, 3, 100)
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)


def test_module_interface(dev):
    """Voxelization module: (train, test) max_voxels pair, dynamic mode with max_points=-1."""
    pts = torch.from_numpy(synthetic.lidar_sweep(20000, seed=5)).to(dev)
    vox = Voxelization(NUSC["voxel_size"], NUSC["point_cloud_range"], 10, (500, 120000))
    vox.train()
    v, c, n = vox(pts)
    assert v.shape[0] == 500 and c.shape == (500, 3) and n.shape == (500,)
    vox.eval()
    v2, c2, n2 = vox(pts)
    assert v2.shape[0] > 500 and torch.equal(c2[:500], c) and torch.equal(v2[:500], v)
    assert "Voxelization(voxel_size=" in repr(vox)
    dyn = Voxelization(NUSC["voxel_size"], NUSC["point_cloud_range"], -1, -1)
    coors = dyn(pts)
    assert coors.shape == (20000, 3) and coors.dtype == torch.int32
    want = oracle.dynamic_voxelize(pts.cpu().numpy(), NUSC["voxel_size"], NUSC["point_cloud_range"])
    assert np.array_equal(coors.cpu().numpy(), want)


def test_deterministic_across_runs(dev):
    pts = synthetic.lidar_sweep(40000, seed=77)
    a = _hard(dev, pts, NUSC["voxel_size"], NUSC["point_cloud_range"], 10, 120000)
    for _ in range(3):
        b = _hard(dev, pts, NUSC["voxel_size"], NUSC["point_cloud_range"], 10, 120000)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
