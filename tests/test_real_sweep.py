"""The one real LiDAR sweep the reference ships (demo/data/nuscenes/*LIDAR_TOP*.pcd.bin, committed as data in
tests/golden/real_sweep.npz together with what the reference's compiled dynamic_voxelize makes of it -- make_golden.py (E)).

Two pins that do not come from this repo's own oracle:
  * voxel coordinates: the REFERENCE's CPU extension on the nuScenes grid (32 330 points inside, 17 509 voxels);
  * rulebook sizes: SURVEY.md Appendix C / section 8 a-7, measured by the survey on the same sweep with a 41-cell z axis
    (z range [-5, 3.2): 17 675 voxels -> 29 672 -> 21 725 -> 11 236 -> 9 256 active sites, SubM pairs
    55 723 / 285 182 / 268 655 / 154 548, strided pairs 58 690 / 98 910 / 71 642 / 15 183).  The reference's voxelizer itself
    has 40 z cells (round(8 / 0.2), voxelization_cpu.cpp:121-124), which gives 17 509 voxels; both grids are tested.
"""
import os

import numpy as np
import pytest

import bevfusion_amd  # noqa: F401
import oracle
from bevfusion_amd import synthetic

from util import sha

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "real_sweep.npz")
N = synthetic.NUSC
RANGE41 = list(N["point_cloud_range"][:5]) + [3.2]           # 41 z cells, the survey's grid
# (N_in, SubM pairs, strided pairs, N_out) per stage
APPENDIX_C = [(17675, 55723, 58690, 29672), (29672, 285182, 98910, 21725), (21725, 268655, 71642, 11236),
              (11236, 154548, 15183, 9256)]
# the same chain from the reference's own 40-cell grid (oracle; cross-checked against the HIP path below)
REF_GRID = [(17509, 55517, 58336, 29374), (29374, 282806, 98238, 21571), (21571, 267243, 71304, 11174),
            (11174, 153870, 15121, 9204)]
STRIDED = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 0)),
           ((1, 1, 3), (1, 1, 2), (0, 0, 0))]   # the encoder's four strided layers (bevfusion.nuscenes_config)


@pytest.fixture(scope="module")
def sweep():
    return np.load(GOLDEN)


def _oracle_chain(coors):
    idx = np.concatenate([np.zeros((len(coors), 1), np.int32), coors.astype(np.int32)], 1)
    shape, out = [1440, 1440, 41], []
    for k, s, p in STRIDED:
        subm = int((oracle.rulebook_subm(idx, shape, 3) >= 0).sum())
        oi, pf, _, osz = oracle.rulebook_sparse(idx, shape, k, s, p)
        out.append((len(idx), subm, int((pf >= 0).sum()), len(oi)))
        idx, shape = oi, list(osz)
    return out


def test_oracle_voxelization_equals_reference_on_the_real_sweep(sweep):
    pts = sweep["points"]
    assert pts.shape == (34688, 5)
    dyn = oracle.dynamic_voxelize(pts, N["voxel_size"], N["point_cloud_range"])
    assert sha(dyn) == str(sweep["dyn_coors_sha"])           # per-point cells == the reference's compiled dynamic_voxelize
    assert int((dyn[:, 0] >= 0).sum()) == int(sweep["n_inside"]) == 32330
    v, c, n = oracle.hard_voxelize(pts, N["voxel_size"], N["point_cloud_range"], 10, 120000)
    assert len(c) == int(sweep["n_voxels"]) == 17509
    assert sha(c) == str(sweep["hard_coors_sha"]) and sha(n) == str(sweep["hard_num_sha"])


def test_oracle_rulebooks_reproduce_survey_appendix_c(sweep):
    """The restated rulebook semantics (oracle/spconv_oracle.c, parity otherwise unpinned: spconv is not in the tree) give the
    active-site and pair counts the SURVEY measured independently on this sweep."""
    _, c41, _ = oracle.hard_voxelize(sweep["points"], N["voxel_size"], RANGE41, 10, 120000)
    assert _oracle_chain(c41) == APPENDIX_C
    _, c40, _ = oracle.hard_voxelize(sweep["points"], N["voxel_size"], N["point_cloud_range"], 10, 120000)
    assert _oracle_chain(c40) == REF_GRID


@pytest.mark.gpu
def test_hip_voxelization_of_the_real_sweep(dev, sweep):
    import torch
    from bevfusion_amd.ops import Voxelization
    pts = torch.from_numpy(sweep["points"]).to(dev)
    vox = Voxelization(N["voxel_size"], N["point_cloud_range"], 10, 120000).to(dev)
    v, c, n = vox(pts)
    assert c.shape[0] == 17509
    assert sha(c.cpu().numpy()) == str(sweep["hard_coors_sha"]) and sha(n.cpu().numpy()) == str(sweep["hard_num_sha"])
    np.testing.assert_array_equal(c[:64].cpu().numpy(), sweep["hard_coors_head"])
    dyn = Voxelization(N["voxel_size"], N["point_cloud_range"], -1, -1).to(dev)(pts)
    assert sha(dyn.cpu().numpy()) == str(sweep["dyn_coors_sha"])
    wv, _, _ = oracle.hard_voxelize(sweep["points"], N["voxel_size"], N["point_cloud_range"], 10, 120000)
    np.testing.assert_array_equal(v.cpu().numpy(), wv)


@pytest.mark.gpu
@pytest.mark.parametrize("grid", ["survey41", "reference40"])
def test_hip_encoder_rulebooks_on_the_real_sweep(dev, sweep, grid):
    """The HIP encoder on the real sweep: actives and pair counts of every stage == SURVEY Appendix C (41-cell grid) and the
    oracle's chain (reference grid); rulebooks bit-exact vs the oracle."""
    import torch
    from bevfusion_amd import spconv as sp
    from bevfusion_amd.bevfusion import nuscenes_config
    from bevfusion_amd.ops import Voxelization
    from bevfusion_amd.registry import MODELS
    rng, want = (RANGE41, APPENDIX_C) if grid == "survey41" else (list(N["point_cloud_range"]), REF_GRID)
    _, c, n = Voxelization(N["voxel_size"], rng, 10, 120000).to(dev)(torch.from_numpy(sweep["points"]).to(dev))
    assert c.shape[0] == want[0][0]
    coors = torch.cat([torch.zeros_like(c[:, :1]), c], 1)
    torch.manual_seed(0)
    enc = MODELS.build(nuscenes_config(camera=False)["pts_middle_encoder"]).to(dev).eval()
    seen = []
    orig = sp._SparseConvFunction.forward

    def spy(ctx, features, weight, data, n_in):
        seen.append((data.is_subm, n_in, int((data.pair_fwd >= 0).sum()), data.pair_fwd.shape[1], data))
        return orig(ctx, features, weight, data, n_in)

    sp._SparseConvFunction.forward = staticmethod(spy)
    try:
        with torch.no_grad():
            out = enc(torch.rand(c.shape[0], 5, device=dev), coors, 1)
    finally:
        sp._SparseConvFunction.forward = staticmethod(orig)
    assert out.shape == (1, 256, 180, 180) and torch.isfinite(out).all()
    assert len(seen) == 21                                   # 17 SubM + 4 strided
    strided = [s for s in seen if not s[0]]
    subm = [s for s in seen if s[0]]
    for stage, (n_in, p_subm, p_str, n_out) in enumerate(want):
        st = strided[stage]
        assert (st[1], st[2], st[3]) == (n_in, p_str, n_out), (stage, st[:4])
        mine = [s for s in subm if s[1] == n_in]
        assert mine and all(s[2] == p_subm for s in mine), (stage, [s[:4] for s in mine])
    # bit-exact rulebook of the first strided layer vs the oracle
    idx = coors.cpu().numpy().astype(np.int32)
    oi, pf, pb, _ = oracle.rulebook_sparse(idx, [1440, 1440, 41], *STRIDED[0])
    d0 = strided[0][4]
    np.testing.assert_array_equal(d0.out_indices.cpu().numpy(), oi)
    np.testing.assert_array_equal(d0.pair_fwd.cpu().numpy(), pf)
    np.testing.assert_array_equal(d0.pair_bwd.cpu().numpy(), pb)
