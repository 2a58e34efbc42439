"""CPU: the oracle (oracle/*.c) against the golden vectors generated from the reference
(tests/golden/make_golden.py) and against independent formulations."""
import numpy as np
import pytest
import torch

import bevfusion_amd  # noqa: F401
import oracle
from bevfusion_amd import synthetic

from util import make_bev_pool_case, rel_err, sha

NUSC = synthetic.NUSC


# ------------------------------------------------------------------ voxelization
def test_known_answer_voxel_generator(golden_vox):
    """The reference's only numeric known answer (tests/.../test_voxel_generator.py:7-20):
    expected coors there are (z,y,x); this fork's ops emit (x,y,z)."""
    np.random.seed(0)
    pts = np.random.uniform(0, 4, (20, 3)).astype(np.float32)
    assert np.array_equal(pts, golden_vox["kat_points"])
    vox, coors, num = oracle.hard_voxelize(pts, [5, 5, 1], [0, 0, 0, 20, 40, 4], 5, 20)
    assert np.array_equal(coors[:, ::-1], np.array([[2, 0, 0], [3, 0, 0], [0, 0, 0], [1, 0, 0]]))
    assert np.array_equal(num, np.array([5, 5, 5, 3]))
    assert vox.shape == (4, 5, 3)
    assert np.array_equal(vox, golden_vox["kat_voxels"])
    assert np.array_equal(coors, golden_vox["kat_coors"])


def test_hard_voxelize_vs_reference_cubic(golden_vox):
    pts = synthetic.lidar_sweep(40000, seed=1000)
    assert sha(pts) == str(golden_vox["cubic_in_sha"]), "synthetic input drifted from the golden's"
    rng = [-40.0, -40.0, -40.0, 40.0, 40.0, 40.0]
    vox, coors, num = oracle.hard_voxelize(pts, [1.0, 1.0, 1.0], rng, 10, 20000)
    assert np.array_equal(coors, golden_vox["cubic_coors"])
    assert np.array_equal(num, golden_vox["cubic_num"])
    assert sha(vox) == str(golden_vox["cubic_voxels_sha"])
    assert np.array_equal(vox[:64], golden_vox["cubic_voxels_head"])


def test_hard_voxelize_vs_reference_caps(golden_vox):
    """max_voxels and max_points both binding."""
    pts = synthetic.lidar_sweep(40000, seed=1000)
    vox, coors, num = oracle.hard_voxelize(pts, [0.5, 0.5, 0.5], [-40.0, -40.0, -40.0, 40.0, 40.0, 40.0], 3, 3000)
    assert coors.shape[0] == 3000
    assert np.array_equal(coors, golden_vox["cap_coors"])
    assert np.array_equal(num, golden_vox["cap_num"])
    assert sha(vox) == str(golden_vox["cap_voxels_sha"])


def test_hard_voxelize_vs_reference_uniform(golden_vox):
    pts = synthetic.uniform_points(20000, seed=7, rng_range=(-20, -20, -20, 20, 20, 20), margin=2.0)
    assert sha(pts) == str(golden_vox["uni_in_sha"])
    vox, coors, num = oracle.hard_voxelize(pts, [0.5, 0.5, 0.5], [-20, -20, -20, 20, 20, 20], 10, 30000)
    assert np.array_equal(coors, golden_vox["uni_coors"])
    assert np.array_equal(num, golden_vox["uni_num"])
    assert sha(vox) == str(golden_vox["uni_voxels_sha"])


def test_dynamic_voxelize_vs_reference_nuscenes_grid(golden_vox):
    pts = synthetic.lidar_sweep(40000, seed=1000)
    coors = oracle.dynamic_voxelize(pts, NUSC["voxel_size"], NUSC["point_cloud_range"])
    assert np.array_equal(coors, golden_vox["dyn_nusc_coors"])
    upts = synthetic.uniform_points(40000, seed=11)
    assert sha(upts) == str(golden_vox["dyn_uni_in_sha"])
    coors = oracle.dynamic_voxelize(upts, NUSC["voxel_size"], NUSC["point_cloud_range"])
    assert np.array_equal(coors, golden_vox["dyn_uni_coors"])
    assert (coors[:, 0] == -1).sum() > 100  # the case does contain out-of-range points


def test_grid_size_matches_reference_chain():
    assert list(oracle.grid_size(NUSC["voxel_size"], NUSC["point_cloud_range"])) == [1440, 1440, 40]


def test_hard_voxelize_consistent_with_dynamic_at_nuscenes_grid():
    """At the nuScenes grid the reference's hard CPU path segfaults (table shape bug); the
    restatement is cross-checked point by point against the (reference-pinned) dynamic coords."""
    pts = synthetic.lidar_sweep(40000, seed=1001)
    dc = oracle.dynamic_voxelize(pts, NUSC["voxel_size"], NUSC["point_cloud_range"])
    vox, coors, num = oracle.hard_voxelize(pts, NUSC["voxel_size"], NUSC["point_cloud_range"], 10, 120000)
    valid = dc[:, 0] >= 0
    keys = (dc[:, 0].astype(np.int64) * 1440 + dc[:, 1]) * 40 + dc[:, 2]
    uniq, first_idx, counts = np.unique(keys[valid], return_index=True, return_counts=True)
    order = np.argsort(first_idx)  # first-occurrence order
    assert coors.shape[0] == len(uniq)
    want_coors = dc[valid][first_idx[order]]
    assert np.array_equal(coors, want_coors)
    assert np.array_equal(num, np.minimum(counts[order], 10))
    # every stored row is the right point, in point order
    vpts = pts[valid]
    vkeys = keys[valid]
    for v in (0, 1, len(uniq) // 2, len(uniq) - 1):
        members = vpts[vkeys == uniq[order[v]]][:10]
        assert np.array_equal(vox[v, :len(members)], members)
        assert not vox[v, len(members):].any()


# ------------------------------------------------------------------ bev_pool
@pytest.mark.parametrize("integer_valued", [True, False])
def test_bev_pool_three_formulations_agree(integer_valued):
    n, c, b, d, h, w = 20000, 16, 2, 1, 24, 20
    x, geom, ranks = make_bev_pool_case(n, c, b, d, h, w, seed=3, integer_valued=integer_valued, long_tail=True)
    starts, lengths = oracle.intervals_from_ranks(ranks)
    out = oracle.bev_pool_fwd(x, geom, starts, lengths, b, d, h, w)
    # (2) fp64 QuickCumsum restated from bev_pool.py:7-34
    qc = oracle.quickcumsum_f64(x, ranks)
    g0 = geom[starts]
    got = out[g0[:, 3], g0[:, 2], g0[:, 0], g0[:, 1]]
    # (3) index_add in fp64
    ia = np.zeros((len(starts), c))
    np.add.at(ia, np.repeat(np.arange(len(starts)), lengths), x.astype(np.float64))
    assert np.allclose(qc, ia, rtol=0, atol=1e-9)
    if integer_valued:
        assert np.array_equal(got.astype(np.float64), qc)
    else:
        assert rel_err(got, qc) < 1e-6
    # untouched cells are zero
    mask = np.ones((b, d, h, w), bool)
    mask[g0[:, 3], g0[:, 2], g0[:, 0], g0[:, 1]] = False
    assert not out[mask].any()
    # backward = broadcast of the cell gradient to each member row
    og = np.random.default_rng(0).standard_normal(out.shape).astype(np.float32)
    xg = oracle.bev_pool_bwd(og, geom, starts, lengths, n)
    assert np.array_equal(xg, og[geom[:, 3], geom[:, 2], geom[:, 0], geom[:, 1]])


def test_bev_pool_matches_torch_autograd():
    n, c, b, d, h, w = 5000, 8, 1, 2, 10, 12
    x, geom, ranks = make_bev_pool_case(n, c, b, d, h, w, seed=9)
    starts, lengths = oracle.intervals_from_ranks(ranks)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    flat = torch.from_numpy((((geom[:, 3].astype(np.int64) * d + geom[:, 2]) * h + geom[:, 0]) * w + geom[:, 1]))
    out_t = torch.zeros(b * d * h * w, c, dtype=torch.float64).index_add(0, flat, xt)
    og = torch.randn(b, d, h, w, c, dtype=torch.float64)
    (out_t.view(b, d, h, w, c) * og).sum().backward()
    out = oracle.bev_pool_fwd(x, geom, starts, lengths, b, d, h, w)
    assert rel_err(out, out_t.detach().view(b, d, h, w, c).numpy()) < 1e-6
    xg = oracle.bev_pool_bwd(og.float().numpy(), geom, starts, lengths, n)
    assert rel_err(xg, xt.grad.numpy()) < 1e-6


# ------------------------------------------------------------------ LSS geometry (reference-pinned)
def _geometry_from_rig(rig, frustum):
    K = rig["camera_intrinsics"][..., :3, :3]
    aug = rig["img_aug_matrix"]
    post_rots_inv = torch.inverse(torch.from_numpy(aug[..., :3, :3])).numpy()
    post_trans = aug[..., :3, 3]
    combine = torch.from_numpy(rig["camera2lidar"][..., :3, :3]).matmul(torch.inverse(torch.from_numpy(K))).numpy()
    c2l_t = rig["camera2lidar"][..., :3, 3]
    la = rig["lidar_aug_matrix"]
    return oracle.frustum_geometry(frustum, post_trans, post_rots_inv, combine, c2l_t, la[..., :3, :3], la[..., :3, 3])


def test_geometry_vs_reference_tiny(golden_lss):
    rig = {k[len("tiny_rig_"):]: golden_lss[k] for k in golden_lss.files if k.startswith("tiny_rig_")}
    frustum = synthetic.create_frustum((64, 176), (8, 22), [1.0, 61.0, 3.0]).numpy()
    assert np.array_equal(frustum, golden_lss["tiny_frustum"])
    geom = _geometry_from_rig(rig, frustum)
    ref = golden_lss["tiny_geom"]
    assert geom.shape == ref.shape
    # fixed-order fp32 evaluation vs torch's BLAS evaluation: ~1 ulp of the largest magnitude (~100 m)
    assert np.abs(geom - ref).max() < 5e-5


def test_bev_cells_bit_exact_vs_reference_tiny(golden_lss):
    """Cells / kept / ranks computed by the oracle FROM THE REFERENCE'S geometry equal the
    reference's bev_pool_aux outputs exactly (truncation, range mask, rank formula)."""
    ref_geom = golden_lss["tiny_geom"]
    dx, bx, nx = golden_lss["tiny_dx"], golden_lss["tiny_bx"], golden_lss["tiny_nx"]
    origin = (torch.from_numpy(bx) - torch.from_numpy(dx) / 2.0).numpy()
    gf, kept, ranks, idx = oracle.bev_pool_aux(ref_geom, ref_geom.shape[0], origin, dx, nx)
    assert np.array_equal(kept, golden_lss["tiny_kept"])
    assert np.array_equal(ranks, golden_lss["tiny_ranks"])
    assert np.array_equal(gf, golden_lss["tiny_geom_feats"])


def test_view_transform_glue_vs_reference_tiny(golden_lss):
    """BaseViewTransform.bev_pool (reshape, kept, sort gather, op, permute, unbind/cat) restated
    with the oracle equals the reference's python glue output."""
    ref_geom = golden_lss["tiny_geom"]
    B, N, D, fH, fW, _ = ref_geom.shape
    C = 8
    g = torch.Generator().manual_seed(123)
    x = torch.randn(B, N, D, fH, fW, C, generator=g).numpy()
    assert sha(x) == str(golden_lss["tiny_x_sha"])
    dx, bx, nx = golden_lss["tiny_dx"], golden_lss["tiny_bx"], golden_lss["tiny_nx"]
    origin = (torch.from_numpy(bx) - torch.from_numpy(dx) / 2.0).numpy()
    gf, kept, ranks, idx = oracle.bev_pool_aux(ref_geom, B, origin, dx, nx)
    xs = x.reshape(-1, C)[kept][idx]
    starts, lengths = oracle.intervals_from_ranks(ranks)
    out = oracle.bev_pool_fwd(xs, gf, starts, lengths, B, int(nx[2]), int(nx[0]), int(nx[1]))
    bev = np.concatenate(list(np.transpose(out, (0, 4, 1, 2, 3)).transpose(2, 0, 1, 3, 4)), 1)
    ref = golden_lss["tiny_bev"]
    assert bev.shape == ref.shape
    # the reference's argsort is unstable -> different within-interval order -> fp32 rounding only
    assert rel_err(bev, ref) < 1e-6


def test_geometry_full_size_vs_reference(golden_lss):
    """Full nuScenes-size frustum (1 993 728 points): counts identical to the reference's, cells
    differ only where a coordinate sits within float rounding of a cell boundary."""
    rig = synthetic.camera_rig(batch=1)
    frustum = synthetic.create_frustum().numpy()
    assert sha(frustum) == str(golden_lss["full_frustum_sha"])
    geom = _geometry_from_rig(rig, frustum)
    sample = geom.reshape(-1, 3)[::997]
    assert np.abs(sample - golden_lss["full_geom_sample"]).max() < 5e-5
    dx, bx, nx = golden_lss["full_dx"], golden_lss["full_bx"], golden_lss["full_nx"]
    origin = (torch.from_numpy(bx) - torch.from_numpy(dx) / 2.0).numpy()
    gf, kept, ranks, idx = oracle.bev_pool_aux(geom, 1, origin, dx, nx)
    starts, lengths = oracle.intervals_from_ranks(ranks)
    nprime, nkept, m, maxlen, medlen = [int(v) for v in golden_lss["full_counts"]]
    assert kept.size == nprime == 6 * 118 * 32 * 88
    assert abs(int(kept.sum()) - nkept) <= 20          # boundary-rounding flips only
    assert abs(len(starts) - m) <= 20
    assert abs(int(lengths.max()) - maxlen) <= 4


# ------------------------------------------------------------------ fused lift-splat restatement
def test_lift_splat_equals_materialised_path():
    """oracle_lift_splat_fwd == outer product (depth_lss.py:723-725) + gathers + bev_pool, bit for bit."""
    rng = np.random.default_rng(5)
    BN, D, fH, fW, C = 4, 6, 5, 7, 8
    B, N = 2, 2
    depth = rng.random((BN, D, fH, fW)).astype(np.float32)
    feat = rng.standard_normal((BN, C, fH, fW)).astype(np.float32)
    x = (depth[:, None] * feat[:, :, None])  # [BN, C, D, fH, fW]
    x = x.reshape(B, N, C, D, fH, fW).transpose(0, 1, 3, 4, 5, 2).reshape(-1, C)
    nprime = x.shape[0]
    h, w, d = 9, 8, 1
    cells = rng.integers(-1, h * w, nprime)
    keptm = cells >= 0
    bidx = np.arange(nprime) // (nprime // B)
    gx, gy = cells // w, cells % w
    rank = (gx * (w * d * B) + gy * (d * B) + bidx).astype(np.int64)
    src_all = np.arange(nprime)[keptm]
    order = np.argsort(rank[keptm], kind="stable")
    src = src_all[order].astype(np.int32)
    geom = np.stack([gx, gy, np.zeros_like(gx), bidx], 1)[keptm][order].astype(np.int32)
    ranks = rank[keptm][order]
    starts, lengths = oracle.intervals_from_ranks(ranks)
    want = oracle.bev_pool_fwd(x[src], geom, starts, lengths, B, d, h, w)
    got = oracle.lift_splat_fwd(depth, feat, src, geom, starts, lengths, B, d, h, w)
    assert np.array_equal(got, want)
    # backward against torch autograd of the materialised formulation (fp64)
    dt = torch.tensor(depth, dtype=torch.float64, requires_grad=True)
    ft = torch.tensor(feat, dtype=torch.float64, requires_grad=True)
    xt = (dt[:, None] * ft[:, :, None]).reshape(B, N, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2).reshape(-1, C)
    flat = torch.from_numpy((((geom[:, 3].astype(np.int64) * d + geom[:, 2]) * h + geom[:, 0]) * w + geom[:, 1]))
    out_t = torch.zeros(B * d * h * w, C, dtype=torch.float64).index_add(0, flat, xt[torch.from_numpy(src.astype(np.int64))])
    og = torch.randn(B, d, h, w, C, dtype=torch.float64)
    (out_t.view(B, d, h, w, C) * og).sum().backward()
    dd, df = oracle.lift_splat_bwd(og.float().numpy(), depth, feat, src, geom, starts, lengths)
    assert rel_err(dd, dt.grad.numpy()) < 1e-5
    assert rel_err(df, ft.grad.numpy()) < 1e-5


# ------------------------------------------------------------------ sparse depth rasteriser + GT histogram
def _raster_inputs(golden_lss):
    rig = {k[len("rast_rig_"):]: golden_lss[k] for k in golden_lss.files if k.startswith("rast_rig_")}
    pts = [synthetic.lidar_sweep(6000, seed=300 + i)[:, :5].copy() for i in range(2)]
    assert sha(np.stack(pts)) == str(golden_lss["rast_points_sha"])
    return rig, pts


def test_rasteriser_vs_reference(golden_lss):
    """The restated projection (fixed fp32 op order) against the depth images the reference's own loop produced
    (BF/depth_lss.py:372-449, run on the CPU where scatter_ keeps the last duplicate): identical up to points whose
    projection lands within float rounding of a pixel edge."""
    rig, pts = _raster_inputs(golden_lss)
    ref = golden_lss["rast_depth"]
    inv = np.linalg.inv(rig["lidar_aug_matrix"].astype(np.float64)).astype(np.float32)
    inv = torch.inverse(torch.from_numpy(rig["lidar_aug_matrix"])).numpy()
    diff = 0
    for b in range(2):
        got = oracle.rasterise_depth(pts[b], inv[b, :3, :3], rig["lidar_aug_matrix"][b, :3, 3], rig["lidar2image"][b],
                                     rig["img_aug_matrix"][b], 64, 176)
        r = ref[b, :, 0]
        assert got.shape == r.shape
        diff += int((np.abs(got - r) > 1e-4 * np.maximum(np.abs(r), 1.0)).sum())
    assert (ref > 0).sum() > 5000
    assert diff <= 12, diff  # a handful of edge pixels out of ~6000 hits


def test_depth_histogram_bit_exact_vs_reference(golden_lss):
    """Histogram / distribution computed by the oracle FROM THE REFERENCE'S depth images equal the reference's
    get_cam_feats outputs exactly (clamp, +0.5*step, truncation, bin 0 cleared, normalisation)."""
    ref = golden_lss["rast_depth"]
    counts, distr = oracle.depth_histogram(ref.reshape(12, 64, 176), 8, 22, 20, [1.0, 61.0, 3.0])
    assert np.array_equal(counts.reshape(2, 6, 8, 22, 20), golden_lss["rast_counts"])
    assert np.array_equal(distr.reshape(2, 6, 8, 22, 20), golden_lss["rast_gt_distr"])


# ------------------------------------------------------------------ bev_pool vs the reference's own QuickCumsum
def _bev_case(golden_bev, name):
    seed, n, C, B, D, H, W, integer = [int(v) for v in golden_bev[name + "_cfg"]]
    x, geom, ranks = synthetic.bev_pool_case(seed, n, C, B, D, H, W, bool(integer))
    assert sha(x) == str(golden_bev[name + "_x_sha"]) and sha(geom) == str(golden_bev[name + "_geom_sha"])
    assert sha(ranks) == str(golden_bev[name + "_ranks_sha"]), "synthetic input drifted from the golden's"
    return x, geom, ranks, (B, D, H, W, C)


@pytest.mark.parametrize("name", ["int", "flt", "odd"])
def test_bev_pool_vs_reference_quickcumsum(golden_bev, name):
    """The oracle's bev_pool sum, interval tables and backward against the outputs of the REFERENCE's QuickCumsum
    (BF/ops/bev_pool/bev_pool.py:7-34, run in fp64) and of its interval construction (:48-54).  Integer-valued inputs:
    bit-exact; normal inputs: <= 1e-6 of the row scale (fp32 serial sum vs the exact fp64 sum)."""
    x, geom, ranks, (B, D, H, W, C) = _bev_case(golden_bev, name)
    starts, lengths = oracle.intervals_from_ranks(ranks)
    assert np.array_equal(starts, golden_bev[name + "_starts"]) and np.array_equal(lengths, golden_bev[name + "_lengths"])
    out = oracle.bev_pool_fwd(x, geom, starts, lengths, B, D, H, W)
    rg = golden_bev[name + "_row_geom"]
    assert np.array_equal(rg, geom[starts + lengths - 1])          # QuickCumsum keeps the LAST row of each rank
    rows = out[rg[:, 3], rg[:, 2], rg[:, 0], rg[:, 1]]
    want = golden_bev[name + "_rows"]
    if name == "int":
        assert np.array_equal(rows, want) and sha(out) == str(golden_bev[name + "_dense_sha_f32"])
    else:
        scale = np.abs(want).max()
        assert np.abs(rows - want).max() <= 1e-6 * scale * np.sqrt(lengths.max())
    assert np.count_nonzero(out) <= rows.size
    # backward: every member row of an interval receives its cell's gradient (QuickCumsum.backward: gradx[back])
    og = np.zeros((B, D, H, W, C), np.float32)
    og[rg[:, 3], rg[:, 2], rg[:, 0], rg[:, 1]] = golden_bev[name + "_grad_rows"]
    xg = oracle.bev_pool_bwd(og, geom, starts, lengths, x.shape[0])
    assert sha(xg) == str(golden_bev[name + "_xgrad_sha_f32"])
    assert np.array_equal(xg[::211], golden_bev[name + "_xgrad_sample"])
