"""BASELINE configs[0] ("CPU-only forward through the reference path, plumbing"): oracle/cpu_pipeline.py.

CPU tests pin the pipeline's own arithmetic to the C oracle (which the golden vectors pin to the reference); the GPU tests
compare the product model on the MI355X with the CPU pipeline on the same weights and inputs -- branch outputs at the
north-star tolerance, and the whole forward of one frame."""
import copy

import numpy as np
import pytest
import torch

import bevfusion_amd  # noqa: F401
import oracle
from bevfusion_amd import synthetic
from bevfusion_amd.bevfusion import nuscenes_config
from bevfusion_amd.registry import MODELS
from oracle import cpu_pipeline as cp

from test_spconv_oracle import random_sparse
from util import rel_err

N = synthetic.NUSC
MAT_KEYS = (("lidar2image", "lidar2img"), ("camera_intrinsics", "cam2img"), ("camera2lidar", "cam2lidar"),
            ("img_aug_matrix", "img_aug_matrix"), ("lidar_aug_matrix", "lidar_aug_matrix"))


def test_quickcumsum_restatement_vs_oracle_and_golden(golden_bev):
    """The torch restatement of QuickCumsum used by the CPU baseline equals the reference's own QuickCumsum outputs."""
    seed, n, C, B, D, H, W, integer = [int(v) for v in golden_bev["flt_cfg"]]
    x, geom, ranks = synthetic.bev_pool_case(seed, n, C, B, D, H, W, bool(integer))
    xt = torch.from_numpy(x).double().requires_grad_(True)
    rows, g = cp.QuickCumsum.apply(xt, torch.from_numpy(geom), torch.from_numpy(ranks))
    assert np.array_equal(g.numpy(), golden_bev["flt_row_geom"])
    assert np.abs(rows.detach().numpy() - golden_bev["flt_rows"]).max() < 1e-12
    rows.backward(torch.from_numpy(golden_bev["flt_grad_rows"]).double())
    assert np.array_equal(xt.grad.numpy().astype(np.float32)[::211], golden_bev["flt_xgrad_sample"])
    dense = cp.bev_pool_quickcumsum(torch.from_numpy(x).double(), torch.from_numpy(geom), torch.from_numpy(ranks), B, D, H, W)
    starts, lengths = oracle.intervals_from_ranks(ranks)
    want = oracle.bev_pool_fwd(x, geom, starts, lengths, B, D, H, W)
    assert rel_err(dense.permute(0, 2, 3, 4, 1).numpy(), want) < 1e-6


def test_sparse_conv_mm_vs_oracle():
    idx, feats = random_sparse(2, (20, 18, 9), 1500, 16, seed=3)
    w = np.random.default_rng(1).standard_normal((32, 3, 3, 3, 16)).astype(np.float32)
    pf = oracle.rulebook_subm(idx, (20, 18, 9), 3)
    got = cp.sparse_conv_mm(torch.from_numpy(feats), torch.from_numpy(w), pf, len(idx)).numpy()
    assert rel_err(got, oracle.spconv_fwd(feats, w, pf)) < 1e-5
    oi, pf, pb, osz = oracle.rulebook_sparse(idx, (20, 18, 9), 3, 2, 1)
    got = cp.sparse_conv_mm(torch.from_numpy(feats), torch.from_numpy(w), pf, len(oi)).numpy()
    assert rel_err(got, oracle.spconv_fwd(feats, w, pf)) < 1e-5


def _frame(B, n_points=20000):
    pts = [synthetic.lidar_sweep(n_points, seed=1000 + i) for i in range(B)]
    rig = synthetic.camera_rig(batch=B, seed=1, train_aug=True)
    mats = {dst: rig[src] for src, dst in MAT_KEYS}
    imgs = torch.from_numpy(np.random.default_rng(5).standard_normal((B, 6, 3, 256, 704)).astype(np.float32))
    return pts, mats, imgs


@pytest.mark.gpu
def test_lidar_branch_gpu_vs_cpu_pipeline(dev):
    """hard voxelize + mean + 21-layer sparse encoder + dense(): HIP path (fp32) vs the CPU pipeline, 1e-3 rel."""
    torch.manual_seed(0)
    model = MODELS.build(nuscenes_config(camera=False, lidar=True)).train()
    pts, _, _ = _frame(2)
    with torch.no_grad():
        vf, coords = cp.voxelize_mean(pts, N)
        want = cp.sparse_encoder_forward(model.pts_middle_encoder, vf, coords, 2)
        model = model.to(dev)
        got = model.extract_pts_feat({"points": [torch.from_numpy(p).to(dev) for p in pts]})
    assert got.shape == want.shape == (2, 256, 180, 180)
    assert rel_err(got.cpu().numpy(), want.numpy()) < 1e-3


@pytest.mark.gpu
def test_view_transform_gpu_vs_cpu_pipeline(dev):
    """rasteriser + dtransform/depthnet + softmax + outer product + geometry/ranks + bev_pool + downsample: the fused HIP
    path (fp32 island, as the reference) vs the CPU pipeline (QuickCumsum in fp64), 1e-3 rel."""
    torch.manual_seed(0)
    model = MODELS.build(nuscenes_config()).train()
    vt = model.view_transform
    pts, mats, _ = _frame(1)
    feats = torch.from_numpy(np.random.default_rng(7).standard_normal((1, 6, 256, 32, 88)).astype(np.float32))
    with torch.no_grad():
        want = cp.view_transform_forward(copy.deepcopy(vt), feats, pts, mats, exact=True)
        vt = vt.to(dev)
        t = {k: torch.from_numpy(v).to(dev) for k, v in mats.items()}
        got, _ = vt(feats.to(dev), [torch.from_numpy(p).to(dev) for p in pts], t["lidar2img"], t["cam2img"], t["cam2lidar"],
                    t["img_aug_matrix"], t["lidar_aug_matrix"])
    assert got.shape == want.shape == (1, 80, 180, 180)
    assert rel_err(got.float().cpu().numpy(), want.numpy()) < 1e-3


@pytest.mark.gpu
def test_config0_full_forward_gpu_vs_cpu_pipeline(dev):
    """BASELINE configs[0]: one synthetic nuScenes sample (40 k points, 6 x 256 x 704) through the whole forward on the
    CPU (reference formulation) and through the product on the MI355X in fp32: fused BEV features and dense heat-map
    logits agree (relative L2; the proposal top-k downstream of the heat-map is discontinuous and not compared)."""
    torch.manual_seed(0)
    model = MODELS.build(nuscenes_config()).train()
    pts, mats, imgs = _frame(1, n_points=40000)
    with torch.no_grad():
        outs_c, x_c, _ = cp.model_forward(copy.deepcopy(model), pts, imgs, mats, N, exact=True)
        model = model.to(dev)
        inp = {"points": [torch.from_numpy(p).to(dev) for p in pts], "imgs": imgs.to(dev)}
        inp.update({k: torch.from_numpy(v).to(dev) for k, v in mats.items()})
        x_g, _ = model.extract_feat(inp)
        outs_g = model.bbox_head(x_g)
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())  # noqa: E731
    assert x_g[0].shape == x_c[0].shape == (1, 512, 180, 180)
    assert l2(x_g[0].float().cpu(), x_c[0]) < 2e-3
    assert l2(outs_g[0][0]["dense_heatmap"].float().cpu(), outs_c[0][0]["dense_heatmap"]) < 2e-3


@pytest.mark.gpu
def test_view_transform_bf16_conv_stacks_vs_fp32_cpu_pipeline(dev):
    """The benchmark's default runs the view transform's dense conv stacks (dtransform / depthnet / downsample) in bf16
    (`DepthLSSTransform.conv_dtype`) where the reference keeps the whole view transform in an fp32 island
    (BF/bevfusion.py:177); geometry, ranks, softmax inputs' index paths and the pooling's accumulation stay fp32.
    Against the fp32 CPU pipeline the deviation is bounded by the north star's bf16 tolerance: 1e-2 relative."""
    torch.manual_seed(0)
    model = MODELS.build(nuscenes_config()).train()
    vt = model.view_transform
    pts, mats, _ = _frame(1)
    feats = torch.from_numpy(np.random.default_rng(7).standard_normal((1, 6, 256, 32, 88)).astype(np.float32))
    with torch.no_grad():
        want = cp.view_transform_forward(copy.deepcopy(vt), feats, pts, mats, exact=True)
        vt = vt.to(dev)
        vt.conv_dtype = torch.bfloat16
        t = {k: torch.from_numpy(v).to(dev) for k, v in mats.items()}
        got, _ = vt(feats.to(dev).to(torch.bfloat16), [torch.from_numpy(p).to(dev) for p in pts], t["lidar2img"], t["cam2img"],
                    t["cam2lidar"], t["img_aug_matrix"], t["lidar_aug_matrix"])
    assert got.shape == want.shape == (1, 80, 180, 180)
    err = rel_err(got.float().cpu().numpy(), want.numpy())
    l2 = float((got.float().cpu().double() - want.double()).norm() / want.double().norm())
    assert err < 1e-2 and l2 < 1e-2, (err, l2)
