"""GPU parity: csrc/lift_splat.hip (bev plan + fused lift-splat) vs the oracle and the reference goldens."""
import numpy as np
import pytest
import torch

import bevfusion_amd  # noqa: F401
import oracle
from bevfusion_amd import synthetic
from bevfusion_amd.depth_lss import LSSTransform, DepthLSSTransform, lift_splat

from util import rel_err, sha

pytestmark = pytest.mark.gpu
NUSC = synthetic.NUSC

TINY = dict(in_channels=16, out_channels=8, image_size=(64, 176), feature_size=(8, 22),
            xbound=[-54.0, 54.0, 1.2], ybound=[-54.0, 54.0, 1.2], zbound=[-10.0, 10.0, 20.0], dbound=[1.0, 61.0, 3.0])
FULL = dict(in_channels=256, out_channels=80, image_size=NUSC["image_size"], feature_size=NUSC["feature_size"],
            xbound=NUSC["xbound"], ybound=NUSC["ybound"], zbound=NUSC["zbound"], dbound=NUSC["dbound"])


def _calib(vt, rig, dev):
    t = {k: torch.from_numpy(v).to(dev) for k, v in rig.items()}
    return vt._calibration(t["camera_intrinsics"], t["camera2lidar"], t["img_aug_matrix"], t["lidar_aug_matrix"])


def _oracle_plan(vt, cal, B):
    """The same plan on the CPU oracle, from the same host-prepared matrices."""
    c = {k: v.cpu() for k, v in cal.items()}
    combine = cal["camera2lidar_rots"].matmul(cal["intrins_inverse"]).cpu().numpy()  # same device matmul as the product
    geom = oracle.frustum_geometry(vt.frustum.detach().cpu().numpy(), c["post_trans"].numpy(),
                                   c["post_rots_inverse"].numpy(), combine, c["camera2lidar_trans"].numpy(),
                                   c["extra_rots"].numpy(), c["extra_trans"].numpy())
    gf, kept, ranks, idx = oracle.bev_pool_aux(geom, B, np.array(vt._origin_host, np.float32),
                                               np.array(vt._dx_host, np.float32), np.array(vt._nx_host, np.int32))
    return geom, gf, kept, ranks, idx


def _tiny_rig(golden_lss):
    return {k[len("tiny_rig_"):]: golden_lss[k] for k in golden_lss.files if k.startswith("tiny_rig_")}


def test_plan_tiny_vs_oracle_and_reference(dev, golden_lss):
    vt = LSSTransform(**TINY).to(dev)
    rig = _tiny_rig(golden_lss)
    B = 2
    cal = _calib(vt, rig, dev)
    plan = vt.make_plan(**cal, with_reference_outputs=True, with_geometry=True)
    geom, gf, kept, ranks, idx = _oracle_plan(vt, cal, B)
    nk, m = [int(v) for v in plan.counts.cpu()]
    # geometry: bit-identical to the oracle's fixed-order evaluation; ~1 ulp from the reference's torch evaluation
    assert np.array_equal(plan.geom_xyz.cpu().numpy().reshape(geom.shape), geom)
    assert np.abs(plan.geom_xyz.cpu().numpy().reshape(geom.shape) - golden_lss["tiny_geom"]).max() < 5e-5
    # cells / kept / ranks: bit-identical to the oracle
    assert nk == kept.sum()
    assert np.array_equal(plan.kept.cpu().numpy().astype(bool), kept)
    assert np.array_equal(plan.ranks_sorted[:nk].cpu().numpy(), ranks)
    assert np.array_equal(plan.geom_sorted[:nk].cpu().numpy(), gf)
    # stable order: the k-th sorted row is frustum point src[k]
    pd = plan.sorted_pd[:nk].cpu().numpy().view(np.uint32)
    D, HW = plan.D, plan.HW
    pix, dd = pd >> 8, pd & 255
    src = (pix // HW) * (D * HW) + dd * HW + (pix % HW)
    assert np.array_equal(src, np.flatnonzero(kept)[idx])
    starts, lengths = oracle.intervals_from_ranks(ranks)
    assert m == len(starts)
    assert np.array_equal(plan.starts[:m].cpu().numpy(), starts)
    assert np.array_equal(plan.lengths[:m].cpu().numpy(), lengths)
    assert not plan.lengths[m:].any()
    g0 = gf[starts]
    nx = vt._nx_host
    assert np.array_equal(plan.cell_of_interval[:m].cpu().numpy(), ((g0[:, 3] * nx[2] + g0[:, 2]) * nx[0] + g0[:, 0]) * nx[1] + g0[:, 1])
    # against the REFERENCE's bev_pool_aux outputs (its own torch geometry): identical here
    assert np.array_equal(plan.kept.cpu().numpy().astype(bool), golden_lss["tiny_kept"])
    assert np.array_equal(plan.ranks_sorted[:nk].cpu().numpy(), golden_lss["tiny_ranks"])
    assert np.array_equal(plan.geom_sorted[:nk].cpu().numpy(), golden_lss["tiny_geom_feats"])


def test_plan_full_size_counts_vs_reference(dev, golden_lss):
    vt = LSSTransform(**FULL).to(dev)
    cal = _calib(vt, synthetic.camera_rig(batch=1), dev)
    plan = vt.make_plan(**cal, with_reference_outputs=True)
    nk, m = [int(v) for v in plan.counts.cpu()]
    nprime, ref_kept, ref_m, ref_maxlen, _ = [int(v) for v in golden_lss["full_counts"]]
    assert plan.nprime == nprime
    assert abs(nk - ref_kept) <= 20 and abs(m - ref_m) <= 20      # boundary-rounding flips only
    assert abs(int(plan.lengths.max()) - ref_maxlen) <= 4
    geom, gf, kept, ranks, idx = _oracle_plan(vt, cal, 1)
    assert nk == kept.sum()
    assert np.array_equal(plan.kept.cpu().numpy().astype(bool), kept)
    assert np.array_equal(plan.ranks_sorted[:nk].cpu().numpy(), ranks)
    rk = plan.ranks_sorted[:nk]
    assert bool((rk[1:] >= rk[:-1]).all())                          # sortedness
    assert int(plan.lengths[:m].sum()) == nk                        # intervals partition the kept rows


def _random_depth_feat(dev, P, D, C, seed):
    g = torch.Generator().manual_seed(seed)
    depth = torch.softmax(torch.randn(P, D, generator=g), 1).to(dev)
    feat = torch.randn(P, C, generator=g).to(dev)
    return depth, feat


def _to_ref_layout(depth, feat, BN, fH, fW):
    D, C = depth.shape[1], feat.shape[1]
    d = depth.view(BN, fH, fW, D).permute(0, 3, 1, 2).contiguous().cpu().numpy()
    f = feat.view(BN, fH, fW, C).permute(0, 3, 1, 2).contiguous().cpu().numpy()
    return d, f


@pytest.mark.parametrize("train_aug", [False, True])
def test_lift_splat_fwd_bwd_vs_oracle(dev, train_aug):
    cfg = dict(TINY, out_channels=80)
    vt = LSSTransform(**cfg).to(dev)
    B = 2
    rig = synthetic.camera_rig(batch=B, seed=3, train_aug=train_aug)
    rig["img_aug_matrix"][..., 0, 0] = rig["img_aug_matrix"][..., 1, 1] = 0.12
    rig["img_aug_matrix"][..., 0, 3], rig["img_aug_matrix"][..., 1, 3] = -8.0, -44.0
    cal = _calib(vt, rig, dev)
    plan = vt.make_plan(**cal)
    fH, fW = cfg["feature_size"]
    BN, D, C = B * 6, vt.D, 80
    depth, feat = _random_depth_feat(dev, BN * fH * fW, D, C, seed=1)
    depth.requires_grad_(True)
    feat.requires_grad_(True)
    out = lift_splat(depth, feat, plan)
    geom, gf, kept, ranks, idx = _oracle_plan(vt, cal, B)
    starts, lengths = oracle.intervals_from_ranks(ranks)
    src = np.flatnonzero(kept)[idx].astype(np.int32)
    d_ref, f_ref = _to_ref_layout(depth.detach(), feat.detach(), BN, fH, fW)
    nx = vt._nx_host
    want = oracle.lift_splat_fwd(d_ref, f_ref, src, gf, starts, lengths, B, nx[2], nx[0], nx[1])
    assert np.array_equal(out.detach().cpu().numpy(), want)          # same order, same rounding: bit-exact
    og = torch.randn(out.shape, generator=torch.Generator().manual_seed(2)).to(dev)
    out.backward(og)
    dd, df = oracle.lift_splat_bwd(og.cpu().numpy(), d_ref, f_ref, src, gf, starts, lengths)
    dd_pm = np.transpose(dd, (0, 2, 3, 1)).reshape(-1, D)
    df_pm = np.transpose(df, (0, 2, 3, 1)).reshape(-1, C)
    assert rel_err(depth.grad.cpu().numpy(), dd_pm) < 1e-5            # tree vs sequential channel sum
    assert rel_err(feat.grad.cpu().numpy(), df_pm) < 1e-5


def test_lift_splat_bf16_output_is_the_rounded_fp32_result(dev):
    """out_dtype = bf16: the forward stores round-to-nearest-even(fp32 sum) -- bit for bit what `.to(torch.bfloat16)` of the fp32
    output gives -- and the backward with a bf16 out_grad equals the fp32 kernel fed the widened gradient, bit for bit."""
    cfg = dict(TINY, out_channels=80)
    vt = LSSTransform(**cfg).to(dev)
    B = 2
    rig = synthetic.camera_rig(batch=B, seed=5, train_aug=True)
    rig["img_aug_matrix"][..., 0, 0] = rig["img_aug_matrix"][..., 1, 1] = 0.12
    rig["img_aug_matrix"][..., 0, 3], rig["img_aug_matrix"][..., 1, 3] = -8.0, -44.0
    plan = vt.make_plan(**_calib(vt, rig, dev))
    fH, fW = cfg["feature_size"]
    BN, D, C = B * 6, vt.D, 80
    depth, feat = _random_depth_feat(dev, BN * fH * fW, D, C, seed=2)
    d32, f32 = depth.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    d16, f16 = depth.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    o32 = lift_splat(d32, f32, plan)
    o16 = lift_splat(d16, f16, plan, torch.bfloat16)
    assert o16.dtype == torch.bfloat16 and torch.equal(o16, o32.to(torch.bfloat16))
    og = torch.randn(o32.shape, generator=torch.Generator().manual_seed(3)).to(dev).to(torch.bfloat16)
    o32.backward(og.float())
    o16.backward(og)
    assert torch.equal(d16.grad, d32.grad) and torch.equal(f16.grad, f32.grad)


def test_module_fused_equals_op_boundary_path(dev, golden_lss):
    """BaseViewTransform.bev_pool(x, geom) (the reference's materialised path through the bev_pool op)
    reproduces the reference's python-glue golden, and the fused lift-splat equals it."""
    vt = LSSTransform(**TINY).to(dev).eval()
    rig = _tiny_rig(golden_lss)
    B, N = 2, 6
    D, (fH, fW), C = vt.D, TINY["feature_size"], TINY["out_channels"]
    x = torch.randn(B, N, D, fH, fW, C, generator=torch.Generator().manual_seed(123))
    assert sha(x.numpy()) == str(golden_lss["tiny_x_sha"])
    geom = torch.from_numpy(golden_lss["tiny_geom"]).to(dev)
    bev = vt.bev_pool(x.to(dev), geom)
    assert bev.shape == golden_lss["tiny_bev"].shape
    assert rel_err(bev.cpu().numpy(), golden_lss["tiny_bev"]) < 1e-6
    # fused: random depth/feat through both paths
    cal = _calib(vt, rig, dev)
    plan = vt.make_plan(**cal)
    g = torch.Generator().manual_seed(7)
    depth = torch.softmax(torch.randn(B * N, D, fH, fW, generator=g), 1).to(dev)
    feat = torch.randn(B * N, C, fH, fW, generator=g).to(dev)
    fused = vt.lift_splat_bev(depth, feat, plan)
    xm = (depth.unsqueeze(1) * feat.unsqueeze(2)).view(B, N, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2)
    mat = vt.bev_pool(xm, vt.get_geometry(**{k: v for k, v in cal.items()}))
    assert torch.equal(fused, mat)


def test_depth_lss_transform_forward_backward(dev):
    """DepthLSSTransform end to end at reduced size: runs, shapes as the reference documents, gradients flow."""
    cfg = dict(TINY, in_channels=32, out_channels=16, downsample=2)
    vt = DepthLSSTransform(**cfg).to(dev).train()
    B, N = 2, 6
    rig = synthetic.camera_rig(batch=B, seed=1, train_aug=True)
    rig["img_aug_matrix"][..., 0, 0] = rig["img_aug_matrix"][..., 1, 1] = 0.12
    rig["img_aug_matrix"][..., 0, 3], rig["img_aug_matrix"][..., 1, 3] = -8.0, -44.0
    t = {k: torch.from_numpy(v).to(dev) for k, v in rig.items()}
    img = torch.randn(B, N, 32, 8, 22, device=dev, requires_grad=True)
    pts = [torch.from_numpy(synthetic.lidar_sweep(5000, seed=s)).to(dev) for s in (1, 2)]
    keep = [p.clone() for p in pts]
    x, depth_loss = vt(img, pts, t["lidar2image"], t["camera_intrinsics"], t["camera2lidar"], t["img_aug_matrix"],
                       t["lidar_aug_matrix"], None)
    assert x.shape == (B, 16, 45, 45)
    assert all(torch.equal(a, b) for a, b in zip(pts, keep))  # inputs not mutated
    (x.sum() + depth_loss).backward()
    assert img.grad is not None and torch.isfinite(img.grad).all() and img.grad.abs().sum() > 0
    assert vt.depthnet[0].weight.grad.abs().sum() > 0


def test_full_size_properties(dev):
    """BASELINE-size fused op (1 993 728 frustum points, C=80, 360x360): linearity in feat, mass
    conservation, and equality with the op-boundary bev_pool on the materialised tensor."""
    from bevfusion_amd.ops import bev_pool_ext
    vt = LSSTransform(**FULL).to(dev)
    cal = _calib(vt, synthetic.camera_rig(batch=1), dev)
    plan = vt.make_plan(**cal, with_reference_outputs=True)
    nk, m = [int(v) for v in plan.counts.cpu()]
    P, D, C = 6 * 32 * 88, vt.D, 80
    depth, f1 = _random_depth_feat(dev, P, D, C, seed=5)
    _, f2 = _random_depth_feat(dev, P, D, C, seed=6)
    o1, o2, o12 = lift_splat(depth, f1, plan), lift_splat(depth, f2, plan), lift_splat(depth, f1 + f2, plan)
    assert rel_err((o1 + o2).cpu().numpy(), o12.cpu().numpy()) < 1e-5
    # materialised path through the op: x rows in sorted order
    pd = plan.sorted_pd[:nk].long() & 0xFFFFFFFF
    pix, dd = pd >> 8, pd & 255
    x = depth[pix, dd].unsqueeze(1) * f1[pix]
    ref = bev_pool_ext.bev_pool_forward(x, plan.geom_sorted[:nk].contiguous(), plan.lengths[:m].contiguous(),
                                        plan.starts[:m].contiguous(), 1, 1, 360, 360)
    assert torch.equal(o1, ref)
    assert torch.allclose(o1.double().sum((0, 1, 2, 3)), x.double().sum(0), rtol=1e-6, atol=1e-3)


def test_camera_major_interval_order(dev, monkeypatch):
    """bfhip_bev_plan's `interval_order`: a permutation of the intervals, stably sorted by (sample, camera) of the first
    member -- and lift_splat_fwd produces bit-identical cells whether it walks them in that order or in rank order."""
    from bevfusion_amd import depth_lss, synthetic
    from bevfusion_amd.depth_lss import LSSTransform, lift_splat
    cfg = dict(in_channels=16, out_channels=80, image_size=(64, 176), feature_size=(8, 22), xbound=[-54.0, 54.0, 1.2],
               ybound=[-54.0, 54.0, 1.2], zbound=[-10.0, 10.0, 20.0], dbound=[1.0, 61.0, 1.0])
    vt = LSSTransform(**cfg).to(dev)
    rig = synthetic.camera_rig(batch=2, seed=3, train_aug=True)
    rig["img_aug_matrix"][..., 0, 0] = rig["img_aug_matrix"][..., 1, 1] = 0.12
    rig["img_aug_matrix"][..., 0, 3], rig["img_aug_matrix"][..., 1, 3] = -8.0, -44.0
    t = {k: torch.from_numpy(v).to(dev) for k, v in rig.items()}
    cal = vt._calibration(t["camera_intrinsics"], t["camera2lidar"], t["img_aug_matrix"], t["lidar_aug_matrix"])
    monkeypatch.setattr(depth_lss, "CAMERA_MAJOR", True)   # opt-in (BFHIP_LIFT_SPLAT_ORDER=1)
    plan = vt.make_plan(**cal)
    n_kept, m = [int(v) for v in plan.counts.cpu()]
    order = plan.interval_order.cpu().numpy()[:m]
    assert np.array_equal(np.sort(order), np.arange(m))
    HW = 8 * 22
    pd = plan.sorted_pd.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    group = (pd[plan.starts.cpu().numpy()[:m]] >> 8) // HW
    g_sorted = group[order]
    assert np.all(np.diff(g_sorted) >= 0)                                   # camera-major
    assert all(np.all(np.diff(order[g_sorted == g]) > 0) for g in np.unique(group))  # rank order inside a camera (stable)
    P, D, C = 2 * 6 * HW, vt.D, 80
    gen = torch.Generator().manual_seed(0)
    depth = torch.softmax(torch.randn(P, D, generator=gen), 1).to(dev)
    feat = torch.randn(P, C, generator=gen).to(dev)
    a = lift_splat(depth, feat, plan)
    plan.interval_order = None                                              # rank order
    b = lift_splat(depth, feat, plan)
    assert torch.equal(a, b)


@pytest.mark.parametrize("C", [80, 8, 16])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_lift_splat_bf16_feature_rows_are_bit_identical(dev, C, out_dtype):
    """feat handed over in bf16 (what a bf16 depthnet stores; the reference widens it with x.float(), BF/depth_lss.py:467-468):
    the forward equals the fp32-feature kernel fed the widened values bit for bit -- and therefore the oracle -- for fp32 and
    bf16 outputs; d_feat is the fp32 kernel's d_feat rounded once to bf16 (bit for bit), d_depth agrees to the reduction
    tree's rounding (1e-5 rel: 10 lanes x 8 channels instead of 20 x 4)."""
    cfg = dict(TINY, out_channels=C)
    vt = LSSTransform(**cfg).to(dev)
    B = 2
    rig = synthetic.camera_rig(batch=B, seed=7, train_aug=True)
    rig["img_aug_matrix"][..., 0, 0] = rig["img_aug_matrix"][..., 1, 1] = 0.12
    rig["img_aug_matrix"][..., 0, 3], rig["img_aug_matrix"][..., 1, 3] = -8.0, -44.0
    cal = _calib(vt, rig, dev)
    plan = vt.make_plan(**cal)
    fH, fW = cfg["feature_size"]
    BN, D = B * 6, vt.D
    depth, feat = _random_depth_feat(dev, BN * fH * fW, D, C, seed=4)
    feat16 = feat.to(torch.bfloat16)
    d32, f32 = depth.clone().requires_grad_(True), feat16.float().requires_grad_(True)
    d16, f16 = depth.clone().requires_grad_(True), feat16.clone().requires_grad_(True)
    o32 = lift_splat(d32, f32, plan, out_dtype)
    o16 = lift_splat(d16, f16, plan, out_dtype)
    assert o16.dtype == out_dtype and torch.equal(o16, o32)
    if out_dtype == torch.float32:
        geom, gf, kept, ranks, idx = _oracle_plan(vt, cal, B)
        starts, lengths = oracle.intervals_from_ranks(ranks)
        src = np.flatnonzero(kept)[idx].astype(np.int32)
        d_ref, f_ref = _to_ref_layout(depth, feat16.float(), BN, fH, fW)
        nx = vt._nx_host
        want = oracle.lift_splat_fwd(d_ref, f_ref, src, gf, starts, lengths, B, nx[2], nx[0], nx[1])
        assert np.array_equal(o16.detach().cpu().numpy(), want)
    og = torch.randn(o32.shape, generator=torch.Generator().manual_seed(3)).to(dev).to(out_dtype)
    o32.backward(og)
    o16.backward(og)
    assert f16.grad.dtype == torch.bfloat16 and torch.equal(f16.grad, f32.grad.to(torch.bfloat16))
    assert rel_err(d16.grad.cpu().numpy(), d32.grad.cpu().numpy()) < 1e-5
    # a feature matrix that is a channel slice of a wider bf16 tensor (pitch > C, 16-byte aligned rows) is consumed in place
    wide = torch.zeros(feat16.shape[0], C + 24, dtype=torch.bfloat16, device=dev)
    wide[:, 8:8 + C] = feat16
    o_slice = lift_splat(depth, wide[:, 8:8 + C], plan, out_dtype)
    assert torch.equal(o_slice, o32.detach())


def test_depth_lss_bf16_features_equal_the_widened_path(dev, monkeypatch):
    """DepthLSSTransform with bf16 conv stacks: handing the feature channels to the lift-splat as bf16 (default) gives the
    same BEV map and the same input / weight gradients as widening the whole depthnet output first (the reference's
    x.float(), BFHIP_LIFT_SPLAT_BF16_FEAT=0); the op-level test above shows the kernels themselves are bit-identical."""
    from bevfusion_amd import depth_lss
    cfg = dict(TINY, in_channels=32, out_channels=16, downsample=2)
    torch.manual_seed(0)
    vt = DepthLSSTransform(**cfg).to(dev).train()
    vt.conv_dtype = torch.bfloat16
    B, N = 2, 6
    rig = synthetic.camera_rig(batch=B, seed=1, train_aug=True)
    rig["img_aug_matrix"][..., 0, 0] = rig["img_aug_matrix"][..., 1, 1] = 0.12
    rig["img_aug_matrix"][..., 0, 3], rig["img_aug_matrix"][..., 1, 3] = -8.0, -44.0
    t = {k: torch.from_numpy(v).to(dev) for k, v in rig.items()}
    pts = [torch.from_numpy(synthetic.lidar_sweep(5000, seed=s)).to(dev) for s in (1, 2)]
    img0 = torch.randn(B, N, 32, 8, 22, device=dev)
    res = {}
    state = {k: v.clone() for k, v in vt.state_dict().items()}
    for flag in (True, False):
        monkeypatch.setattr(depth_lss, "BF16_FEAT", flag)
        vt.load_state_dict(state)
        for p in vt.parameters():
            p.grad = None
        img = img0.clone().requires_grad_(True)
        x, depth_loss = vt(img, pts, t["lidar2image"], t["camera_intrinsics"], t["camera2lidar"], t["img_aug_matrix"],
                           t["lidar_aug_matrix"], None)
        (x.float().square().mean() + depth_loss).backward()
        res[flag] = (x.detach().clone(), img.grad.clone(), vt.depthnet[0].weight.grad.clone())
    # same values enter the pooling either way; what may differ is the fp32 softmax's summation order (last-dim kernel on the
    # split-off logits vs the strided-dim kernel on the widened tensor), i.e. isolated 1-ulp flips that bf16 layers downstream
    # can turn into one bf16 step on single elements: relative L2, not bit equality
    for a, b in zip(res[True], res[False]):
        l2 = float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30))
        assert l2 < 2e-3, l2
