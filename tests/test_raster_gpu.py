"""GPU parity: csrc/raster.hip (sparse depth rasteriser + GT depth histogram) vs the oracle and the reference goldens."""
import numpy as np
import pytest
import torch

import bevfusion_amd  # noqa: F401
import oracle
from bevfusion_amd import synthetic
from bevfusion_amd.depth_lss import DepthLSSTransform

from test_oracle_golden import _raster_inputs

pytestmark = pytest.mark.gpu

TINY = dict(in_channels=16, out_channels=8, image_size=(64, 176), feature_size=(8, 22), xbound=[-54.0, 54.0, 1.2],
            ybound=[-54.0, 54.0, 1.2], zbound=[-10.0, 10.0, 20.0], dbound=[1.0, 61.0, 3.0])


def test_rasteriser_and_histogram(dev, golden_lss):
    rig, pts = _raster_inputs(golden_lss)
    vt = DepthLSSTransform(**TINY).to(dev)
    t = {k: torch.from_numpy(v).to(dev) for k, v in rig.items()}
    P = [torch.from_numpy(p).to(dev) for p in pts]
    keep = [p.clone() for p in P]
    img = torch.zeros(2, 6, 16, 8, 22, device=dev)
    depth, counts = vt.rasterise_depth(img, P, t["lidar2image"], t["img_aug_matrix"], t["lidar_aug_matrix"], with_histogram=True)
    assert all(torch.equal(a, b) for a, b in zip(P, keep))      # inputs not mutated (the reference mutates them)
    inv = torch.inverse(t["lidar_aug_matrix"]).cpu().numpy()     # same device inverse as the product uses
    want = np.stack([oracle.rasterise_depth(pts[b], inv[b, :3, :3], rig["lidar_aug_matrix"][b, :3, 3], rig["lidar2image"][b],
                                            rig["img_aug_matrix"][b], 64, 176) for b in range(2)])
    got = depth.cpu().numpy()[:, :, 0]
    assert np.array_equal(got, want)                               # bit-exact vs the oracle (last point wins)
    ref = golden_lss["rast_depth"][:, :, 0]
    assert int((np.abs(got - ref) > 1e-4 * np.maximum(np.abs(ref), 1.0)).sum()) <= 12   # vs the reference's own loop
    # histogram accumulated in the same pass == oracle histogram of the image == standalone histogram kernel
    distr, c3 = vt.depth_distribution(counts=counts.clone())
    wc, wd = oracle.depth_histogram(want.reshape(12, 64, 176), 8, 22, 20, TINY["dbound"])
    assert np.array_equal(c3.cpu().numpy().reshape(wc.shape), wc)
    assert np.array_equal(distr.cpu().numpy().reshape(wd.shape), wd)
    d2, c2 = vt.gt_depth_distribution(depth.view(12, 1, 64, 176), 2, 6)
    assert torch.equal(d2.reshape(-1), distr.reshape(-1)) and torch.equal(c2.reshape(-1), c3.reshape(-1))
    # the reference's own histogram of ITS depth images through the HIP kernel: bit-exact
    d3, c3r = vt.gt_depth_distribution(torch.from_numpy(golden_lss["rast_depth"]).to(dev).view(12, 1, 64, 176), 2, 6)
    assert np.array_equal(c3r.cpu().numpy(), golden_lss["rast_counts"])
    assert np.array_equal(d3.cpu().numpy(), golden_lss["rast_gt_distr"])


def test_duplicates_last_point_wins_and_empty(dev):
    vt = DepthLSSTransform(**TINY).to(dev)
    rig = synthetic.camera_rig(batch=1)
    rig["img_aug_matrix"][..., 0, 0] = rig["img_aug_matrix"][..., 1, 1] = 0.12
    rig["img_aug_matrix"][..., 0, 3], rig["img_aug_matrix"][..., 1, 3] = -8.0, -44.0
    t = {k: torch.from_numpy(v).to(dev) for k, v in rig.items()}
    img = torch.zeros(1, 6, 16, 8, 22, device=dev)
    # 1000 copies of the same point with different intensity channel do not matter; depth identical -> stable
    p = torch.tensor([[0.5, 12.0, -0.2, 0, 0]], device=dev).repeat(1000, 1)
    p[:, 1] += torch.linspace(0, 1e-3, 1000, device=dev)      # tiny forward offsets: same pixel, increasing depth
    d = vt.rasterise_depth(img, [p], t["lidar2image"], t["img_aug_matrix"], t["lidar_aug_matrix"])
    inv = torch.inverse(t["lidar_aug_matrix"]).cpu().numpy()
    want = oracle.rasterise_depth(p.cpu().numpy(), inv[0, :3, :3], rig["lidar_aug_matrix"][0, :3, 3], rig["lidar2image"][0],
                                  rig["img_aug_matrix"][0], 64, 176)
    assert np.array_equal(d.cpu().numpy()[0, :, 0], want) and (want > 0).sum() >= 1
    empty = vt.rasterise_depth(img, [torch.zeros(0, 5, device=dev)], t["lidar2image"], t["img_aug_matrix"], t["lidar_aug_matrix"])
    assert not empty.any()


@pytest.mark.parametrize("shape", [(24, 256, 704), (3, 37, 53), (1, 1, 5)])
def test_first_dtransform_layer_gradients(dev, shape):
    """Conv2d(1, 8, 1) on the one-channel depth image (BF/depth_lss.py:592-594) as y = b + d * w in channels-last bf16, with
    both parameter gradients from one pass over dy (csrc/raster.hip depth_lift_bwd_kernel): forward = the conv on the bf16-rounded
    operands up to the final bf16 rounding; dw / db against the fp64 sums over the same bf16 values (<= 1e-5 rel: fp32 accumulation only, where
    torch's own bf16 backward rounds the product tensor first); at the full batch-4 size (24 x 256 x 704) and on odd extents;
    deterministic run to run."""
    from bevfusion_amd.depth_lss import _DepthLift
    BN, H, W = shape
    g = torch.Generator(device="cpu").manual_seed(5)
    d = (torch.rand(BN, H, W, 1, generator=g) * 60).to(dev)
    d = torch.where(torch.rand(BN, H, W, 1, generator=g).to(dev) < 0.9, torch.zeros_like(d), d).to(torch.bfloat16)   # sparse depth image
    w = torch.randn(8, 1, 1, 1, generator=g).to(dev).requires_grad_(True)
    b = torch.randn(8, generator=g).to(dev).requires_grad_(True)
    dy = torch.randn(BN, H, W, 8, generator=g).to(dev).to(torch.bfloat16)
    y = _DepthLift.apply(d, w, b)
    ref = torch.nn.functional.conv2d(d.permute(0, 3, 1, 2).float(), w.detach().to(torch.bfloat16).float(), b.detach().to(torch.bfloat16).float())
    # one bf16 rounding of b + d * w either way (fused vs separate multiply-add in fp32 may flip a rounding: <= 1 ulp)
    assert y.dtype == torch.bfloat16 and bool(((y.permute(0, 3, 1, 2).float() - ref).abs() <= 2.0 ** -7 * ref.abs() + 1e-30).all())
    y.backward(dy)
    dw64 = (dy.double() * d.double()).sum(dim=(0, 1, 2))
    db64 = dy.double().sum(dim=(0, 1, 2))
    assert w.grad.shape == w.shape and w.grad.dtype == torch.float32
    assert float((w.grad.view(8).double() - dw64).norm() / dw64.norm().clamp_min(1e-30)) < 1e-5
    assert float((b.grad.double() - db64).norm() / db64.norm()) < 1e-5
    first = (w.grad.clone(), b.grad.clone())
    w.grad = b.grad = None
    _DepthLift.apply(d, w, b).backward(dy)
    assert torch.equal(w.grad, first[0]) and torch.equal(b.grad, first[1])
