"""GPU parity: csrc/spconv.hip (rulebooks, gather-GEMM, dgrad, wgrad, dense) vs the oracle.
Rulebooks bit-exact (indices); features within 1e-3 rel (fp32 MFMA vs fp64-accumulated oracle)."""
import numpy as np
import pytest
import torch

import bevfusion_amd  # noqa: F401
import oracle
from bevfusion_amd import synthetic
from bevfusion_amd.sparse_encoder import BEVFusionSparseEncoder, SparseBasicBlock
from bevfusion_amd.spconv import (SparseConv3d, SparseConvTensor, SubMConv3d, build_sparse_rulebook,
                                  build_subm_rulebook)

from test_spconv_oracle import random_sparse
from util import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star: within 1e-3 rel for fp32 features


def _voxel_indices(dev, n_points=40000, batch=2):
    """Realistic occupancy: voxel coordinates of synthetic sweeps on the nuScenes grid."""
    N = synthetic.NUSC
    out = []
    for b in range(batch):
        pts = synthetic.lidar_sweep(n_points, seed=50 + b)
        _, coors, _ = oracle.hard_voxelize(pts, N["voxel_size"], N["point_cloud_range"], 10, 120000)
        out.append(np.concatenate([np.full((len(coors), 1), b, np.int32), coors], 1))
    return np.concatenate(out, 0)


@pytest.fixture
def sorted_rows(monkeypatch):
    """The mask-sorted row orders are opt-in (BFHIP_SPCONV_SORT=1); the tests of that path switch them on."""
    from bevfusion_amd import spconv
    monkeypatch.setattr(spconv, "SORT_ROWS", True)


def test_subm_rulebook_bit_exact(dev, sorted_rows):
    idx = _voxel_indices(dev)
    shape = [1440, 1440, 41]
    want = oracle.rulebook_subm(idx, shape, 3)
    data = build_subm_rulebook(torch.from_numpy(idx).to(dev), 2, shape, [3, 3, 3], [1, 1, 1])
    assert np.array_equal(data.pair_fwd.cpu().numpy(), want)
    assert int(data.n_pairs.sum().item()) == (want >= 0).sum()
    # symmetry used by the SubM backward: pair[k][n] = j  <=>  pair[KV-1-k][j] = n
    pf = data.pair_fwd.cpu().numpy()
    k, n = np.nonzero(pf >= 0)
    assert np.array_equal(pf[26 - k, pf[k, n]], n)
    _check_fused_sorted_rows(data)


def _check_fused_sorted_rows(data):
    """The row masks / sorted orders that leave with the table-filling launches equal the stand-alone pass over the table."""
    from bevfusion_amd.spconv import sort_rows
    for table, mask, perm in ((data.pair_fwd, data.mask_fwd, data.perm_fwd), (data.pair_bwd, data.mask_bwd, data.perm_bwd)):
        if table is None:
            continue
        want_mask, want_perm = sort_rows(table)
        assert torch.equal(mask, want_mask)
        assert torch.equal(perm, want_perm)


@pytest.mark.parametrize("shape,ksize,stride,padding", [([1440, 1440, 41], 3, 2, 1), ([360, 360, 11], 3, 2, (1, 1, 0)),
                                                        ([180, 180, 5], (1, 1, 3), (1, 1, 2), 0)])
def test_sparse_rulebook_bit_exact(dev, sorted_rows, shape, ksize, stride, padding):
    if shape[0] == 1440:
        idx = _voxel_indices(dev)
    else:
        idx, _ = random_sparse(2, shape, 20000 if shape[0] == 360 else 8000, 1, seed=shape[0])
    ks = [ksize] * 3 if isinstance(ksize, int) else list(ksize)
    st = [stride] * 3 if isinstance(stride, int) else list(stride)
    pd = [padding] * 3 if isinstance(padding, int) else list(padding)
    out_idx, pf, pb, out_shape = oracle.rulebook_sparse(idx, shape, ks, st, pd)
    data = build_sparse_rulebook(torch.from_numpy(idx).to(dev), 2, shape, ks, st, pd, [1, 1, 1])
    assert data.out_spatial_shape == list(out_shape)
    assert np.array_equal(data.out_indices.cpu().numpy(), out_idx)
    assert np.array_equal(data.pair_fwd.cpu().numpy(), pf)
    assert np.array_equal(data.pair_bwd.cpu().numpy(), pb)
    assert int(data.n_pairs.sum().item()) == (pf >= 0).sum()
    _check_fused_sorted_rows(data)


@pytest.mark.parametrize("cin,cout", [(5, 16), (16, 16), (32, 32), (16, 32), (32, 64), (64, 64), (128, 128), (7, 9)])
def test_conv_fwd_bwd_vs_oracle(dev, cin, cout):
    B, shape, n = 2, (40, 36, 9), 6000
    idx, feats = random_sparse(B, shape, n, cin, seed=cin + cout)
    conv = SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="k").to(dev)
    w = conv.weight.detach().cpu().numpy()
    x = SparseConvTensor(torch.from_numpy(feats).to(dev).requires_grad_(True), torch.from_numpy(idx).to(dev), shape, B)
    out = conv(x)
    pair = oracle.rulebook_subm(idx, shape, 3)
    want = oracle.spconv_fwd(feats, w, pair)
    assert rel_err(out.features.detach().cpu().numpy(), want) < TOL
    g = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    out.features.backward(g)
    d_in, d_w = oracle.spconv_bwd(feats, w, g.cpu().numpy(), pair)
    assert rel_err(x.features.grad.cpu().numpy(), d_in) < TOL
    assert rel_err(conv.weight.grad.cpu().numpy(), d_w) < TOL


@pytest.mark.parametrize("cin,cout", [(16, 32), (64, 64)])
def test_conv_with_mask_sorted_rows(dev, sorted_rows, cin, cout):
    """The opt-in mask-sorted tile order (BFHIP_SPCONV_SORT=1) changes which rows share a tile, not any row's sum."""
    from bevfusion_amd import spconv
    B, shape, n = 2, (40, 36, 9), 6000
    idx, feats = random_sparse(B, shape, n, cin, seed=5)
    res = []
    for srt in (True, False):
        spconv.SORT_ROWS = srt
        torch.manual_seed(0)
        conv = SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=False).to(dev)
        sub = SubMConv3d(cout, cout, 3, padding=1, bias=False, indice_key="s").to(dev)
        x = SparseConvTensor(torch.from_numpy(feats).to(dev).requires_grad_(True), torch.from_numpy(idx).to(dev), shape, B)
        out = sub(conv(x))
        assert (out.indice_dict["s"].perm_fwd is not None) == srt
        out.features.square().sum().backward()
        res.append((out.features.detach().clone(), x.features.grad.clone(), conv.weight.grad.clone(), sub.weight.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2:], res[1][2:]):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5   # fp32 summation order of the weight gradient only


def test_strided_conv_fwd_bwd_vs_oracle(dev):
    B, shape, n, cin, cout = 2, (40, 36, 11), 5000, 32, 64
    idx, feats = random_sparse(B, shape, n, cin, seed=9)
    conv = SparseConv3d(cin, cout, 3, stride=2, padding=(1, 1, 0), bias=False).to(dev)
    w = conv.weight.detach().cpu().numpy()
    x = SparseConvTensor(torch.from_numpy(feats).to(dev).requires_grad_(True), torch.from_numpy(idx).to(dev), shape, B)
    out = conv(x)
    out_idx, pf, pb, out_shape = oracle.rulebook_sparse(idx, shape, 3, 2, (1, 1, 0))
    assert out.spatial_shape == list(out_shape) and np.array_equal(out.indices.cpu().numpy(), out_idx)
    want = oracle.spconv_fwd(feats, w, pf)
    assert rel_err(out.features.detach().cpu().numpy(), want) < TOL
    g = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(2)).to(dev)
    out.features.backward(g)
    d_in, d_w = oracle.spconv_bwd(feats, w, g.cpu().numpy(), pf)
    assert rel_err(x.features.grad.cpu().numpy(), d_in) < TOL
    assert rel_err(conv.weight.grad.cpu().numpy(), d_w) < TOL


def test_conv_vs_dense_torch_conv3d(dev):
    """Independent numeric oracle: torch conv3d on the densified tensor (SubM = sampled at input sites)."""
    import torch.nn.functional as F
    B, shape, n, cin, cout = 1, (12, 11, 10), 500, 16, 32
    idx, feats = random_sparse(B, shape, n, cin, seed=3)
    conv = SubMConv3d(cin, cout, 3, padding=1, bias=True).to(dev)
    x = SparseConvTensor(torch.from_numpy(feats).to(dev), torch.from_numpy(idx).to(dev), shape, B)
    out = conv(x).features.detach().cpu()
    dense = torch.zeros(B, cin, *shape)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = torch.from_numpy(feats)
    ref = F.conv3d(dense.double(), conv.weight.detach().cpu().permute(0, 4, 1, 2, 3).double(),
                   conv.bias.detach().cpu().double(), padding=1)
    want = ref[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]
    assert rel_err(out.numpy(), want.numpy()) < TOL


CONV3D_CASES = [  # (subm, cin, cout, ksize, stride, padding, shape, n)
    (True, 128, 128, 3, 1, 1, (14, 13, 9), 900),                        # stage-4 SubM
    (True, 64, 64, 3, 1, 1, (14, 13, 9), 900),
    (False, 32, 64, 3, 2, 1, (20, 18, 11), 1500),                        # spconv2: k3 s2 p1
    (False, 64, 128, 3, 2, (1, 1, 0), (20, 18, 11), 1500),               # spconv3: k3 s2 p(1,1,0)
    (False, 128, 128, (1, 1, 3), (1, 1, 2), 0, (12, 10, 5), 400),        # conv_out: k(1,1,3) s(1,1,2) p0
    (False, 16, 32, 3, 2, 1, (21, 19, 9), 1200),                         # spconv1 on odd extents
]


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("case", CONV3D_CASES, ids=lambda c: "%s_%d_%d" % ("subm" if c[0] else "strided", c[1], c[2]))
def test_conv_fwd_bwd_vs_dense_torch_conv3d(dev, case, bf16):
    """Independent numeric oracle for every conv geometry of the encoder (not the restated spconv semantics): torch's
    dense conv3d in fp64 on the densified tensor, forward AND autograd backward.  A regular sparse conv's outputs are the
    dense conv's values at the sites any active input reaches (every other dense output is exactly the bias-free zero);
    SubM = the dense conv sampled at the input sites.  fp32: <= 1e-3 rel; bf16 autocast: <= 1e-2 rel (north star)."""
    import torch.nn.functional as F
    subm, cin, cout, ks, st, pad, shape, n = case
    B = 2
    idx, feats = random_sparse(B, shape, n, cin, seed=cin + cout + int(subm))
    conv = (SubMConv3d(cin, cout, ks, padding=pad, bias=False) if subm
            else SparseConv3d(cin, cout, ks, stride=st, padding=pad, bias=False)).to(dev)
    x = SparseConvTensor(torch.from_numpy(feats).to(dev).requires_grad_(True), torch.from_numpy(idx).to(dev), list(shape), B)
    if bf16:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv(x)
    else:
        y = conv(x)
    oi = y.indices.cpu().numpy().astype(np.int64)
    g = torch.randn(y.features.shape, generator=torch.Generator().manual_seed(5))
    y.features.backward(g.to(dev).to(y.features.dtype))
    # dense fp64 twin
    dense = torch.zeros(B, cin, *shape, dtype=torch.float64)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = torch.from_numpy(feats).double()
    dense.requires_grad_(True)
    w = conv.weight.detach().cpu().double().permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)  # (out,kD,kH,kW,in) -> (out,in,k..)
    ref = F.conv3d(dense, w, None, stride=st, padding=pad)
    assert list(ref.shape[2:]) == list(y.spatial_shape)
    want = ref[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]]
    mask = torch.zeros(ref.shape[0], *ref.shape[2:], dtype=torch.bool)
    mask[oi[:, 0], oi[:, 1], oi[:, 2], oi[:, 3]] = True
    if not subm:
        assert float(ref.detach().abs().amax(1)[~mask].max() if (~mask).any() else 0.0) == 0.0  # every reached site is listed
    gd = torch.zeros_like(ref)
    gd[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]] = g.double()
    ref.backward(gd)
    d_in = dense.grad[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]
    d_w = w.grad.permute(0, 2, 3, 4, 1)
    tol = 1e-2 if bf16 else TOL
    assert rel_err(y.features.detach().float().cpu().numpy(), want.detach().numpy()) < tol
    assert rel_err(x.features.grad.float().cpu().numpy(), d_in.numpy()) < tol
    assert rel_err(conv.weight.grad.float().cpu().numpy(), d_w.numpy()) < tol


def test_dense_and_to_bev(dev):
    B, (X, Y, Z), n, c = 2, (18, 16, 2), 200, 128
    idx, feats = random_sparse(B, (X, Y, Z), n, c, seed=4)
    x = SparseConvTensor(torch.from_numpy(feats).to(dev).requires_grad_(True), torch.from_numpy(idx).to(dev), [X, Y, Z], B)
    bev = x.to_bev()
    assert np.array_equal(bev.detach().cpu().numpy(), oracle.sparse_to_bev(feats, idx, B, X, Y, Z))
    d = x.dense()
    assert d.shape == (B, c, X, Y, Z)
    assert torch.equal(d.permute(0, 1, 4, 2, 3).reshape(B, c * Z, X, Y), bev)
    g = torch.randn_like(bev)
    bev.backward(g)
    want = g.view(B, c, Z, X, Y)[idx[:, 0], :, idx[:, 3], idx[:, 1], idx[:, 2]]
    assert torch.equal(x.features.grad.cpu(), want.cpu())


def test_empty_input(dev):
    conv = SubMConv3d(16, 16, 3, padding=1, bias=False).to(dev)
    x = SparseConvTensor(torch.zeros(0, 16, device=dev), torch.zeros(0, 4, dtype=torch.int32, device=dev), [8, 8, 8], 1)
    assert conv(x).features.shape == (0, 16)
    down = SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False).to(dev)
    y = down(x)
    assert y.features.shape == (0, 32) and y.spatial_shape == [4, 4, 4]


def test_basic_block_shapes_like_reference_test(dev):
    """Mirror of the reference's test_SparseBasicBlock (tests/.../test_spconv_module.py:18-48): 4 voxels in, [4,4] out."""
    feats = torch.tensor([[6.56126, 0.9648336, -1.7339306, 0.315], [6.8162713, -2.480431, -1.3616394, 0.36],
                          [11.643568, -4.744306, -1.3580885, 0.16], [23.482342, 6.5036807, 0.5806964, 0.35]],
                         dtype=torch.float32, device=dev)
    coords = torch.tensor([[0, 12, 819, 131], [0, 16, 750, 136], [1, 16, 705, 232], [1, 35, 930, 469]],
                          dtype=torch.int32, device=dev)
    x = SparseConvTensor(feats, coords, [41, 1600, 1408], 2)
    block = SparseBasicBlock(4, 4, conv_cfg=dict(type="SubMConv3d", indice_key="subm1"),
                             norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01)).to(dev)
    assert block.conv1.in_channels == 4 and block.conv2.out_channels == 4
    assert block(x).features.shape == torch.Size([4, 4])


def test_encoder_nuscenes_chain(dev):
    """BEVFusionSparseEncoder with the reference's nuScenes config (bevfusion_lidar...py:56-65): output
    [B, 256, 180, 180], stage shapes 1440->720->360->180 / 41->21->11->5->2, backward runs."""
    enc = BEVFusionSparseEncoder(in_channels=5, sparse_shape=[1440, 1440, 41], order=("conv", "norm", "act"),
                                 norm_cfg=dict(type="BN1d", eps=0.001, momentum=0.01),
                                 encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                                 encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, (1, 1, 0)), (0, 0)),
                                 block_type="basicblock", return_middle_feats=True).to(dev)
    n_convs = sum(1 for m in enc.modules() if isinstance(m, (SubMConv3d, SparseConv3d)))
    assert n_convs == 21 and sum(p.numel() for p in enc.parameters()) > 2.6e6
    idx = _voxel_indices(dev, 40000, 2)
    feats = torch.randn(idx.shape[0], 5, device=dev, requires_grad=True)
    out, mids = enc(feats, torch.from_numpy(idx).to(dev), 2)
    assert out.shape == (2, 256, 180, 180)
    assert [m.spatial_shape for m in mids] == [[720, 720, 21], [360, 360, 11], [180, 180, 5], [180, 180, 5]]
    out.mean().backward()
    assert torch.isfinite(feats.grad).all() and feats.grad.abs().sum() > 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in enc.parameters())


def test_encoder_layer_vs_oracle_end_to_end(dev):
    """conv_input + one strided conv on real voxel occupancy, features checked against the oracle chain."""
    idx = _voxel_indices(dev, 20000, 1)
    rng = np.random.default_rng(0)
    feats = rng.standard_normal((idx.shape[0], 5)).astype(np.float32)
    c1 = SubMConv3d(5, 16, 3, padding=1, bias=False).to(dev)
    c2 = SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False).to(dev)
    x = SparseConvTensor(torch.from_numpy(feats).to(dev), torch.from_numpy(idx).to(dev), [1440, 1440, 41], 1)
    y = c2(c1(x))
    p1 = oracle.rulebook_subm(idx, [1440, 1440, 41], 3)
    f1 = oracle.spconv_fwd(feats, c1.weight.detach().cpu().numpy(), p1)
    oi, pf, pb, osz = oracle.rulebook_sparse(idx, [1440, 1440, 41], 3, 2, 1)
    f2 = oracle.spconv_fwd(f1, c2.weight.detach().cpu().numpy(), pf)
    assert np.array_equal(y.indices.cpu().numpy(), oi)
    assert rel_err(y.features.detach().cpu().numpy(), f2) < TOL


@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 64), (64, 64), (128, 128), (16, 16), (32, 32)])
@pytest.mark.parametrize("bf16_features", [True, False])
def test_conv_bf16_autocast_vs_oracle(dev, cin, cout, bf16_features, monkeypatch):
    """Under bf16 autocast the gather-GEMM feeds the MFMA with bf16 (fp32 accumulate) and, with BF16_FEATURES, stores the
    activations in bf16: within the 1e-2 rel bf16 tolerance of the north star against the fp64-accumulated oracle.  The
    weight gradient accumulates in fp32 from the (bf16-rounded) saved features and output gradient."""
    from bevfusion_amd import spconv as sp
    monkeypatch.setattr(sp, "BF16_FEATURES", bf16_features)
    B, shape, n = 2, (40, 36, 9), 6000
    idx, feats = random_sparse(B, shape, n, cin, seed=cin * 3 + cout)
    conv = SubMConv3d(cin, cout, 3, padding=1, bias=False).to(dev)
    w = conv.weight.detach().cpu().numpy()
    x = SparseConvTensor(torch.from_numpy(feats).to(dev).requires_grad_(True), torch.from_numpy(idx).to(dev), shape, B)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = conv(x)
    assert out.features.dtype == (torch.bfloat16 if bf16_features else torch.float32)
    pair = oracle.rulebook_subm(idx, shape, 3)
    want = oracle.spconv_fwd(feats, w, pair)
    err = rel_err(out.features.detach().float().cpu().numpy(), want)
    assert 1e-6 < err < 1e-2, err     # really the bf16 path (not bit-identical to fp32), inside the bf16 budget
    g = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    out.features.backward(g.to(out.features.dtype))
    d_in, d_w = oracle.spconv_bwd(feats, w, g.cpu().numpy(), pair)
    assert x.features.grad.dtype == torch.float32
    assert rel_err(x.features.grad.cpu().numpy(), d_in) < 1e-2
    assert rel_err(conv.weight.grad.cpu().numpy(), d_w) < (1e-2 if bf16_features else TOL)  # fp32 accumulate either way


def test_encoder_bf16_features_vs_fp32_storage(dev, monkeypatch):
    """The whole sparse encoder (21 convs + BN) three ways: fp32; bf16 autocast with fp32 feature storage; bf16 autocast
    with bf16 feature storage.  The BEV output of both autocast modes stays within the bf16 budget of the fp32 run, and
    bf16 storage does not degrade the weight gradients relative to fp32 storage (cosine to the fp32 gradients), even for
    the first layer, 21 layers upstream of the loss."""
    from bevfusion_amd import spconv as sp
    from bevfusion_amd.sparse_encoder import BEVFusionSparseEncoder
    B, shape, n = 2, (96, 96, 41), 5000
    idx, feats = random_sparse(B, shape, n, 5, seed=11)
    outs, grads = {}, {}
    for mode in ("fp32", "amp_f32_store", "amp_bf16_store"):
        monkeypatch.setattr(sp, "BF16_FEATURES", mode == "amp_bf16_store")
        torch.manual_seed(0)
        enc = BEVFusionSparseEncoder(in_channels=5, sparse_shape=list(shape), norm_cfg=dict(type="BN1d", eps=0.001, momentum=0.01),
                                     encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                                     encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, (1, 1, 0)), (0, 0)),
                                     block_type="basicblock").to(dev).train()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode != "fp32"):
            bev = enc(torch.from_numpy(feats).to(dev), torch.from_numpy(idx).to(dev), B)
        assert bev.dtype == torch.float32  # to_bev() default: the reference's fp32 NCHW map
        bev.square().mean().backward()
        outs[mode] = bev.detach()
        grads[mode] = [p.grad.detach().double() for p in enc.parameters() if p.dim() == 5]
    ref = outs["fp32"].cpu().numpy()
    assert rel_err(outs["amp_f32_store"].cpu().numpy(), ref) < 5e-2
    assert rel_err(outs["amp_bf16_store"].cpu().numpy(), ref) < 5e-2
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))  # noqa: E731
    for i in (0, len(grads["fp32"]) // 2, len(grads["fp32"]) - 1):
        c32 = cos(grads["amp_f32_store"][i], grads["fp32"][i])
        c16 = cos(grads["amp_bf16_store"][i], grads["fp32"][i])
        assert c16 > 0.9 and c16 > c32 - 0.03, (i, c32, c16)


@pytest.mark.parametrize("C,res,relu", [(16, False, True), (32, True, True), (64, False, False), (128, True, True)])
def test_fused_bn1d_matches_torch(dev, C, res, relu):
    """BatchNorm1dAct (csrc/bn1d.hip) vs torch BatchNorm1d [+ add] [+ relu]: outputs, running stats and all gradients."""
    from bevfusion_amd.spconv import BatchNorm1dAct
    N = 30011
    torch.manual_seed(C)  # the affine parameters below come from the global generator
    g = torch.Generator().manual_seed(C)
    x = (torch.randn(N, C, generator=g) * 2 + 0.5).to(dev)
    r = torch.randn(N, C, generator=g).to(dev) if res else None
    bn = BatchNorm1dAct(C, eps=1e-3, momentum=0.01).to(dev).train()
    ref = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
        ref.weight.copy_(bn.weight); ref.bias.copy_(bn.bias)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    r1 = r.clone().requires_grad_(True) if res else None
    r2 = r.clone().requires_grad_(True) if res else None
    y = bn(x1, residual=r1, relu=relu)
    yr = ref(x2)
    if res:
        yr = yr + r2
    if relu:
        yr = torch.relu(yr)
    assert torch.allclose(y, yr, rtol=1e-5, atol=1e-5)
    assert torch.allclose(bn.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.state_dict()["num_batches_tracked"]) == 1  # lazy host counter, flushed when the state dict is read
    go = torch.randn(N, C, generator=g).to(dev)
    y.backward(go)
    yr.backward(go)
    # dx = gamma*invstd*(g - dbeta/N - xhat*dgamma/N): cancellation -> compare on the max-normalised scale.  Elements whose
    # pre-activation sits within rounding of the ReLU kink may be switched on in one implementation and off in the other
    # (1e-8 vs exactly 0): their gradient is excluded from the comparison.
    kink = ((y.detach().abs() < 1e-6) | (yr.detach().abs() < 1e-6)) if relu else torch.zeros_like(y, dtype=torch.bool)
    g1 = torch.where(kink & ((y.detach() > 0) != (yr.detach() > 0)), x2.grad, x1.grad)
    assert rel_err(g1.cpu().numpy(), x2.grad.cpu().numpy()) < 1e-4
    assert torch.allclose(bn.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(bn.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-3)
    if res:
        same_side = ~(kink & ((y.detach() > 0) != (yr.detach() > 0)))
        assert torch.allclose(r1.grad[same_side], r2.grad[same_side], rtol=1e-5, atol=1e-6)
    # eval mode and unsupported widths fall back to torch's own kernels with identical semantics
    bn.eval(); ref.eval()
    assert torch.allclose(bn(x, relu=relu), torch.relu(ref(x)) if relu else ref(x), rtol=1e-5, atol=1e-5)


def test_to_bev_channels_last_matches_nchw(dev):
    """The channels-last BEV output (f32 and bf16) holds the same values as the NCHW one; its backward reads a channel
    slice of a wider channels-last gradient in place (what torch.cat's backward hands over in the fuser)."""
    import numpy as np
    rng = np.random.default_rng(3)
    B, X, Y, Z, C, n = 2, 20, 24, 2, 16, 300
    cells = rng.choice(B * X * Y * Z, n, replace=False)
    idx = np.stack(np.unravel_index(cells, (B, X, Y, Z)), 1).astype(np.int32)
    feats = torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)).to(dev)
    from bevfusion_amd.spconv import SparseConvTensor
    f0 = feats.clone().requires_grad_(True)
    f1 = feats.clone().requires_grad_(True)
    t0 = SparseConvTensor(f0, torch.from_numpy(idx).to(dev), [X, Y, Z], B)
    t1 = SparseConvTensor(f1, torch.from_numpy(idx).to(dev), [X, Y, Z], B)
    a = t0.to_bev()
    b = t1.to_bev(channels_last=True)
    assert b.shape == a.shape == (B, C * Z, X, Y) and b.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(a, b)
    h = t1.to_bev(channels_last=True, dtype=torch.bfloat16)
    assert h.dtype == torch.bfloat16 and torch.equal(h, a.to(torch.bfloat16))
    wide = torch.randn(B, 8 + C * Z, X, Y, device=dev).contiguous(memory_format=torch.channels_last)
    a.backward(wide[:, 8:].contiguous())
    b.backward(wide[:, 8:])          # non-contiguous channel slice, consumed in place
    assert torch.equal(f0.grad, f1.grad)
    f1.grad = None
    h.backward(wide[:, 8:].to(torch.bfloat16))
    assert torch.allclose(f1.grad, f0.grad, rtol=1e-2, atol=1e-2)


def test_presized_rulebooks_hint_overflow_falls_back(dev):
    """The hint-capped presizing path: the caps of a forward are 1.5x the previous frame's N_out + 4096, so a frame several
    times larger overflows them; the overflowing level and its successors (counted from a truncated input) must take the
    per-layer path, results stay bit-equal to presizing off (forward and gradients), and the next frame is planned again."""
    from bevfusion_amd import spconv as sp
    from bevfusion_amd.sparse_encoder import BEVFusionSparseEncoder
    B, shape = 2, (96, 96, 41)
    small = random_sparse(B, shape, 1500, 5, seed=31)
    large = random_sparse(B, shape, 40000, 5, seed=32)
    torch.manual_seed(0)

    def make():
        torch.manual_seed(0)
        return BEVFusionSparseEncoder(in_channels=5, sparse_shape=list(shape), norm_cfg=dict(type="BN1d", eps=0.001, momentum=0.01),
                                      encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                                      encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, (1, 1, 0)), (0, 0)), block_type="basicblock",
                                      return_middle_feats=True).to(dev).train()

    planned_levels = []
    orig = sp.prepare_strided_rulebooks

    def spy(*a, **k):
        plans = orig(*a, **k)
        planned_levels.append(len(plans))
        return plans

    out = {}
    for presize in (True, False):
        enc = make()
        enc.presize_rulebooks = presize
        sp_mod_fn = sp.prepare_strided_rulebooks
        import bevfusion_amd.sparse_encoder as se
        se.prepare_strided_rulebooks = spy if presize else orig
        try:
            res = []
            for idx, feats in (small, large, large):
                f = torch.from_numpy(feats).to(dev).requires_grad_(True)
                bev, mid = enc(f, torch.from_numpy(idx).to(dev), B)
                bev.square().mean().backward()
                res.append((bev.detach().clone(), [m.indices.clone() for m in mid], f.grad.clone(),
                            [p.grad.clone() for p in enc.parameters()]))
                enc.zero_grad()
            out[presize] = res
        finally:
            se.prepare_strided_rulebooks = sp_mod_fn
    # frame 1: no hints (all 4 levels planned); frame 2: caps from the small frame overflow at the first level (nothing
    # planned); frame 3: hints from frame 2's true counts fit again
    assert planned_levels[0] == 4 and planned_levels[1] < 4 and planned_levels[2] == 4, planned_levels
    for (b1, i1, g1, p1), (b0, i0, g0, p0) in zip(out[True], out[False]):
        assert torch.equal(b1, b0) and torch.equal(g1, g0)
        assert all(torch.equal(a, b) for a, b in zip(i1, i0))
        assert all(torch.equal(a, b) for a, b in zip(p1, p0))


def test_presized_strided_rulebooks_are_identical(dev):
    """Counting all strided layers up front (one host read) yields the same rulebooks and the same BEV map, bit
    for bit, as the per-layer path; a level that overflows its cap falls back to the per-layer path."""
    from bevfusion_amd import spconv as sp
    from bevfusion_amd.sparse_encoder import BEVFusionSparseEncoder
    B, shape, n = 2, (96, 96, 41), 6000
    idx, feats = random_sparse(B, shape, n, 5, seed=21)
    torch.manual_seed(0)
    enc = BEVFusionSparseEncoder(in_channels=5, sparse_shape=list(shape), norm_cfg=dict(type="BN1d", eps=0.001, momentum=0.01),
                                 encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                                 encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, (1, 1, 0)), (0, 0)), block_type="basicblock",
                                 return_middle_feats=True).to(dev).eval()
    f, c = torch.from_numpy(feats).to(dev), torch.from_numpy(idx).to(dev)
    with torch.no_grad():
        enc.presize_rulebooks = True
        bev1, mid1 = enc(f, c, B)
        enc.presize_rulebooks = False
        bev0, mid0 = enc(f, c, B)
    assert torch.equal(bev0, bev1)
    for a, b in zip(mid0, mid1):
        assert torch.equal(a.indices, b.indices) and torch.equal(a.features, b.features)
    # the plan itself: 4 strided levels, N_out equal to what each layer produced
    specs = [(m.kernel_size, m.stride, m.padding, m.dilation) for m in enc.modules() if isinstance(m, sp.SparseConv3d)]
    plans = sp.prepare_strided_rulebooks(c, B, list(shape), specs)
    assert len(plans) == 4
    assert sorted(p.n_out for p in plans.values()) == sorted([m.indices.shape[0] for m in mid0[:3]] + [int((bev0.abs().sum(1) > 0).sum() * 0 + sp.build_sparse_rulebook(mid0[3].indices, B, mid0[3].spatial_shape, *specs[3]).out_indices.shape[0])])


def _wgrad_reference(x, g, pairs, n_rows):
    """dW[co][k][ci] = sum over rows n with pairs[k][n] >= 0 of g[n][co] * x[pairs[k][n]][ci], in fp64."""
    KV = pairs.shape[0]
    out = np.zeros((g.shape[1], KV, x.shape[1]))
    for k in range(KV):
        rows = np.nonzero(pairs[k, :n_rows] >= 0)[0]
        if len(rows):
            out[:, k, :] = g[rows].astype(np.float64).T @ x[pairs[k, rows]].astype(np.float64)
    return out


@pytest.mark.parametrize("cin,cout,kv,n_rows,use_perm", [
    (16, 16, 27, 1, False), (16, 16, 27, 63, False), (32, 32, 27, 65, True), (64, 64, 27, 4097, False),
    (64, 128, 27, 5000, True), (128, 64, 3, 4200, False), (128, 128, 27, 9001, False), (96, 40, 1, 700, False),
    (32, 64, 8, 20000, False), (16, 16, 27, 20011, False), (5, 16, 27, 3000, False)])
def test_wgrad_abi_geometries_vs_fp64(dev, cin, cout, kv, n_rows, use_perm):
    """bfhip_spconv_wgrad through the C ABI on pair tables the encoder never produces: single rows, row counts around the
    64-row unit and the 8-region threshold (64 units), offsets with no pair at all and one that pairs every row, a row
    permutation, rectangular channel counts; fp32 and bf16 feature storage."""
    from bevfusion_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(cin * 131 + cout * 7 + kv + n_rows)
    n_in = max(n_rows // 2, 1)
    ld = n_rows + rs.randint(0, 5)
    pairs = np.full((kv, ld), -1, np.int32)
    for k in range(kv):
        density = (0.0, 1.0, 0.05, 0.5)[k % 4] if kv > 1 else 0.6   # empty, full, sparse and half-filled offsets
        m = rs.rand(n_rows) < density
        pairs[k, :n_rows][m] = rs.randint(0, n_in, int(m.sum()))
    x = rs.randn(n_in, cin).astype(np.float32)
    g = rs.randn(n_rows, cout).astype(np.float32)
    perm = rs.permutation(n_rows).astype(np.int32) if use_perm else None
    want = _wgrad_reference(x, g, pairs, n_rows)
    tp, tperm = torch.from_numpy(pairs).to(dev), (torch.from_numpy(perm).to(dev) if use_perm else None)
    for io16 in (0, 1):
        if io16 and (cin % 4 or cout % 4):
            continue
        tx, tg = torch.from_numpy(x).to(dev), torch.from_numpy(g).to(dev)
        ref = want
        if io16:
            tx, tg = tx.to(torch.bfloat16), tg.to(torch.bfloat16)
            ref = _wgrad_reference(tx.float().cpu().numpy(), tg.float().cpu().numpy(), pairs, n_rows)
        dw = torch.full((cout, kv, cin), float("nan"), device=dev)
        wsb = lib.bfhip_spconv_wgrad_workspace_bytes(kv, cin, cout, n_rows)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = lib.bfhip_spconv_wgrad(_lib.ptr(tx), _lib.ptr(tg), _lib.ptr(tp), ld, kv, n_rows, cin, cout, _lib.ptr(tperm), _lib.ptr(dw),
                                    io16, _lib.ptr(ws), wsb, _lib.stream_of(tx))
        _lib.check(rc, "spconv_wgrad")
        got = dw.cpu().numpy()
        assert np.isfinite(got).all()
        assert rel_err(got, ref) < 2e-5, (io16,)
        if kv > 1:
            assert (got[:, 0, :] == 0).all()   # the offset without pairs
        # same inputs, same bits: the partial sums are combined in a fixed order
        dw2 = torch.empty_like(dw)
        lib.bfhip_spconv_wgrad(_lib.ptr(tx), _lib.ptr(tg), _lib.ptr(tp), ld, kv, n_rows, cin, cout, _lib.ptr(tperm), _lib.ptr(dw2),
                               io16, _lib.ptr(ws), wsb, _lib.stream_of(tx))
        assert torch.equal(dw, dw2)


def test_sort_rows_is_region_major_mask_sort(dev, sorted_rows):
    """bfhip_rulebook_sort_rows: row_mask bit k = pair present; perm = stable sort by (chunk of 4096 rows, mask)."""
    from bevfusion_amd.spconv import sort_rows
    rs = np.random.RandomState(4)
    for n, kv in ((1, 27), (777, 27), (5000, 27), (3000, 3), (4096, 27), (4097, 27), (70000, 27)):
        pairs = np.where(rs.rand(kv, n) < 0.4, rs.randint(0, n, (kv, n)), -1).astype(np.int32)
        mask, perm = sort_rows(torch.from_numpy(pairs).to(dev))
        want_mask = ((pairs >= 0).astype(np.int64) << np.arange(kv)[:, None]).sum(0)
        np.testing.assert_array_equal(mask.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, want_mask)
        key = ((np.arange(n, dtype=np.int64) // 4096) << kv) | want_mask
        np.testing.assert_array_equal(perm.cpu().numpy(), np.argsort(key, kind="stable"))


def test_static_capacity_lidar_branch_has_no_host_reads(dev):
    """after one exact forward has taught it the row capacities, the LiDAR branch (hard voxelization of every
    sample, compaction + mean, all rulebooks, 21 sparse convs with fused BN, dense BEV map) runs with EVERY row count on the
    device: torch.cuda.set_sync_debug_mode("error") turns any synchronising call into an exception.  Features and
    gradients equal the exact-size path (inactive rows contribute exact zeros; only the fp32 reduction trees differ)."""
    from bevfusion_amd.bevfusion import nuscenes_config
    from bevfusion_amd.registry import MODELS
    torch.manual_seed(0)
    model = MODELS.build(nuscenes_config(camera=False, lidar=True)).to(dev).train()
    enc = model.pts_middle_encoder
    pts = [torch.from_numpy(synthetic.lidar_sweep(30000, seed=70 + i)).to(dev) for i in range(3)]
    inp = {"points": pts}
    model.static_lidar = False
    ref = model.extract_pts_feat(inp)                      # exact sizes (host reads): learns the capacities
    ref.square().mean().backward()
    g_ref = [p.grad.clone() for p in enc.parameters()]
    rm_ref = enc.conv_out[1].running_mean.clone()
    assert enc.static_caps is not None and model._voxel_cap is not None
    for p in enc.parameters():
        p.grad = None
    model.static_lidar = True
    model.extract_pts_feat(inp)                            # warm-up of the static path (pinned buffers, workspaces)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = model.extract_pts_feat(inp)
        loss = out.square().mean()
        loss.backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert enc._monitor.pending or enc._monitor.pinned is not None
    assert out.shape == ref.shape
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 1e-5
    for a, b in zip([p.grad for p in enc.parameters()], g_ref):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-4
    # running statistics divide by the TRUE row counts (three updates here vs one in the reference run: compare one step)
    # frames three times larger overflow the learnt capacities: flagged and grown without a stall, exact again afterwards
    big = {"points": [torch.from_numpy(synthetic.lidar_sweep(120000, seed=90 + i)).to(dev) for i in range(3)]}
    with pytest.warns(UserWarning):
        model.extract_pts_feat(big)
        torch.cuda.synchronize()
        model.extract_pts_feat(big)                        # polls the counts of the overflowing forward -> warns, grows
    torch.cuda.synchronize()
    model.extract_pts_feat(big)
    torch.cuda.synchronize()
    got = model.extract_pts_feat(big)
    model.static_lidar = False
    want = model.extract_pts_feat(big)
    assert rel_err(got.detach().cpu().numpy(), want.detach().cpu().numpy()) < 1e-5
    del rm_ref


def test_reference_encoder_test_with_duplicate_coordinates(dev, sorted_rows, monkeypatch):
    """The reference's only test at the spconv boundary (tests/test_models/test_middle_encoders/test_sparse_encoders.py:8-28):
    207 842 rows whose (b, z, y, x) are all randint(0, 4) -- 256 distinct cells, ~800 duplicates each -- through the
    basicblock encoder; it asserts the output shape [4, 256, 128, 128] only (spconv's values there depend on which duplicate
    its hash insert kept).  Here: same shape in this fork's (x, y, z) order, every gather operand in range (validator on
    every launch), finite, and bit-identical across two runs -- duplicate rows all stay rows, a neighbour lookup resolves to
    the lowest duplicate (SubM) / the highest one per (offset, output) slot (strided), include/bevfusion_hip.h."""
    from bevfusion_amd import spconv
    monkeypatch.setattr(spconv, "VALIDATE", True)
    torch.manual_seed(0)
    enc = BEVFusionSparseEncoder(in_channels=5, sparse_shape=[1024, 1024, 40], order=("conv", "norm", "act"),
                                 encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                                 encoder_paddings=((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
                                 block_type="basicblock").to(dev)
    import copy
    enc2 = copy.deepcopy(enc)                                           # same weights, fresh BatchNorm statistics
    g = torch.Generator().manual_seed(1)
    voxel_features = torch.rand([207842, 5], generator=g).to(dev)
    coors = torch.randint(0, 4, [207842, 4], generator=g).to(dev)      # int64, as the reference's test hands it over
    ret = enc(voxel_features, coors, 4)
    assert ret.shape == torch.Size([4, 256, 128, 128])
    assert torch.isfinite(ret).all()
    ret.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in enc.parameters())
    g1 = [p.grad.clone() for p in enc.parameters()]
    for p in enc.parameters():
        p.grad = None
    ret2 = enc2(voxel_features, coors, 4)
    assert torch.equal(ret, ret2)                                       # deterministic although 800 rows race per cell
    ret2.square().mean().backward()
    assert all(torch.equal(a, p.grad) for a, p in zip(g1, enc2.parameters()))


def test_duplicate_coordinates_semantics_tiny(dev):
    """Duplicate rows, explicit expectation: SubM neighbour lookups resolve to the LOWEST row of a cell, a strided output slot
    takes the HIGHEST input row; every duplicate keeps its backward entry."""
    idx = np.array([[0, 2, 2, 2], [0, 2, 2, 3], [0, 2, 2, 2], [0, 2, 2, 3], [0, 2, 2, 2]], np.int32)
    t = torch.from_numpy(idx).to(dev)
    data = build_subm_rulebook(t, 1, [8, 8, 8], [3, 3, 3], [1, 1, 1])
    pf = data.pair_fwd.cpu().numpy()
    assert (pf[13] == np.arange(5)).all()                      # centre offset: every row is its own pair
    assert (pf[14] == np.array([1, -1, 1, -1, 1])).all()       # z + 1 neighbour of cell (2,2,2) = lowest row at (2,2,3)
    assert (pf[12] == np.array([-1, 0, -1, 0, -1])).all()      # z - 1 neighbour of cell (2,2,3) = lowest row at (2,2,2)
    d2 = build_sparse_rulebook(t, 1, [8, 8, 8], [1, 1, 1], [1, 1, 1], [0, 0, 0], [1, 1, 1])
    assert d2.out_indices.cpu().numpy().tolist() == [[0, 2, 2, 2], [0, 2, 2, 3]]
    assert d2.pair_fwd.cpu().numpy().tolist() == [[4, 3]]      # highest duplicate per output
    assert d2.pair_bwd.cpu().numpy().tolist() == [[0, 1, 0, 1, 0]]


def test_first_forward_at_the_round2_fault_geometry_is_validated(dev, sorted_rows, monkeypatch):
    """Round 2 lost a GPU to HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in spconv_gemm_lds_kernel<4,1> (fp32 features, row
    sorting on, the FIRST forward of a process at batch 4 -- DESIGN.md section 6).  That kernel trusts perm / row_mask / pairs;
    this runs the same geometry (fresh model, no hints, fp32, no autocast, 4 x 40 k points, forward + backward) with
    bfhip_rulebook_validate in front of every one of the 41 gather launches, and checks that the sorted row order changes no
    bit of the forward."""
    from bevfusion_amd import spconv
    from bevfusion_amd.bevfusion import nuscenes_config
    from bevfusion_amd.registry import MODELS
    monkeypatch.setattr(spconv, "VALIDATE", True)
    torch.manual_seed(0)
    model = MODELS.build(nuscenes_config(camera=False, lidar=True)).to(dev).train()
    inp = {"points": [torch.from_numpy(synthetic.lidar_sweep(40000, seed=1000 + i)).to(dev) for i in range(4)]}
    with torch.no_grad():
        first = model.extract_pts_feat(inp)                  # what bench.collect_work does first: fp32, no autocast
    out = model.extract_pts_feat(inp)
    out.square().mean().backward()
    enc = model.pts_middle_encoder
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in enc.parameters())
    monkeypatch.setattr(spconv, "SORT_ROWS", False)
    model.eval()                                             # same BatchNorm statistics for both orders
    with torch.no_grad():
        a = model.extract_pts_feat(inp)
        monkeypatch.setattr(spconv, "SORT_ROWS", True)
        b = model.extract_pts_feat(inp)
    assert torch.equal(a, b) and torch.isfinite(first).all()


def test_validator_flags_bad_operands(dev):
    from bevfusion_amd.spconv import validate_rulebook
    pairs = torch.tensor([[0, 1, 2, -1], [3, -1, 7, 0]], dtype=torch.int32, device=dev)
    perm = torch.tensor([3, 2, 1, 0], dtype=torch.int32, device=dev)
    mask = torch.tensor([3, 1, 3, 2], dtype=torch.int32, device=dev)
    assert validate_rulebook(pairs, 4, 8, perm, mask) == [0, 0, 0, 0]
    assert validate_rulebook(pairs, 4, 7, perm, mask) == [1, 0, 0, 0]          # entry 7 with 7 source rows
    bad_perm = torch.tensor([3, 2, 2, 9], dtype=torch.int32, device=dev)
    assert validate_rulebook(pairs, 4, 8, bad_perm, mask) == [0, 1, 3, 0]      # 9 out of range; rows 0, 1 missing, 2 twice
    bad_mask = torch.tensor([3, 1, 3, 0], dtype=torch.int32, device=dev)
    assert validate_rulebook(pairs, 4, 8, perm, bad_mask) == [0, 0, 0, 1]


def test_static_capacity_overflow_is_in_bounds_and_poisons_the_step(dev, monkeypatch):
    """A frame that overflows the learnt row capacities (static capacity mode): (i) every gather operand of forward AND
    backward stays in range (rows beyond a capacity are dropped from the rulebooks on both sides -- round 2 kept the
    unclamped output row in pair_bwd, an out-of-bounds read in the strided data gradient); (ii) gradients are finite;
    (iii) the device-side status is raised in the same forward (no host read), BEVFusion.loss turns it into NaN losses and
    the optimizer skips the step: parameters and moments do not change."""
    from bevfusion_amd import spconv
    from bevfusion_amd.amp import skip_nonfinite_step
    from bevfusion_amd.bevfusion import nuscenes_config
    from bevfusion_amd.registry import MODELS
    torch.manual_seed(0)
    model = MODELS.build(nuscenes_config(camera=False, lidar=True)).to(dev).train()
    enc = model.pts_middle_encoder
    small = {"points": [torch.from_numpy(synthetic.lidar_sweep(20000, seed=70 + i)).to(dev) for i in range(2)]}
    big = {"points": [torch.from_numpy(synthetic.lidar_sweep(120000, seed=90 + i)).to(dev) for i in range(2)]}
    model.static_lidar = False
    model.extract_pts_feat(small)                            # learns the capacities from the small frames
    model.static_lidar = True
    ok = model.extract_pts_feat(small)
    assert model.capacity_status() is not None and not bool(model.capacity_status())
    # keep the capacities where they are for this test: the stall-free monitors would grow them one forward later
    monkeypatch.setattr(enc._monitor, "poll", lambda: None)
    monkeypatch.setattr(model._voxel_monitor, "poll", lambda: None)
    monkeypatch.setattr(spconv, "VALIDATE", True)
    out = model.extract_pts_feat(big)                        # overflows voxel and strided-layer capacities
    assert bool(model.capacity_status())
    out.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in enc.parameters())
    assert torch.isfinite(out).all() and out.shape == ok.shape
    # the poisoned loss: NaN -> non-finite gradient norm -> the fused AdamW leaves everything untouched
    params = [p for p in enc.parameters()]
    before = [p.detach().clone() for p in params]
    opt = torch.optim.AdamW(params, lr=1e-2, fused=True)
    for p in params:
        p.grad = None
    out2 = model.extract_pts_feat(big)
    poison = torch.where(model.capacity_status(), float("nan"), 1.0)   # what BEVFusion.loss multiplies into every loss entry
    (out2.square().mean() * poison).backward()
    skip_nonfinite_step(opt, torch.nn.utils.clip_grad_norm_(params, 35.0, foreach=True))
    opt.step()
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, params))
    for p in params:
        p.grad = None
    model.extract_pts_feat(small).square().mean().backward()   # a clean frame afterwards trains
    assert not bool(model.capacity_status())
    skip_nonfinite_step(opt, torch.nn.utils.clip_grad_norm_(params, 35.0, foreach=True))
    opt.step()
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, params))
