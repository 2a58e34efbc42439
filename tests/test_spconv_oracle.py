"""CPU: the sparse-conv restatement (oracle/spconv_oracle.c) against torch.nn.functional.conv3d on the
densified tensor -- the independent numeric oracle for a path whose reference arithmetic (spconv 2.x) is not
available (parity unpinned by fixtures, see the file header)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle


def random_sparse(B, shape, n, cin, seed):
    rng = np.random.default_rng(seed)
    cells = rng.choice(B * shape[0] * shape[1] * shape[2], n, replace=False)
    rng.shuffle(cells)
    z = cells % shape[2]; cells = cells // shape[2]
    y = cells % shape[1]; cells = cells // shape[1]
    x = cells % shape[0]; b = cells // shape[0]
    idx = np.stack([b, x, y, z], 1).astype(np.int32)
    feats = rng.standard_normal((n, cin)).astype(np.float32)
    return idx, feats


def densify(idx, feats, B, shape):
    d = np.zeros((B, feats.shape[1], *shape), np.float32)
    d[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = feats
    return d


def torch_weight(w):  # (Cout,k0,k1,k2,Cin) -> (Cout,Cin,k0,k1,k2)
    return torch.from_numpy(w).permute(0, 4, 1, 2, 3).contiguous()


def test_conv_out_shape_chain():
    """The shape chain the reference records (BF/sparse_encoder.py:132,148): 1440->720->360->180, 41->21->11->5->2."""
    s = [1440, 1440, 41]
    s = list(oracle.conv_out_shape(s, 3, 2, 1)); assert s == [720, 720, 21]
    s = list(oracle.conv_out_shape(s, 3, 2, 1)); assert s == [360, 360, 11]
    s = list(oracle.conv_out_shape(s, 3, 2, [1, 1, 0])); assert s == [180, 180, 5]
    s = list(oracle.conv_out_shape(s, [1, 1, 3], [1, 1, 2], 0)); assert s == [180, 180, 2]


@pytest.mark.parametrize("ksize", [3, (1, 1, 3)])
def test_subm_matches_dense_conv3d(ksize):
    B, shape, n, cin, cout = 2, (9, 8, 7), 300, 5, 6
    idx, feats = random_sparse(B, shape, n, cin, 1)
    ks = [ksize] * 3 if isinstance(ksize, int) else list(ksize)
    w = np.random.default_rng(2).standard_normal((cout, *ks, cin)).astype(np.float32)
    pair = oracle.rulebook_subm(idx, shape, ks)
    out = oracle.spconv_fwd(feats, w, pair)
    dense = F.conv3d(torch.from_numpy(densify(idx, feats, B, shape)).double(), torch_weight(w).double(),
                     padding=[k // 2 for k in ks])
    want = dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]].numpy()  # SubM = dense conv sampled at the input sites
    assert np.allclose(out, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("ksize,stride,padding", [(3, 2, 1), (3, 2, (1, 1, 0)), ((1, 1, 3), (1, 1, 2), 0)])
def test_strided_matches_dense_conv3d(ksize, stride, padding):
    B, shape, n, cin, cout = 2, (10, 9, 11), 250, 4, 5
    idx, feats = random_sparse(B, shape, n, cin, 3)
    ks = [ksize] * 3 if isinstance(ksize, int) else list(ksize)
    st = [stride] * 3 if isinstance(stride, int) else list(stride)
    pd = [padding] * 3 if isinstance(padding, int) else list(padding)
    w = np.random.default_rng(4).standard_normal((cout, *ks, cin)).astype(np.float32)
    out_idx, pf, pb, out_shape = oracle.rulebook_sparse(idx, shape, ks, st, pd)
    out = oracle.spconv_fwd(feats, w, pf)
    dense = F.conv3d(torch.from_numpy(densify(idx, feats, B, shape)).double(), torch_weight(w).double(), stride=st, padding=pd)
    assert list(dense.shape[2:]) == list(out_shape)
    got = densify(out_idx, out, B, tuple(out_shape))
    # active output set = sites reachable from an input; everything else is exactly zero in the dense conv too
    assert np.allclose(got, dense.numpy(), rtol=1e-5, atol=1e-5)
    # canonical order: ascending linear index
    lin = ((out_idx[:, 0].astype(np.int64) * out_shape[0] + out_idx[:, 1]) * out_shape[1] + out_idx[:, 2]) * out_shape[2] + out_idx[:, 3]
    assert (np.diff(lin) > 0).all()
    # pair_bwd is the inverse relation of pair_fwd
    kv, n_out = pf.shape
    for k in range(kv):
        o = np.flatnonzero(pf[k] >= 0)
        assert np.array_equal(pb[k][pf[k][o]], o)
    assert (pf >= 0).sum() == (pb >= 0).sum()


def test_backward_matches_torch_autograd():
    B, shape, n, cin, cout = 1, (8, 8, 6), 200, 4, 3
    idx, feats = random_sparse(B, shape, n, cin, 5)
    w = np.random.default_rng(6).standard_normal((cout, 3, 3, 3, cin)).astype(np.float32)
    pair = oracle.rulebook_subm(idx, shape, 3)
    d_out = np.random.default_rng(7).standard_normal((n, cout)).astype(np.float32)
    d_in, d_w = oracle.spconv_bwd(feats, w, d_out, pair)
    ft = torch.from_numpy(feats).double().requires_grad_(True)
    wt = torch.from_numpy(w).double().requires_grad_(True)
    lin = ((torch.from_numpy(idx[:, 0]).long() * shape[0] + idx[:, 1]) * shape[1] + idx[:, 2]) * shape[2] + idx[:, 3]
    flat = torch.zeros(B * shape[0] * shape[1] * shape[2], cin, dtype=torch.float64).index_add(0, lin, ft)
    dense_in = flat.view(B, *shape, cin).permute(0, 4, 1, 2, 3)
    dense = F.conv3d(dense_in, wt.permute(0, 4, 1, 2, 3), padding=1)
    sel = dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]
    (sel * torch.from_numpy(d_out).double()).sum().backward()
    assert np.allclose(d_in, ft.grad.numpy(), rtol=1e-5, atol=1e-5)
    assert np.allclose(d_w, wt.grad.numpy(), rtol=1e-5, atol=1e-5)


def test_sparse_to_bev_matches_reference_permute():
    B, (X, Y, Z), n, c = 2, (6, 5, 2), 30, 4
    idx, feats = random_sparse(B, (X, Y, Z), n, c, 8)
    got = oracle.sparse_to_bev(feats, idx, B, X, Y, Z)
    dense = torch.from_numpy(densify(idx, feats, B, (X, Y, Z)))          # [B, C, X, Y, Z]
    want = dense.permute(0, 1, 4, 2, 3).contiguous().view(B, c * Z, X, Y)  # BF/sparse_encoder.py:149-151
    assert np.array_equal(got, want.numpy())
