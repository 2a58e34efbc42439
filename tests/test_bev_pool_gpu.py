"""GPU parity: csrc/bev_pool.hip through the C ABI vs the oracle (same seeded inputs)."""
import numpy as np
import pytest
import torch

import bevfusion_amd  # noqa: F401
import oracle
from bevfusion_amd.ops import bev_pool, bev_pool_ext
from bevfusion_amd.ops.bev_pool.bev_pool import intervals_from_ranks

from util import make_bev_pool_case, rel_err

pytestmark = pytest.mark.gpu


def _run_fwd(dev, x, geom, starts, lengths, b, d, h, w):
    t = lambda a: torch.from_numpy(a).to(dev)
    return bev_pool_ext.bev_pool_forward(t(x), t(geom), t(lengths), t(starts), b, d, h, w).cpu().numpy()


@pytest.mark.parametrize("c", [80, 16, 4, 64, 256])
@pytest.mark.parametrize("long_tail", [False, True])
def test_fwd_bit_exact_vs_oracle(dev, c, long_tail):
    """Same summation order as the reference kernel (bev_pool_cuda.cu:38-40) -> bit-identical fp32."""
    n, b, d, h, w = 60000, 2, 1, 40, 36
    x, geom, ranks = make_bev_pool_case(n, c, b, d, h, w, seed=c + long_tail, long_tail=long_tail)
    starts, lengths = oracle.intervals_from_ranks(ranks)
    want = oracle.bev_pool_fwd(x, geom, starts, lengths, b, d, h, w)
    got = _run_fwd(dev, x, geom, starts, lengths, b, d, h, w)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("c", [5, 7, 81])
def test_fwd_generic_channel_counts(dev, c):
    n, b, d, h, w = 20000, 1, 2, 16, 20
    x, geom, ranks = make_bev_pool_case(n, c, b, d, h, w, seed=c)
    starts, lengths = oracle.intervals_from_ranks(ranks)
    want = oracle.bev_pool_fwd(x, geom, starts, lengths, b, d, h, w)
    got = _run_fwd(dev, x, geom, starts, lengths, b, d, h, w)
    assert np.array_equal(got, want)


def test_bwd_bit_exact_vs_oracle(dev):
    n, c, b, d, h, w = 50000, 80, 2, 1, 30, 30
    x, geom, ranks = make_bev_pool_case(n, c, b, d, h, w, seed=2, long_tail=True)
    starts, lengths = oracle.intervals_from_ranks(ranks)
    og = np.random.default_rng(1).standard_normal((b, d, h, w, c)).astype(np.float32)
    want = oracle.bev_pool_bwd(og, geom, starts, lengths, n)
    t = lambda a: torch.from_numpy(a).to(dev)
    for cover in (False, True):
        got = bev_pool_ext.bev_pool_backward(t(og), t(geom), t(lengths), t(starts), b, d, h, w, _cover_all=cover)
        assert np.array_equal(got.cpu().numpy(), want)


def test_bwd_partial_cover_zero_fills(dev):
    """Intervals that do not cover every row: uncovered rows get zero gradient (bev_pool.cpp:76-78)."""
    n, c, b, d, h, w = 1000, 8, 1, 1, 8, 8
    x, geom, ranks = make_bev_pool_case(n, c, b, d, h, w, seed=4)
    starts, lengths = oracle.intervals_from_ranks(ranks)
    starts, lengths = starts[::2].copy(), lengths[::2].copy()
    og = np.random.default_rng(1).standard_normal((b, d, h, w, c)).astype(np.float32)
    want = oracle.bev_pool_bwd(og, geom, starts, lengths, n)
    t = lambda a: torch.from_numpy(a).to(dev)
    got = bev_pool_ext.bev_pool_backward(t(og), t(geom), t(lengths), t(starts), b, d, h, w)
    assert np.array_equal(got.cpu().numpy(), want)


def test_empty_and_single(dev):
    c, b, d, h, w = 80, 1, 1, 4, 4
    e = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    out = bev_pool_ext.bev_pool_forward(e((0, c), torch.float32), e((0, 4), torch.int32), e((0,), torch.int32),
                                        e((0,), torch.int32), b, d, h, w)
    assert out.shape == (b, d, h, w, c) and not out.any()
    x = torch.arange(c, dtype=torch.float32, device=dev).view(1, c)
    geom = torch.tensor([[3, 2, 0, 0]], dtype=torch.int32, device=dev)
    one = torch.tensor([1], dtype=torch.int32, device=dev)
    zero = torch.tensor([0], dtype=torch.int32, device=dev)
    out = bev_pool_ext.bev_pool_forward(x, geom, one, zero, b, d, h, w)
    assert torch.equal(out[0, 0, 3, 2], x[0]) and out.sum() == x.sum()


def test_wrong_dtype_raises(dev):
    x = torch.zeros(4, 8, device=dev, dtype=torch.float64)
    g = torch.zeros(4, 4, device=dev, dtype=torch.int32)
    i = torch.zeros(1, device=dev, dtype=torch.int32)
    with pytest.raises(RuntimeError):
        bev_pool_ext.bev_pool_forward(x, g, i, i, 1, 1, 2, 2)
    with pytest.raises(RuntimeError):
        bev_pool_ext.bev_pool_forward(x.float().cpu(), g.cpu(), i.cpu(), i.cpu(), 1, 1, 2, 2)


def test_python_op_and_autograd(dev):
    """bev_pool(feats, coords, ranks, B, D, H, W, is_training): output layout [B,C,D,H,W] and the
    training path's gradient (reference: bev_pool.py:146-172, 43-90)."""
    n, c, b, d, h, w = 30000, 80, 2, 1, 24, 24
    x, geom, ranks = make_bev_pool_case(n, c, b, d, h, w, seed=8)
    starts, lengths = oracle.intervals_from_ranks(ranks)
    want = np.transpose(oracle.bev_pool_fwd(x, geom, starts, lengths, b, d, h, w), (0, 4, 1, 2, 3))
    feats = torch.from_numpy(x).to(dev).requires_grad_(True)
    coords = torch.from_numpy(geom).to(dev).long()
    rk = torch.from_numpy(ranks).to(dev)
    out = bev_pool(feats, coords, rk, b, d, h, w, True)
    assert out.shape == (b, c, d, h, w)
    assert np.array_equal(out.detach().cpu().numpy(), want)
    og = torch.randn_like(out)
    out.backward(og)
    og_l = og.permute(0, 2, 3, 4, 1).contiguous().cpu().numpy()
    assert np.array_equal(feats.grad.cpu().numpy(), oracle.bev_pool_bwd(og_l, geom, starts, lengths, n))
    out_eval = bev_pool(feats.detach(), coords, rk, torch.tensor(b), torch.tensor(d), torch.tensor(h), torch.tensor(w),
                        False)
    assert torch.equal(out_eval, out.detach())
    s2, l2 = intervals_from_ranks(rk)
    assert np.array_equal(s2.cpu().numpy(), starts) and np.array_equal(l2.cpu().numpy(), lengths)


def test_full_size_properties(dev):
    """BASELINE-size case (n ~ 1.83 M rows, C = 80, 360x360): size-independent properties --
    linearity, sum conservation against an fp64 column sum, and a strided spot check vs the oracle."""
    from bevfusion_amd import synthetic
    n, c, b, d, h, w = 1830000, 80, 1, 1, 360, 360
    rng = np.random.default_rng(0)
    cells = np.sort(rng.integers(0, h * w, n))
    geom = np.stack([cells // w, cells % w, np.zeros_like(cells), np.zeros_like(cells)], 1).astype(np.int32)
    starts, lengths = oracle.intervals_from_ranks(cells.astype(np.int64))
    t = lambda a: torch.from_numpy(a).to(dev)
    x1 = torch.randn(n, c, device=dev)
    x2 = torch.randn(n, c, device=dev)
    G, S, L = t(geom), t(starts), t(lengths)
    o1 = bev_pool_ext.bev_pool_forward(x1, G, L, S, b, d, h, w)
    o2 = bev_pool_ext.bev_pool_forward(x2, G, L, S, b, d, h, w)
    o12 = bev_pool_ext.bev_pool_forward(x1 + x2, G, L, S, b, d, h, w)
    assert rel_err((o1 + o2).cpu().numpy(), o12.cpu().numpy()) < 1e-5
    assert torch.allclose(o1.double().sum((0, 1, 2, 3)), x1.double().sum(0), rtol=1e-6, atol=1e-3)
    sel = slice(0, 200)  # first 200 intervals against the oracle, bit-exact
    nrows = int(starts[200])
    want = oracle.bev_pool_fwd(x1[:nrows].cpu().numpy(), geom[:nrows], starts[sel], lengths[sel], b, d, h, w)
    g0 = geom[starts[sel]]
    assert np.array_equal(o1.cpu().numpy()[0, 0, g0[:, 0], g0[:, 1]], want[0, 0, g0[:, 0], g0[:, 1]])
    # backward: every row receives its cell's gradient -> idempotent under fwd(bwd(.)) scaling by length
    xg = bev_pool_ext.bev_pool_backward(o1, G, L, S, b, d, h, w, _cover_all=True)
    assert torch.equal(xg[S.long()], o1[0, 0, G[S.long(), 0].long(), G[S.long(), 1].long()])


@pytest.mark.parametrize("name", ["int", "flt", "odd"])
def test_vs_reference_quickcumsum_golden(dev, golden_bev, name):
    """HIP op through the C ABI against the outputs of the REFERENCE's own QuickCumsum (bev_pool.py:7-34, fp64) and its
    interval construction (:48-54): tests/golden/bev_pool_ref.npz."""
    from bevfusion_amd import synthetic
    from util import sha
    seed, n, C, B, D, H, W, integer = [int(v) for v in golden_bev[name + "_cfg"]]
    x, geom, ranks = synthetic.bev_pool_case(seed, n, C, B, D, H, W, bool(integer))
    assert sha(x) == str(golden_bev[name + "_x_sha"]) and sha(ranks) == str(golden_bev[name + "_ranks_sha"])
    t = lambda a: torch.from_numpy(a).to(dev)
    s_dev, l_dev = intervals_from_ranks(t(ranks))
    assert np.array_equal(s_dev.cpu().numpy(), golden_bev[name + "_starts"])
    assert np.array_equal(l_dev.cpu().numpy(), golden_bev[name + "_lengths"])
    out = bev_pool_ext.bev_pool_forward(t(x), t(geom), l_dev, s_dev, B, D, H, W).cpu().numpy()
    rg = golden_bev[name + "_row_geom"]
    rows = out[rg[:, 3], rg[:, 2], rg[:, 0], rg[:, 1]]
    want = golden_bev[name + "_rows"]
    if integer:
        assert np.array_equal(rows, want) and sha(out) == str(golden_bev[name + "_dense_sha_f32"])
    else:
        lengths = golden_bev[name + "_lengths"]
        assert np.abs(rows - want).max() <= 1e-6 * np.abs(want).max() * np.sqrt(lengths.max())
    og = np.zeros((B, D, H, W, C), np.float32)
    og[rg[:, 3], rg[:, 2], rg[:, 0], rg[:, 1]] = golden_bev[name + "_grad_rows"]
    xg = bev_pool_ext.bev_pool_backward(t(og), t(geom), l_dev, s_dev, B, D, H, W).cpu().numpy()
    assert sha(xg) == str(golden_bev[name + "_xgrad_sha_f32"])
    # python op (autograd path) too
    feats = t(x).requires_grad_(True)
    o = bev_pool(feats, t(geom).long(), t(ranks), B, D, H, W, True)
    assert np.array_equal(o.detach().permute(0, 2, 3, 4, 1).cpu().numpy(), out)
