"""GPU: split-K X^T Y (csrc/xty.hip), linear_rows and the spelled-out multi-head attention against torch."""
import pytest
import torch
import torch.nn.functional as F

import bevfusion_amd  # noqa: F401
from bevfusion_amd import linear_rows as lr

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("K,M,N", [(129600, 128, 128), (50001, 128, 2), (9000, 2, 128), (777, 72, 200), (64, 4, 4), (100000, 256, 64)])
def test_xty_matches_fp64(dev, dtype, K, M, N):
    torch.manual_seed(0)
    x = torch.randn(K, M, device=dev).to(dtype)
    y = torch.randn(K, N, device=dev).to(dtype)
    out = lr.xty(x, y)
    ref = x.double().t() @ y.double()
    assert out.dtype == torch.float32 and out.shape == (M, N)
    assert rel(out, ref) < 2e-5  # fp32 products / sums of exactly representable inputs


def test_xty_is_deterministic(dev):
    x = torch.randn(129600, 128, device=dev).to(torch.bfloat16)
    y = torch.randn(129600, 128, device=dev).to(torch.bfloat16)
    assert torch.equal(lr.xty(x, y), lr.xty(x, y))


@pytest.mark.parametrize("amp", [False, True])
def test_linear_rows_matches_linear(dev, amp):
    torch.manual_seed(1)
    K = 20000
    x = torch.randn(K, 128, device=dev)
    w = (torch.randn(96, 128, device=dev) * 0.1).requires_grad_(True)
    b = torch.randn(96, device=dev).requires_grad_(True)
    g = torch.randn(K, 96, device=dev)
    xa = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        y = lr.linear_rows(xa, w, b)
        (y.float() * g).sum().backward()
    wr, br, xr = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True), x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        yr = F.linear(xr, wr, br)
        (yr.float() * g).sum().backward()
    tol = 2e-2 if amp else 1e-4
    assert y.dtype == yr.dtype and rel(y.float(), yr.float()) < tol
    assert w.grad.dtype == torch.float32 and rel(w.grad, wr.grad) < tol
    assert rel(b.grad, br.grad) < tol and rel(xa.grad, xr.grad) < tol


def test_spelled_out_mha_equals_nn_multiheadattention(dev):
    from bevfusion_amd.dense_modules import _MHA
    torch.manual_seed(2)
    m = _MHA(128, 8, dropout=0.0).to(dev).train()
    q = torch.randn(2, 50, 128, device=dev, requires_grad=True)
    k = torch.randn(2, 9000, 128, device=dev, requires_grad=True)  # >= MIN_ROWS rows: the split-K path
    qp, kp = torch.randn(2, 50, 128, device=dev), torch.randn(2, 9000, 128, device=dev)
    out = m(q, k, k + kp, qp, kp)
    ref = q + m.attn(q + qp, k + kp, k + kp, need_weights=False)[0]
    assert rel(out, ref) < 1e-4
    gq, gk = torch.autograd.grad(out.square().sum(), (q, k), retain_graph=True)
    rq, rk = torch.autograd.grad(ref.square().sum(), (q, k))
    assert rel(gq, rq) < 1e-3 and rel(gk, rk) < 1e-3
    gw = torch.autograd.grad(out.square().sum(), m.attn.in_proj_weight)[0]
    out2 = q + m.attn(q + qp, k + kp, k + kp, need_weights=False)[0]
    rw = torch.autograd.grad(out2.square().sum(), m.attn.in_proj_weight)[0]
    assert rel(gw, rw) < 1e-3


def test_xty_edge_sizes(dev):
    for K, M, N in ((1, 4, 4), (3, 1, 1), (65, 130, 3), (17, 64, 64)):
        x = torch.randn(K, M, device=dev)
        y = torch.randn(K, N, device=dev)
        ref = x.double().t() @ y.double()
        assert rel(lr.xty(x, y), ref) < 2e-5, (K, M, N)
    # a zero-row gradient is a zero matrix through the public wrapper (plain F.linear path below MIN_ROWS)
    w = torch.randn(8, 4, device=dev, requires_grad=True)
    out = lr.linear_rows(torch.zeros(0, 4, device=dev), w)
    out.sum().backward()
    assert out.shape == (0, 8) and float(w.grad.abs().sum()) == 0.0
