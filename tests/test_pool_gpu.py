"""GPU parity of csrc/pool.hip (3x3 stride-2 pad-1 max pooling of channels-last bf16 maps, the ResNet-50 stem's pooling)
through the C ABI / the module against torch's own max_pool2d on the CPU.

Forward: bit-exact (a maximum of bf16 values is one of them), including ties (first maximum in (kh, kw) scan order wins,
checked through the backward: the gradient must land on the same element torch picks) and NaN propagation.  Backward: every
input element receives the sum of at most four bf16 gradients, accumulated in fp32 and rounded once: compared with torch's
fp32 CPU result rounded to bf16 (<= 1 bf16 ulp; exact on the integer-valued case)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import bevfusion_amd  # noqa: F401
from bevfusion_amd.dense_modules import MaxPool3x3s2

pytestmark = pytest.mark.gpu


def _run(dev, x, gy):
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, stride=2, padding=1)
    ref.backward(gy)
    pool = MaxPool3x3s2()
    xg = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = pool(xg)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last) and y.shape == ref.shape
    y.backward(gy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    return ref.detach(), xr.grad, y.float().cpu(), xg.grad.float().cpu()


@pytest.mark.parametrize("shape", [(2, 64, 32, 44), (3, 16, 17, 23), (1, 8, 1, 1), (2, 24, 2, 5), (1, 8, 128, 352)],
                         ids=lambda s: "x".join(map(str, s)))
def test_matches_torch(dev, shape):
    rng = np.random.default_rng(sum(shape))
    x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(torch.bfloat16).float()
    OH, OW = (shape[2] - 1) // 2 + 1, (shape[3] - 1) // 2 + 1
    gy = torch.from_numpy(rng.standard_normal(shape[:2] + (OH, OW)).astype(np.float32)).to(torch.bfloat16).float()
    ref, gref, y, gx = _run(dev, x, gy)
    assert torch.equal(y, ref)
    assert torch.equal(gx, gref.to(torch.bfloat16).float())   # fp32 sum of <= 4 bf16 values, rounded once


def test_ties_keep_the_first_maximum_and_integer_gradients_are_exact(dev):
    """Few distinct values -> most windows hold ties: the gradient lands where torch's scan (kh, kw ascending, strict >) puts it."""
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.integers(-2, 3, (2, 32, 19, 26)).astype(np.float32))
    gy = torch.from_numpy(rng.integers(-3, 4, (2, 32, 10, 13)).astype(np.float32))
    ref, gref, y, gx = _run(dev, x, gy)
    assert torch.equal(y, ref) and torch.equal(gx, gref)


def test_nan_and_infinities_propagate_like_torch(dev):
    x = torch.zeros((1, 8, 6, 6))
    x[0, 0, 2, 2] = float("nan")
    x[0, 1] = float("-inf")
    x[0, 2, 3, 3] = float("inf")
    gy = torch.ones((1, 8, 3, 3))
    ref, gref, y, gx = _run(dev, x, gy)
    assert torch.equal(torch.isnan(y), torch.isnan(ref))
    assert torch.equal(torch.nan_to_num(y, nan=7.0), torch.nan_to_num(ref, nan=7.0))
    assert torch.equal(gx, gref)


def test_falls_back_outside_its_domain(dev):
    pool = MaxPool3x3s2()
    x = torch.randn(2, 12, 9, 9, device=dev)                       # fp32, channels not a multiple of 8
    assert torch.equal(pool(x), F.max_pool2d(x, 3, stride=2, padding=1))
    xc = torch.randn(1, 8, 9, 9)                                   # CPU
    assert torch.equal(pool(xc), F.max_pool2d(xc, 3, stride=2, padding=1))
