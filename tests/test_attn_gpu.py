"""GPU: split-key cross attention (csrc/attn.hip) against torch's scaled_dot_product_attention maths in fp32 on the same bf16
inputs (tolerance 2e-2 of the output scale: bf16 probabilities feed the second MFMA, as in any fused attention), with and
without dropout (the kernels' own keep mask is read back and applied to the reference)."""
import math

import pytest
import torch

import bevfusion_amd  # noqa: F401
from bevfusion_amd import attention as at

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))


def reference(q, k, v, H, mask=None, p=0.0):
    B, Lq, E = q.shape
    d = E // H
    hq, hk, hv = [t.float().view(B, -1, H, d).transpose(1, 2) for t in (q, k, v)]
    s = hq @ hk.transpose(-1, -2) / math.sqrt(d)
    a = s.softmax(-1)
    if mask is not None:
        a = a * mask / (1 - p)
    return (a @ hv).transpose(1, 2).reshape(B, Lq, E)


@pytest.mark.parametrize("B,H,Lq,Lk", [(2, 8, 200, 32400), (1, 2, 37, 5000), (3, 4, 256, 2049), (1, 8, 16, 4096)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_cross_attention_matches_reference(dev, B, H, Lq, Lk, p):
    torch.manual_seed(0)
    E = H * 16
    q = (torch.randn(B, Lq, E, device=dev) * 1.5).to(torch.bfloat16).requires_grad_(True)
    k = (torch.randn(B, Lk, E, device=dev) * 1.5).to(torch.bfloat16).requires_grad_(True)
    v = torch.randn(B, Lk, E, device=dev).to(torch.bfloat16).requires_grad_(True)
    assert at.supported(q, k, v, H)
    seed = 1234567
    out = at.cross_attention(q, k, v, H, p, seed)
    mask = at.dropout_mask(B, H, Lq, Lk, p, seed, dev) if p > 0 else None
    if mask is not None:
        assert abs(float(mask.float().mean()) - (1 - p)) < 5e-3
    qr, kr, vr = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
    ref = reference(qr, kr, vr, H, mask, p)
    assert out.dtype == torch.bfloat16 and rel(out.float(), ref) < 2e-2
    g = torch.randn_like(ref)
    out.backward(g.to(torch.bfloat16))
    ref.backward(g.to(torch.bfloat16).float())
    assert rel(q.grad.float(), qr.grad) < 3e-2
    assert rel(k.grad.float(), kr.grad) < 3e-2
    assert rel(v.grad.float(), vr.grad) < 3e-2


def test_cross_attention_is_reproducible_and_seeded(dev):
    torch.manual_seed(1)
    q = torch.randn(2, 200, 128, device=dev).to(torch.bfloat16)
    k = torch.randn(2, 8000, 128, device=dev).to(torch.bfloat16)
    v = torch.randn(2, 8000, 128, device=dev).to(torch.bfloat16)
    a = at.cross_attention(q, k, v, 8, 0.1, 42)
    b = at.cross_attention(q, k, v, 8, 0.1, 42)
    c = at.cross_attention(q, k, v, 8, 0.1, 43)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert at.next_seed() != at.next_seed()


def test_decoder_layer_uses_split_attention_under_autocast(dev):
    """The spelled-out MHA takes the split-key path for the BEV keys under bf16 autocast and matches the SDPA path."""
    from bevfusion_amd.dense_modules import _MHA
    torch.manual_seed(2)
    m = _MHA(128, 8, dropout=0.0).to(dev).train()
    q = torch.randn(2, 200, 128, device=dev)
    k = torch.randn(2, 9000, 128, device=dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = m(q, k, k)
        at.ENABLED = False
        try:
            ref = m(q, k, k)
        finally:
            at.ENABLED = True
    assert rel(out.float(), ref.float()) < 2e-2


def test_cross_attention_edge_sizes(dev):
    """One query, a key count that is not a multiple of the 512-key chunk, and the smallest supported key count."""
    torch.manual_seed(3)
    for B, H, Lq, Lk in ((1, 1, 1, 2048), (2, 3, 5, 2500), (1, 8, 256, 3001)):
        E = H * 16
        q = torch.randn(B, Lq, E, device=dev).to(torch.bfloat16).requires_grad_(True)
        k = torch.randn(B, Lk, E, device=dev).to(torch.bfloat16).requires_grad_(True)
        v = torch.randn(B, Lk, E, device=dev).to(torch.bfloat16).requires_grad_(True)
        out = at.cross_attention(q, k, v, H, 0.0)
        qr, kr, vr = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
        ref = reference(qr, kr, vr, H)
        assert rel(out.float(), ref) < 2e-2, (B, H, Lq, Lk)
        out.float().sum().backward()
        ref.sum().backward()
        assert rel(v.grad.float(), vr.grad) < 3e-2 and rel(q.grad.float(), qr.grad) < 5e-2
    assert not at.supported(torch.zeros(1, 300, 128, device=dev, dtype=torch.bfloat16), k[:1, :, :128].expand(1, -1, 128), k[:1, :, :128].expand(1, -1, 128), 8)
