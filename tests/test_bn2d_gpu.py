"""GPU: fused BatchNorm2d(+residual)(+ReLU) on channels-last activations (csrc/bn2d.hip) against torch's own
batch_norm / add / relu in fp32 on the same inputs.  Tolerances: fp32 1e-4 (max-normalised); bf16 storage 1e-2."""
import pytest
import torch
import torch.nn.functional as F

import bevfusion_amd  # noqa: F401
from bevfusion_amd import bn2d

pytestmark = pytest.mark.gpu


def _ref(x, res, bn, relu):
    """fp32 torch reference with autograd."""
    x32 = x.detach().float().requires_grad_(True)
    r32 = res.detach().float().requires_grad_(True) if res is not None else None
    w = bn.weight.detach().clone().requires_grad_(True)
    b = bn.bias.detach().clone().requires_grad_(True)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    y = F.batch_norm(x32, rm, rv, w, b, True, bn.momentum, bn.eps)
    if r32 is not None:
        y = y + r32
    if relu:
        y = F.relu(y)
    return x32, r32, w, b, rm, rv, y


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 1.5e-2)])
@pytest.mark.parametrize("shape", [(6, 64, 33, 47), (2, 8, 64, 176), (3, 80, 19, 21), (2, 2048, 8, 22), (5, 256, 20, 20), (1, 32, 7, 5), (2, 328, 9, 11)])
@pytest.mark.parametrize("res,relu", [(False, True), (True, True), (False, False), (True, False)])
def test_bn2d_matches_torch(dev, dtype, tol, shape, res, relu):
    torch.manual_seed(0)
    N, C, H, W = shape
    x = (torch.randn(shape, device=dev) * 1.7 + 0.4).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn(shape, device=dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True) if res else None
    bn = bn2d.BatchNorm2dAct(C, eps=1e-3, momentum=0.05).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    x32, r32, w, b, rm, rv, yref = _ref(x, r, bn, relu)
    assert bn.fusable(x)
    y = bn(x, residual=r, relu=relu)
    assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    assert rel(y.float(), yref) < tol
    assert rel(bn.running_mean, rm) < 1e-4 and rel(bn.running_var, rv) < 1e-4
    g = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last)
    y.backward(g.to(dtype))
    yref.backward(g.to(dtype).float())
    assert rel(x.grad.float(), x32.grad) < tol
    assert rel(bn.weight.grad, w.grad) < tol and rel(bn.bias.grad, b.grad) < tol
    if res:
        assert rel(r.grad.float(), r32.grad) < tol


def test_relu_mask_recomputed_from_x_equals_mask_from_y(dev):
    """Without a residual the backward recomputes the ReLU mask from x; it must zero exactly the entries that are 0 in y."""
    torch.manual_seed(1)
    x = torch.randn(4, 64, 30, 30, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bn = bn2d.BatchNorm2dAct(64, act=True).to(dev).train()
    y = bn(x)
    y.backward(torch.ones_like(y))
    z = torch.zeros_like(x).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    # with dy = 1 and the same mask, a residual input's gradient IS the mask
    bn2 = bn2d.BatchNorm2dAct(64, act=True).to(dev).train()
    y2 = bn2(x.detach(), residual=z)
    y2.backward(torch.ones_like(y2))
    assert torch.equal(y, y2)
    assert torch.equal(z.grad > 0, y > 0)
    assert torch.equal(bn.bias.grad, bn2.bias.grad)  # dbeta = sum of the masked gradient: same mask both ways


def test_unsupported_shapes_fall_back(dev):
    bn = bn2d.BatchNorm2dAct(6, act=True).to(dev).train()
    x = torch.randn(2, 6, 5, 5, device=dev).contiguous(memory_format=torch.channels_last)
    assert not bn.fusable(x)               # 6 channels: not a multiple of the 16-byte vector
    y = bn(x)
    assert torch.allclose(y, F.relu(F.batch_norm(x, None, None, bn.weight, bn.bias, True, 0.1, bn.eps)), atol=1e-5)
    bn64 = bn2d.BatchNorm2dAct(64, act=True).to(dev).train()
    xc = torch.randn(2, 64, 5, 5, device=dev)   # NCHW-contiguous: not channels-last
    assert not bn64.fusable(xc)
    assert bn64.eval()(xc.contiguous(memory_format=torch.channels_last)).shape == xc.shape  # eval: torch path
    assert int(bn64.num_batches_tracked) == 0
    bn64.train()(xc.contiguous(memory_format=torch.channels_last))
    assert int(bn64.state_dict()["num_batches_tracked"]) == 1  # lazy counter flushed when the state dict is read


def test_resnet_block_state_dict_keys(dev):
    from bevfusion_amd.dense_modules import ResNet50, SECOND
    keys = set(ResNet50().state_dict().keys())
    for k in ("conv1.weight", "bn1.running_var", "layer1.0.bn3.weight", "layer1.0.downsample.1.num_batches_tracked",
              "layer4.2.conv3.weight"):
        assert k in keys, k
    sk = set(SECOND(256, [128, 256], [5, 5], [1, 2]).state_dict().keys())
    assert "blocks.0.1.weight" in sk and "blocks.0.3.weight" in sk and "blocks.1.16.running_mean" in sk


def test_position_encoding_rows_equals_conv1d_stack(dev):
    """PositionEncodingLearned evaluated on the row-major [B*N, C] matrix (fused BN) == the Conv1d/BN1d/ReLU/Conv1d stack of
    the reference (BF/transformer.py:10-23) built from the same parameters."""
    from torch import nn
    from bevfusion_amd.dense_modules import PositionEncodingLearned
    torch.manual_seed(0)
    pe = PositionEncodingLearned(2, 128).to(dev).train()
    ref = nn.Sequential(nn.Conv1d(2, 128, 1), nn.BatchNorm1d(128), nn.ReLU(), nn.Conv1d(128, 128, 1)).to(dev).train()
    ref.load_state_dict(pe.position_embedding_head.state_dict())
    xyz = torch.rand(3, 500, 2, device=dev) * 180
    out = pe(xyz)
    want = ref(xyz.transpose(1, 2).contiguous())
    assert out.shape == want.shape == (3, 128, 500)
    assert rel(out, want) < 1e-4
    out.square().mean().backward()
    want.square().mean().backward()
    for (n, p), (_, q) in zip(pe.position_embedding_head.named_parameters(), ref.named_parameters()):
        if n == "0.bias":   # a bias in front of a BatchNorm has exactly zero gradient
            assert float(p.grad.abs().max()) < 1e-8 and float(q.grad.abs().max()) < 1e-8
            continue
        assert rel(p.grad, q.grad) < 1e-3, n
    assert rel(pe.position_embedding_head[1].running_var, ref[1].running_var) < 1e-4


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("shape", [(3, 64, 8, 22), (2, 256, 16, 44), (1, 8, 1, 1), (2, 16, 5, 3)])
def test_upsample2x_matches_interpolate(dev, dtype, tol, shape):
    from bevfusion_amd.dense_modules import upsample_to
    torch.manual_seed(0)
    x = torch.randn(shape, device=dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    cfg = dict(mode="bilinear", align_corners=False)
    size = (2 * shape[2], 2 * shape[3])
    y = upsample_to(x, size, cfg)
    xr = x.detach().float().requires_grad_(True)
    yr = F.interpolate(xr, size=size, **cfg)
    assert y.dtype == dtype and y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert rel(y.float(), yr) < tol
    g = torch.randn_like(yr)
    y.backward(g.to(dtype))
    yr.backward(g.to(dtype).float())
    assert rel(x.grad.float(), xr.grad) < tol
    # non-2x sizes take the torch path
    assert upsample_to(x.detach(), (size[0] + 1, size[1]), cfg).shape[2] == size[0] + 1


def test_bn2d_tiny_and_odd_sizes(dev):
    """M smaller than one slab, M not a multiple of the row lanes, a single spatial position."""
    torch.manual_seed(5)
    for shape in ((2, 16, 1, 1), (1, 64, 3, 1), (7, 8, 2, 3), (2, 1024, 1, 2)):
        x = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        bn = bn2d.BatchNorm2dAct(shape[1], act=True).to(dev).train()
        assert bn.fusable(x)
        y = bn(x)
        xr = x.detach().clone().requires_grad_(True)
        yr = F.relu(F.batch_norm(xr, None, None, bn.weight, bn.bias, True, 0.1, bn.eps))
        assert torch.allclose(y, yr, atol=2e-5, rtol=1e-4), shape
        g = torch.randn_like(y)
        y.backward(g)
        yr.backward(g)
        assert rel(x.grad, xr.grad) < 1e-3 or float(xr.grad.abs().max()) < 1e-6, shape


def test_bn2d_fold_finalize_opt_in_path():
    """BFHIP_BN2D_FOLD=1 (finalize in the last-arriving block of the reduction launch; measured slower, default off) must stay
    correct: the library reads the switch once per process, so the check runs in a child process over several shapes, twice
    per shape (the arrival counters must come back to zero between launches), against the default path's results."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import bevfusion_amd
from bevfusion_amd import bn2d
dev = torch.device("cuda:0")
out = []
for shape in [(6, 64, 33, 47), (2, 2048, 8, 22), (24, 256, 32, 88), (3, 328, 9, 11)]:
    for rep in range(2):
        torch.manual_seed(1)
        x = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        bn = bn2d.BatchNorm2dAct(shape[1]).to(dev).train()
        y = bn(x, relu=True)
        y.backward(torch.ones_like(y))
        out.append(torch.cat([y.float().flatten()[:4096], x.grad.float().flatten()[:4096], bn.weight.grad, bn.running_var]).cpu())
torch.save(out, sys.argv[1])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for fold in ("0", "1"):
            path = os.path.join(d, "out%s.pt" % fold)
            env = dict(os.environ, BFHIP_BN2D_FOLD=fold)
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=300)
            res[fold] = torch.load(path, weights_only=True)
    for a, b in zip(res["0"], res["1"]):
        # same partial sums, same fixed-order fp64 combine up to the grouping of the tree: last-bit differences only
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("front_end", ["cpp", "python"])
def test_residual_relu_backward_from_the_bit_mask_equals_backward_from_saved_output(dev, monkeypatch, front_end):
    """Bottleneck tail (conv3 -> BN + identity + ReLU): with the statistics handed over by the HIP convolution the forward stores
    the ReLU decision of every element as one bit and the backward reads the bits instead of the saved output
    (bfhip_bn2d_fwd_partials_mask / bfhip_bn2d_bwd_mask).  Output, input gradient, residual gradient and parameter gradients
    must equal the saved-output path bit for bit (odd row count, a channel count with a partial column tile)."""
    from bevfusion_amd import _lib, bn2d as b2
    from bevfusion_amd.conv2d import conv2d
    if front_end == "python":
        monkeypatch.setattr(_lib, "torch_ext", lambda: None)   # the ctypes + autograd.Function path
    elif _lib.torch_ext() is None:
        pytest.skip("C++ front-end not built")
    torch.manual_seed(11)
    N, Cin, C, H, W = 3, 64, 264, 17, 23
    x_in = torch.randn(N, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, Cin, 1, 1, device=dev) / 8).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ident = torch.randn(N, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(N, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = {}
    for bits in (True, False):
        monkeypatch.setattr(b2, "RELU_BITS", bits)
        torch.manual_seed(12)   # the same BatchNorm parameters in both passes
        bn = b2.BatchNorm2dAct(C).to(dev).train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
            bn.weight.copy_(bn.weight.to(torch.bfloat16).float())
        xi, idn = x_in.clone().requires_grad_(True), ident.clone().requires_grad_(True)
        y, partial = conv2d(xi, w, None, 1, 0, 1, True)
        y._bfhip_stat_partial = (partial, y.data_ptr(), y._version)
        out = bn(y, residual=idn, relu=True)
        if front_end == "python":
            saved = [t for t in out.grad_fn.saved_tensors if t is not None]
            assert any(t.dtype == torch.uint8 for t in saved) == bits
        out.backward(gy)
        res[bits] = (out.detach(), xi.grad, idn.grad, bn.weight.grad, bn.bias.grad)
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(129600, 128), (67584, 256), (4681, 72), (5000, 8)])
def test_colsum_matches_torch(dev, dtype, shape):
    """bfhip_colsum (bias gradients: sum of dy over pixels / rows) against an fp64 sum; small matrices take torch's reduction."""
    from bevfusion_amd.bn2d import colsum
    torch.manual_seed(shape[0])
    t = torch.randn(shape, device=dev).to(dtype)
    ref = t.double().sum(0)
    got = colsum(t)
    assert got.dtype == torch.float32 and got.shape == (shape[1],)
    assert torch.allclose(got.double(), ref, rtol=1e-5, atol=1e-3 * float(ref.abs().max()))
    assert torch.equal(colsum(t), got)                                   # fixed summation order
    small = t[:100].contiguous()
    assert torch.allclose(colsum(small).double(), small.double().sum(0), rtol=1e-4, atol=1e-4)
