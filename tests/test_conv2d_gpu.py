"""GPU parity of csrc/conv2d.hip (implicit-GEMM convolution on the matrix cores: forward, data gradient, weight gradient,
BatchNorm statistics in the epilogue) through the C ABI against torch's fp32 convolution on the CPU.

Inputs are rounded to bf16 first, so the oracle sees exactly the operands the kernel sees: what remains is fp32
accumulation order (<= 1e-4 rel, checked on the fp32 outputs: weight gradient, statistics, fp32 forward) and the final
rounding of bf16 outputs (<= 2^-8 rel per element; north star: 1e-2 rel for bf16 features).  No ReLU anywhere: these are
the mask-free gradient checks the module-level tests refer to."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import bevfusion_amd  # noqa: F401
from bevfusion_amd import _lib
from bevfusion_amd.conv2d import Conv2d, conv2d

pytestmark = pytest.mark.gpu

#        N   H   W  Cin Cout k  s  p  d  bias
GEOMS = [(2, 45, 52, 336, 256, 3, 1, 1, 1, False),   # ConvFuser (BF/bevfusion_head.py:26-38), odd extents
         (2, 45, 52, 128, 128, 3, 1, 1, 1, False),   # SECOND block 1
         (2, 44, 52, 128, 256, 3, 2, 1, 1, False),   # SECOND block 2 entry, stride 2
         (2, 45, 52, 512, 128, 3, 1, 1, 1, True),    # shared_conv (bias)
         (3, 32, 88, 320, 256, 3, 1, 1, 1, True),    # depthnet conv 1
         (3, 32, 88, 256, 200, 1, 1, 0, 1, True),    # 1x1 projection, Cout not a multiple of 64
         (2, 90, 92, 80, 80, 3, 2, 1, 1, False),     # downsample, stride 2, Cin = 80 (K pieces straddle taps)
         (2, 47, 33, 32, 64, 5, 2, 2, 1, True),      # dtransform 5x5 stride 2
         (2, 33, 47, 8, 32, 5, 4, 2, 1, True),       # dtransform 5x5 stride 4, 8 input channels
         (1, 20, 24, 64, 64, 3, 1, 2, 2, False),     # dilation 2
         (2, 32, 88, 768, 256, 1, 1, 0, 1, False),   # LSS-FPN lateral 1x1
         (1, 7, 9, 16, 24, 3, 1, 1, 1, False),       # a single partial tile
         (2, 33, 47, 64, 128, 1, 2, 0, 1, False),    # 1x1 stride 2: three of the four parity classes of the data gradient have no tap
         (2, 30, 31, 32, 32, 3, 4, 1, 1, False),     # 3x3 stride 4: 7 of 16 parity classes empty, odd extents
         (1, 17, 19, 16, 16, 5, 4, 2, 1, True),      # 5x5 stride 4 on odd extents: classes with 1, 2 and 4 taps
         (2, 132, 128, 16, 256, 3, 1, 1, 1, False),  # 528 tiles of 128 x 128 = 1.03 residency rounds: forward switches to 128 x 64 tiles
         (2, 223, 225, 16, 256, 3, 1, 1, 1, True),   # >= 384 row tiles x 256 columns: forward takes the 256 x 256 tile kernel
         (2, 223, 225, 256, 16, 3, 1, 1, 1, False),  # ... and here the data gradient does (its GEMM columns are Cin = 256)
         # pointwise kernel (conv_pw_kernel: 1x1, stride 1, <= 128 gathered channels)
         (3, 33, 47, 64, 256, 1, 1, 0, 1, False),    # ResNet bottleneck expand: forward pointwise (one K step), dgrad implicit GEMM
         (3, 33, 47, 256, 64, 1, 1, 0, 1, False),    # ResNet bottleneck reduce: dgrad pointwise over dy (64 channels)
         (2, 33, 47, 72, 40, 1, 1, 0, 1, True),      # 64-column tile, K = 72 (a partial second K step), bias, partial column tile
         (2, 31, 29, 128, 136, 1, 1, 0, 1, False)]   # two K steps, second column tile 8 wide, rows not a multiple of 128


def _bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).float()


def _case(g, seed):
    N, H, W, Cin, Cout, k, s, p, d, bias = g
    rng = np.random.default_rng(seed)
    x = _bf16_round(rng.standard_normal((N, Cin, H, W)).astype(np.float32))
    w = _bf16_round((rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(Cout).astype(np.float32)) if bias else None
    return x, w, b


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("g", GEOMS, ids=lambda g: "x".join(str(v) for v in g[:9]))
def test_forward_backward_vs_torch_cpu(dev, g):
    N, H, W, Cin, Cout, k, s, p, d, bias = g
    x, w, b = _case(g, seed=sum(g[:9]))
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if b is not None else None
    ref = F.conv2d(xr, wr, br, stride=s, padding=p, dilation=d)
    gy = _bf16_round(np.random.default_rng(7).standard_normal(tuple(ref.shape)).astype(np.float32))
    ref.backward(gy)
    xg = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)      # fp32 master-less weights
    bg = b.to(dev).requires_grad_(True) if b is not None else None
    y, partial = conv2d(xg, wg, bg, s, p, d, emit_stats=b is None)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    assert _rel(y.float().cpu(), ref.detach()) < 1e-2 and _l2(y.float().cpu(), ref.detach()) < 4e-3
    if partial is not None:  # BatchNorm statistics of the fp32 accumulators
        M = N * ref.shape[2] * ref.shape[3]
        assert partial.shape == ((M + 127) // 128, 2, Cout)
        s0, s1 = partial[:, 0].double().sum(0).cpu(), partial[:, 1].double().sum(0).cpu()
        r = ref.detach().double()
        assert torch.allclose(s0, r.sum((0, 2, 3)), rtol=1e-4, atol=1e-3 * float(r.abs().max()))
        assert torch.allclose(s1, (r * r).sum((0, 2, 3)), rtol=1e-4)
    y.backward(gy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    assert _rel(xg.grad.float().cpu(), xr.grad) < 1e-2 and _l2(xg.grad.float().cpu(), xr.grad) < 4e-3
    assert wg.grad.dtype == torch.float32 and _rel(wg.grad.cpu(), wr.grad) < 1e-4        # fp32 accumulate, fp32 out
    if b is not None:
        assert _rel(bg.grad.cpu(), br.grad) < 1e-4


@pytest.mark.parametrize("g", [(3, 32, 88, 64, 256, 1, 1, 0, 1, False),     # ResNet bottleneck 1x1 expand
                               (3, 32, 88, 128, 128, 3, 2, 1, 1, False),    # ResNet 3x3 stride 2
                               (2, 44, 52, 256, 64, 1, 1, 0, 1, False)],    # 1x1 reduce
                         ids=lambda g: "x".join(str(v) for v in g[:9]))
@pytest.mark.parametrize("fwd,dgrad", [("lib", "lib"), ("lib", "hip"), ("hip", "hip"), ("hip", "lib")])
def test_library_forward_hip_wgrad_hybrid(dev, g, fwd, dgrad):
    """Conv2dHipWgrad (the ResNet-50 trunk's convolutions): weight gradient by csrc/conv2d.hip, forward and data gradient each
    by the library or the HIP kernel (dense_modules picks per layer); same tolerances as the all-HIP path against torch on the
    CPU in fp32.  A HIP forward must also hand over its BatchNorm statistics."""
    from bevfusion_amd.conv2d import Conv2dHipWgrad
    N, H, W, Cin, Cout, k, s, p, d, _ = g
    x, w, _ = _case(g, seed=sum(g[:9]))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride=s, padding=p, dilation=d)
    gy = _bf16_round(np.random.default_rng(9).standard_normal(tuple(ref.shape)).astype(np.float32))
    ref.backward(gy)
    conv = Conv2dHipWgrad(Cin, Cout, k, stride=s, padding=p, dilation=d, bias=False).to(dev).train()
    conv.fwd, conv.dgrad = fwd, dgrad
    with torch.no_grad():
        conv.weight.copy_(w)
    conv.to(memory_format=torch.channels_last)
    xg = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv(xg)
    assert y.dtype == torch.bfloat16 and _rel(y.float().cpu(), ref.detach()) < 1e-2
    part = getattr(y, "_bfhip_stat_partial", None)
    assert (part is not None) == (fwd == "hip")
    if part is not None:
        r = ref.detach().double()
        assert torch.allclose(part[0][:, 0].double().sum(0).cpu(), r.sum((0, 2, 3)), rtol=1e-4, atol=1e-3 * float(r.abs().max()))
    y.backward(gy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    assert _rel(xg.grad.float().cpu(), xr.grad) < 1e-2 and _l2(xg.grad.float().cpu(), xr.grad) < 4e-3
    assert conv.weight.grad.dtype == torch.float32 and _rel(conv.weight.grad.cpu(), wr.grad) < 1e-4
    # evaluation mode / ineligible calls take nn.Conv2d as is
    conv.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        assert _rel(conv(xg).float().cpu(), ref.detach()) < 1e-2


@pytest.mark.parametrize("geom", [(40, 72, 3, 2, 1), (72, 136, 1, 1, 0), (128, 40, 1, 1, 0)], ids=["3x3s2", "pointwise_72_136", "pointwise_128_40"])
def test_fp32_output_is_exact_on_integer_data(dev, geom):
    """Integer-valued operands: every product and partial sum is exact in fp32, so the fp32-output forward / dgrad and the
    weight gradient must equal torch's CPU result bit for bit -- whatever the summation order.  Catches any misplaced
    tap, row or swizzle that a tolerance could hide.  (The two 1x1 cases run conv_pw_kernel with fp32 output, both tile widths.)"""
    N, H, W, d = 2, 19, 23, 1
    Cin, Cout, k, s, p = geom
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.integers(-3, 4, (N, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy(rng.integers(-2, 3, (Cout, Cin, k, k)).astype(np.float32))
    ref = F.conv2d(x, w, None, stride=s, padding=p, dilation=d)
    OH, OW = ref.shape[2:]
    gy = torch.from_numpy(rng.integers(-2, 3, tuple(ref.shape)).astype(np.float32))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, stride=s, padding=p, dilation=d).backward(gy)
    xd = x.to(dev).to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()
    wd = w.to(dev).to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()
    gd = gy.to(dev).to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()
    y = torch.empty((N, OH, OW, Cout), dtype=torch.float32, device=dev)
    st = _lib.stream_of(xd)
    _lib.call("bfhip_conv2d_fwd", xd.data_ptr(), Cin, wd.data_ptr(), None, y.data_ptr(), Cout, N, H, W, Cin, Cout, k, k, s, p, d, 1,
              None, st)
    assert torch.equal(y.cpu().permute(0, 3, 1, 2), ref)
    lib = _lib.load()
    ws = torch.empty(max(lib.bfhip_conv2d_dgrad_workspace_bytes(Cin, Cout, k, k),
                         lib.bfhip_conv2d_wgrad_workspace_bytes(N, OH, OW, Cin, Cout, k, k)), dtype=torch.uint8, device=dev)
    dx = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dev)
    _lib.call("bfhip_conv2d_dgrad", gd.data_ptr(), Cout, wd.data_ptr(), dx.data_ptr(), Cin, N, H, W, Cin, Cout, k, k, s, p, d, 1,
              ws.data_ptr(), ws.numel(), st)
    assert torch.equal(dx.cpu().permute(0, 3, 1, 2), xr.grad)
    dw = torch.empty((Cout, k, k, Cin), dtype=torch.float32, device=dev)
    _lib.call("bfhip_conv2d_wgrad", xd.data_ptr(), Cin, gd.data_ptr(), Cout, dw.data_ptr(), N, H, W, Cin, Cout, k, k, s, p, d, 0,
              ws.data_ptr(), ws.numel(), st)
    assert torch.equal(dw.cpu().permute(0, 3, 1, 2), wr.grad)


def test_channel_slice_input_and_bf16_weights(dev):
    """A channel slice of a wider channels-last tensor is consumed in place (pixel pitch > C); bf16 parameters get bf16
    weight gradients (the master-weight optimizer's layout)."""
    rng = np.random.default_rng(5)
    wide = torch.from_numpy(rng.standard_normal((2, 96, 34, 36)).astype(np.float32)).to(dev).to(torch.bfloat16)
    wide = wide.contiguous(memory_format=torch.channels_last)
    xs = wide[:, 16:80]
    conv = Conv2d(64, 64, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
    conv.weight.data = conv.weight.data.to(torch.bfloat16)
    conv.train()
    assert conv.hip_eligible(xs)
    y = conv(xs)
    assert getattr(y, "_bfhip_stat_partial", None) is not None
    ref = F.conv2d(xs.float().cpu(), conv.weight.detach().float().cpu(), padding=1)
    assert _l2(y.float().cpu(), ref) < 4e-3
    y.float().square().mean().backward()
    assert conv.weight.grad.dtype == torch.bfloat16 and conv.weight.grad.shape == conv.weight.shape


def test_module_falls_back_outside_its_domain(dev):
    conv = Conv2d(16, 24, 3, padding=1).to(dev)
    x32 = torch.randn(1, 16, 64, 64, device=dev)
    assert not conv.hip_eligible(x32)                       # fp32 without autocast: the library's fp32 convolution
    assert conv(x32).dtype == torch.float32
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert conv.hip_eligible(x32)
        y = conv(x32)
    assert y.dtype == torch.bfloat16
    assert _l2(y.float().cpu(), F.conv2d(x32.cpu(), conv.weight.detach().cpu(), conv.bias.detach().cpu(), padding=1)) < 1e-2
    odd = Conv2d(12, 24, 3, padding=1).to(dev)               # 12 input channels: not a multiple of 8
    assert not odd.hip_eligible(torch.randn(1, 12, 64, 64, device=dev, dtype=torch.bfloat16))
    grouped = Conv2d(16, 16, 3, padding=1, groups=2).to(dev)
    assert not grouped.hip_eligible(torch.randn(1, 16, 64, 64, device=dev, dtype=torch.bfloat16))


def test_conv_bn_relu_fused_statistics_match_unfused(dev):
    """conv -> BatchNorm2dAct with the statistics taken from the conv epilogue equals the same layers with the
    BatchNorm computing its own statistics pass (same kernels otherwise): outputs, running stats, gradients."""
    from bevfusion_amd.bn2d import BatchNorm2dAct
    torch.manual_seed(0)
    conv = Conv2d(64, 128, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last).train()
    res = []
    x0 = torch.randn(2, 64, 45, 52, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for fused in (True, False):
        bn = BatchNorm2dAct(128, act=True).to(dev).train()
        x = x0.clone().requires_grad_(True)
        conv.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv(x)
            assert hasattr(y, "_bfhip_stat_partial")
            if not fused:
                del y._bfhip_stat_partial
            out = bn(y)
        out.float().square().mean().backward()
        res.append((out.detach().float(), bn.running_mean.clone(), bn.running_var.clone(), x.grad.float(), conv.weight.grad.clone()))
    for a, b in zip(*res):
        assert _l2(a, b) < 3e-3, _l2(a, b)   # statistics from fp32 accumulators vs from the bf16-rounded tensor


def test_weight_gradients_on_the_side_stream_are_identical(dev, monkeypatch):
    """conv2d.WGRAD_SIDE_STREAM: the weight gradients of a stack of convolutions launched on their own HIP stream (joined by
    wgrad_join() after the backward) equal, bit for bit, the ones launched in line -- same kernels, same operands; the
    operands are kept alive until the join and the main stream does not read a dW before it."""
    from bevfusion_amd import conv2d as c2
    from bevfusion_amd.conv2d import Conv2d
    torch.manual_seed(0)
    net = torch.nn.Sequential(Conv2d(32, 64, 3, padding=1, bias=False), Conv2d(64, 64, 3, stride=2, padding=1, bias=False),
                              Conv2d(64, 128, 1, bias=False), Conv2d(128, 32, 3, padding=1, bias=False)).to(dev).train()
    for m in net:
        m.weight.data = m.weight.data.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = torch.randn(3, 32, 64, 88, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    grads = {}
    monkeypatch.setattr(c2, "WGRAD_GROUPED", False)  # the in-line launches are the ones compared bit for bit here
    for side in (False, True):
        monkeypatch.setattr(c2, "WGRAD_SIDE_STREAM", side)
        got = []
        for it in range(3):  # several passes in flight behind each other: buffers are recycled while the side stream runs
            for p in net.parameters():
                p.grad = None
            y = net(x * (1.0 + it))
            y.float().square().mean().backward()
            # main-stream work that would overwrite freed operands if they were released too early
            junk = [torch.randn(3, 64, 64, 88, device=dev) for _ in range(4)]
            del junk
            c2.wgrad_join()
            got.append([p.grad.clone() for p in net.parameters()])
        # a second backward WITHOUT clearing .grad: autograd accumulates on the main stream, so these launches stay in line
        net(x).float().square().mean().backward()
        c2.wgrad_join()
        torch.cuda.synchronize()
        got.append([p.grad.clone() for p in net.parameters()])
        grads[side] = [g for step in got for g in step]
    assert all(torch.isfinite(g.float()).all() and g.float().abs().sum() > 0 for g in grads[True])
    for a, b in zip(grads[False], grads[True]):
        assert torch.equal(a, b)


def _wgrad_stack(dev, dtype):
    """A small network that covers both tile shapes of the grouped launch (K > 128: 128 co x 256 k; K <= 128 with Cout > 128:
    256 co x 128 k), a stride-2 3x3 layer, a layer that is NOT groupable (stride 4 onto a 6 x 11 map: 198 output pixels, fewer
    than the six 64-pixel steps the wide kernels want -> its own launch inside the pass), a layer the library serves (its input
    has fewer than conv2d.MIN_PIXELS pixels) and a weight that is used twice (two records for one parameter)."""
    from bevfusion_amd.conv2d import Conv2d
    torch.manual_seed(0)
    convs = torch.nn.ModuleList([Conv2d(32, 64, 3, padding=1, bias=False), Conv2d(64, 64, 3, stride=2, padding=1, bias=False),
                                 Conv2d(64, 256, 1, bias=False), Conv2d(256, 64, 1, bias=False), Conv2d(64, 64, 3, padding=1, bias=True),
                                 Conv2d(64, 32, 3, stride=4, padding=1, bias=False), Conv2d(32, 32, 3, padding=1, bias=False)]).to(dev).train()
    for m in convs:
        m.weight.data = m.weight.data.to(dtype).contiguous(memory_format=torch.channels_last)

    def run(x):
        h = convs[1](convs[0](x))
        h = convs[3](convs[2](h))
        h = convs[4](convs[4](h))       # shared weight
        h = convs[6](convs[5](h))       # [3, 32, 6, 11]
        return h
    return convs, run


@pytest.mark.parametrize("front", ["python", "ext"])
@pytest.mark.parametrize("wdtype", [torch.bfloat16, torch.float32])
def test_grouped_weight_gradients_match_the_in_line_launches(dev, monkeypatch, wdtype, front):
    """conv2d.WGRAD_GROUPED (csrc/conv2d.hip: conv_wgrad_group_kernel): dW of all layers of a backward pass in one launch per tile
    shape at the end of the pass equals the per-layer launches up to the fp32 summation order (the pixel range is cut into a
    different number of splits: <= 1e-5 rel on fp32 dW, one bf16 rounding on bf16 dW); the gradients are in place when
    backward() returns, a second pass without clearing .grad accumulates, and the result is deterministic."""
    from bevfusion_amd import conv2d as c2
    # both front-ends of the convolution Functions: the Python classes of conv2d.py and csrc/torch_binding.cpp
    monkeypatch.setattr(c2, "CONV_EXT", front == "ext")
    if front == "ext" and (_lib.torch_ext() is None or not hasattr(_lib.torch_ext(), "conv2d")):
        pytest.fail("bfhip_torch_ext.so is missing or stale: the C++ front-end must be built in-tree")
    convs, run = _wgrad_stack(dev, wdtype)
    x = torch.randn(3, 32, 48, 88, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    calls = []
    orig = _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
    res = {}
    for grouped in (False, True, True):
        monkeypatch.setattr(c2, "WGRAD_GROUPED", grouped)
        del calls[:]
        for p in convs.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            run(x).float().square().mean().backward()
        first = [p.grad.clone() for p in convs.parameters()]     # readable as soon as backward() has returned
        n_inline, n_group = calls.count("bfhip_conv2d_wgrad"), calls.count("bfhip_conv2d_wgrad_group_launch")
        with torch.autocast("cuda", dtype=torch.bfloat16):
            run(x).float().square().mean().backward()            # accumulates into the existing .grad
        second = [p.grad.clone() for p in convs.parameters()]
        torch.cuda.synchronize()
        res.setdefault(grouped, []).append((first, second, n_inline, n_group))
    (f0, s0, inline0, group0), = res[False]
    (f1, s1, inline1, group1), (f2, s2, _, _) = res[True]
    if front == "python":
        assert (inline0, group0) == (7, 0) and (inline1, group1) == (1, 1)   # 7 HIP conv calls; only the 6 x 11 layer keeps its own launch
    else:
        assert inline0 + group0 + inline1 + group1 == 0 and _lib.torch_ext().pending_wgrads() == 0   # nothing went through ctypes
    assert not c2._PENDING
    tol = 1e-5 if wdtype == torch.float32 else 2.0 ** -7
    for a, b, a2, b2 in zip(f0, f1, s0, s1):
        assert torch.isfinite(b.float()).all() and b.float().abs().sum() > 0
        assert _l2(a.float(), b.float()) < tol, _l2(a.float(), b.float())
        assert _l2(a2.float(), b2.float()) < 2 * tol
        assert _l2(b2.float(), 2 * b.float()) < 2.0 ** -7                # second pass added the same gradient
    for b, c in zip(f1 + s1, f2 + s2):
        assert torch.equal(b, c)                                         # fixed-order slab sums: bit-identical run to run


@pytest.mark.parametrize("front", ["python", "ext"])
def test_grouped_weight_gradients_survive_a_failed_backward_pass(dev, monkeypatch, front):
    """A backward pass that dies with an exception after a convolution has queued its weight gradient never runs its end-of-pass
    callback; the next pass must start its own list and its own callback (records are kept per graph task), not inherit the dead
    one's records or wait for a callback that will not come."""
    from bevfusion_amd import conv2d as c2
    from bevfusion_amd.conv2d import Conv2d
    monkeypatch.setattr(c2, "CONV_EXT", front == "ext")

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t, fail):
            ctx.fail = fail
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            if ctx.fail:
                raise RuntimeError("boom")
            return g, None

    torch.manual_seed(0)
    conv = Conv2d(32, 64, 3, padding=1, bias=False).to(dev).train()
    conv.weight.data = conv.weight.data.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = torch.randn(3, 32, 48, 88, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    conv(Boom.apply(x, False)).float().square().mean().backward()
    want = conv.weight.grad.clone()
    conv.weight.grad = None
    with pytest.raises(RuntimeError, match="boom"):
        conv(Boom.apply(x, True)).float().square().mean().backward()
    assert conv.weight.grad is None
    conv(Boom.apply(x, False)).float().square().mean().backward()
    assert torch.equal(conv.weight.grad, want)


def test_backward_on_the_calling_thread_gives_the_same_gradients(dev):
    """bench.py runs the backward pass on the calling thread (`torch.autograd.set_multithreading_enabled(False)`: one process per
    GPU needs no per-device engine thread, and the hand-over costs 3-4 ms of host time per step).  The grouped weight gradients hang
    on an engine callback and one branch of this network runs on a second HIP stream: both must behave the same either way --
    bit-identical gradients (these kernels are deterministic)."""
    convs, run = _wgrad_stack(dev, torch.bfloat16)
    side = torch.cuda.Stream(device=dev)
    x = torch.randn(3, 32, 48, 88, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def step():
        for p in convs.parameters():
            p.grad = None
        main = torch.cuda.current_stream(dev)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            a = run(x)
            side.wait_stream(main)
            with torch.cuda.stream(side):      # a second branch through the first two layers, on its own stream
                b = convs[1](convs[0](x * 0.5))
            main.wait_stream(side)
            b.record_stream(main)
            (a.float().square().mean() + b.float().square().mean()).backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in convs.parameters()]

    want = step()
    try:
        torch.autograd.set_multithreading_enabled(False)
        got = step()
    finally:
        torch.autograd.set_multithreading_enabled(True)
    assert all(torch.isfinite(g.float()).all() and g.float().abs().sum() > 0 for g in got)
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def test_autograd_grad_with_respect_to_a_weight_is_not_deferred(dev, monkeypatch):
    """torch.autograd.grad(y, (x, weight)): the engine captures both gradients from the graph and leaves .grad alone, so the layer
    must launch its own weight gradient and return it (the C++ front-end sees the pass's explicit input list); the result equals
    what backward() stores through the grouped launch up to the split count."""
    from bevfusion_amd import conv2d as c2
    monkeypatch.setattr(c2, "CONV_EXT", True)
    monkeypatch.setattr(c2, "WGRAD_GROUPED", True)
    if c2._conv_ext() is None:
        pytest.fail("bfhip_torch_ext.so is missing or stale: the C++ front-end must be built in-tree")
    torch.manual_seed(0)
    conv = Conv2d(32, 64, 3, padding=1, bias=False).to(dev).train()
    conv.weight.data = conv.weight.data.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = torch.randn(3, 32, 48, 88, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn(3, 64, 48, 88, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    queued = _lib.torch_ext().pending_wgrads()   # (a pass that died in an earlier test may have left its records behind)
    dx, dw = torch.autograd.grad(conv(x), (x, conv.weight), g)
    assert conv.weight.grad is None and x.grad is None and _lib.torch_ext().pending_wgrads() == queued
    conv(x).backward(g)
    assert _l2(conv.weight.grad.float(), dw.float()) < 2.0 ** -7 and torch.equal(x.grad, dx)


def test_grouped_weight_gradient_full_size_linearity(dev):
    """Size-independent property at the ConvFuser's real size (4 x 336 x 180 x 180 -> 256, 3x3): dW is linear in dy, so
    dW(dy1 + dy2) = dW(dy1) + dW(dy2) up to bf16 rounding of the operands -- checked on fp32 weights with dy values that are
    exact in bf16 (small integers), where every product and every partial sum below 2^24 is exact: the three gradients must
    then agree BIT FOR BIT whatever the split count."""
    from bevfusion_amd import conv2d as c2
    from bevfusion_amd.conv2d import Conv2d
    torch.manual_seed(1)
    conv = Conv2d(336, 256, 3, padding=1, bias=False).to(dev).train()
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    g = torch.Generator(device="cpu").manual_seed(2)
    x = torch.randint(-2, 3, (4, 336, 180, 180), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dys = [torch.randint(-2, 3, (4, 256, 180, 180), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
           for _ in range(2)]
    out = []
    for dy in (dys[0], dys[1], dys[0] + dys[1]):
        conv.weight.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            conv(x).backward(dy)
        out.append(conv.weight.grad.clone())
    assert out[2].abs().max() < 2 ** 24 and out[2].abs().sum() > 0
    assert torch.equal(out[0] + out[1], out[2])
    # and against the per-layer launch: integers again, so the different split count cannot show
    c2.WGRAD_GROUPED = False
    try:
        conv.weight.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            conv(x).backward(dys[0] + dys[1])
    finally:
        c2.WGRAD_GROUPED = True
    assert torch.equal(conv.weight.grad, out[2])


def test_epilogue_statistics_do_not_survive_an_in_place_edit(dev):
    """conv -> y.mul_(2) -> BatchNorm2dAct: the statistics the conv epilogue took describe y BEFORE the edit; the BatchNorm must
    notice (tensor version counter) and run its own statistics pass -- result equal to a BatchNorm that never saw the attribute."""
    from bevfusion_amd.bn2d import BatchNorm2dAct
    torch.manual_seed(0)
    conv = Conv2d(32, 64, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last).train()
    x = torch.randn(2, 32, 40, 44, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    outs = []
    for strip in (False, True):
        bn = BatchNorm2dAct(64, act=True).to(dev).train()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv(x)
        assert hasattr(y, "_bfhip_stat_partial")
        yd = y.detach()                                  # same storage, same version counter, no autograd history
        yd._bfhip_stat_partial = y._bfhip_stat_partial   # a caller that carries the attribute along
        yd.mul_(2.0).add_(1.0)
        if strip:
            del yd._bfhip_stat_partial
        outs.append((bn(yd).float(), bn.running_mean.clone(), bn.running_var.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_transposed_weight_cache_matches_per_call_transpose_and_detects_stale_copies(dev):
    """conv2d.TransposedWeights: one launch transposes the weights of all layers; a data gradient that uses the cached copy is
    bit-identical to one that transposes per call; a weight edited through torch afterwards is NOT served from the stale copy."""
    from bevfusion_amd.conv2d import Conv2dHipWgrad, TransposedWeights
    torch.manual_seed(3)
    convs = [Conv2d(64, 96, 3, padding=1, bias=False), Conv2d(40, 72, 1, bias=False), Conv2d(32, 32, 3, stride=2, padding=1, bias=False),
             Conv2dHipWgrad(64, 128, 1, bias=False), Conv2dHipWgrad(48, 48, 3, padding=1, bias=False)]
    convs[3].fwd = convs[3].dgrad = "hip"     # the last one keeps the library data gradient: not in the table
    convs = [c.to(dev).to(memory_format=torch.channels_last).train() for c in convs]
    convs[1].weight.data = convs[1].weight.data.to(torch.bfloat16)   # a bf16 parameter beside fp32 ones
    xs = [torch.randn(2, c.in_channels, 24, 28, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for c in convs]

    def grads():
        out = []
        for c, x in zip(convs, xs):
            xr = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = c(xr)
            g = torch.ones_like(y) * 0.5
            out.append(torch.autograd.grad(y, xr, g)[0])
        return out

    base = grads()
    tw = TransposedWeights(convs)
    assert len(tw.items) == 4 and all(getattr(c.weight, "_bfhip_wt", None) is not None for c in convs[:4])
    assert getattr(convs[4].weight, "_bfhip_wt", None) is None
    for a, b in zip(base, grads()):
        assert torch.equal(a, b)
    with torch.no_grad():
        convs[0].weight.mul_(2.0)             # version bump: the copy of layer 0 is stale now
    stale = grads()
    assert torch.allclose(stale[0].float(), 2.0 * base[0].float(), rtol=2e-2, atol=1e-3)
    tw.refresh()
    fresh = grads()
    assert torch.equal(fresh[0], stale[0]) and torch.equal(fresh[1], base[1])


@pytest.mark.parametrize("cached", [True, False], ids=["fused_addend", "torch_add"])
def test_forked_conv_adds_the_identity_gradient_in_its_data_gradient(dev, cached):
    """Residual block entry: x feeds a 1x1 conv and the identity branch.  Conv2dHipWgrad.forward_fork returns (conv(x), x') and the
    gradient that reaches x' is added inside the data gradient's epilogue (with the transposed weight at hand) or by a torch add
    (without) -- both must equal dgrad + identity gradient of the plain graph, computed in fp32 on the CPU."""
    from bevfusion_amd.conv2d import Conv2dHipWgrad, TransposedWeights
    N, H, W, Cin, Cout = 3, 33, 47, 256, 64
    x, w, _ = _case((N, H, W, Cin, Cout, 1, 1, 0, 1, False), seed=11)
    rng = np.random.default_rng(12)
    gy = _bf16_round(rng.standard_normal((N, Cout, H, W)).astype(np.float32))
    gi = _bf16_round(rng.standard_normal((N, Cin, H, W)).astype(np.float32))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    (F.conv2d(xr, wr) * gy).sum().backward()
    dx_ref = xr.grad + gi
    conv = Conv2dHipWgrad(Cin, Cout, 1, bias=False).to(dev).train()
    conv.fwd = conv.dgrad = "hip"
    with torch.no_grad():
        conv.weight.copy_(w)
    conv.to(memory_format=torch.channels_last)
    if cached:
        TransposedWeights([conv])
    xg = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y, ident = conv.forward_fork(xg)
    assert ident is not xg and ident.data_ptr() == xg.data_ptr() and getattr(y, "_bfhip_stat_partial", None) is not None
    cl = lambda t: t.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    torch.autograd.backward([y, ident], [cl(gy), cl(gi)])
    assert _rel(xg.grad.float().cpu(), dx_ref) < 1e-2 and _l2(xg.grad.float().cpu(), dx_ref) < 4e-3
    assert _rel(conv.weight.grad.cpu(), wr.grad) < 1e-4
    # only one of the two outputs used: the other gradient is absent, not zero-filled
    xg2 = xg.detach().clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y2, ident2 = conv.forward_fork(xg2)
    ident2.backward(cl(gi))
    assert torch.equal(xg2.grad, cl(gi))


@pytest.mark.parametrize("cached", [True, False], ids=["fused_addend", "torch_add"])
@pytest.mark.parametrize("hw", [(32, 46), (33, 47)], ids=["even", "odd"])
def test_stride2_shortcut_through_the_subsampled_fork(dev, cached, hw):
    """Entry of a down-sampling residual block: x feeds conv1 (1x1, stride 1) and a 1x1 stride-2 shortcut.  forward_fork(x, 2)
    returns conv1(x) and x at its even pixels; the shortcut runs on that as a stride-1 pointwise conv (forward_unstrided) and its
    compact gradient is added at the even pixels inside conv1's data gradient.  Outputs and all gradients against the plain
    two-convolution graph in fp32 on the CPU (odd extents: the last row / column is an even pixel)."""
    from bevfusion_amd.conv2d import Conv2dHipWgrad, TransposedWeights
    H, W = hw
    N, Cin, C1, C2 = 2, 128, 64, 256
    rng = np.random.default_rng(21)
    x = _bf16_round(rng.standard_normal((N, Cin, H, W)).astype(np.float32))
    w1 = _bf16_round((rng.standard_normal((C1, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32))
    w2 = _bf16_round((rng.standard_normal((C2, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32))
    OH, OW = (H + 1) // 2, (W + 1) // 2
    g1 = _bf16_round(rng.standard_normal((N, C1, H, W)).astype(np.float32))
    g2 = _bf16_round(rng.standard_normal((N, C2, OH, OW)).astype(np.float32))
    xr, w1r, w2r = x.clone().requires_grad_(True), w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    y1r, y2r = F.conv2d(xr, w1r), F.conv2d(xr, w2r, stride=2)
    torch.autograd.backward([y1r, y2r], [g1, g2])
    conv1 = Conv2dHipWgrad(Cin, C1, 1, bias=False)
    conv1.fwd = conv1.dgrad = "hip"
    short = Conv2dHipWgrad(Cin, C2, 1, stride=2, bias=False)
    short.cache_wt = True
    for c, w in ((conv1, w1), (short, w2)):
        c.to(dev).train()
        with torch.no_grad():
            c.weight.copy_(w)
        c.to(memory_format=torch.channels_last)
    if cached:
        tw = TransposedWeights([conv1, short])
        assert len(tw.items) == 2
    cl = lambda t: t.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    xg = cl(x).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y1, x_sub = conv1.forward_fork(xg, subsample=2)
        assert tuple(x_sub.shape) == (N, Cin, OH, OW)
        y2 = short.forward_unstrided(x_sub)
    assert _rel(y1.float().cpu(), y1r.detach()) < 1e-2 and _rel(y2.float().cpu(), y2r.detach()) < 1e-2
    assert getattr(y2, "_bfhip_stat_partial", None) is not None
    torch.autograd.backward([y1, y2], [cl(g1), cl(g2)])
    assert _rel(xg.grad.float().cpu(), xr.grad) < 1e-2 and _l2(xg.grad.float().cpu(), xr.grad) < 4e-3
    assert _rel(conv1.weight.grad.cpu(), w1r.grad) < 1e-4 and _rel(short.weight.grad.cpu(), w2r.grad) < 1e-4
    # the shortcut alone (conv1's output unused): its gradient must still reach x, at the even pixels only
    xg2 = xg.detach().clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, x_sub2 = conv1.forward_fork(xg2, subsample=2)
    x_sub2.backward(torch.ones_like(x_sub2))
    want = torch.zeros_like(xg2)
    want[:, :, ::2, ::2] = 1
    assert torch.equal(xg2.grad, want)


def test_fp32_convolution_by_three_bf16_products(dev):
    """An fp32 convolution outside autocast (the heat-map head's fp32 island) runs as three bf16 products per multiply with fp32
    accumulation (conv2d._Conv2dSplitFunction).  Against an fp64 convolution on the CPU: forward, data gradient and weight
    gradient within 3e-5 relative (bound: 2^-16 per product; exact fp32 would be ~1e-6, TF32 ~1e-3), statistics handed to
    the BatchNorm, and the BFHIP_FP32_CONV=lib switch restores the library path."""
    import bevfusion_amd.conv2d as c2
    N, H, W, Cin, Cout = 2, 45, 52, 128, 128
    rng = np.random.default_rng(31)
    x = torch.from_numpy(rng.standard_normal((N, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32))
    gy = torch.from_numpy(rng.standard_normal((N, Cout, H, W)).astype(np.float32))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, padding=1)
    ref.backward(gy.double())
    conv = Conv2d(Cin, Cout, 3, padding=1, bias=False).to(dev).train()
    with torch.no_grad():
        conv.weight.copy_(w)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert not conv.split_eligible(xg) and not conv.hip_eligible(xg)     # outside an fp32 island of a mixed-precision step: exact fp32
    with c2.fp32_island(True):
        assert conv.split_eligible(xg)
        y = conv(xg)
    assert y.dtype == torch.float32 and getattr(y, "_bfhip_stat_partial", None) is not None
    y.backward(gy.to(dev))
    e_y, e_dx, e_dw = _l2(y.detach().cpu(), ref.detach()), _l2(xg.grad.cpu(), xr.grad), _l2(conv.weight.grad.cpu(), wr.grad)
    assert e_y < 3e-5 and e_dx < 3e-5 and e_dw < 3e-5, (e_y, e_dx, e_dw)
    assert _rel(y.detach().cpu(), ref.detach()) < 1e-4
    part = y._bfhip_stat_partial[0]
    assert torch.allclose(part[:, 0].double().sum(0).cpu(), ref.detach().sum((0, 2, 3)), rtol=1e-3, atol=1e-3 * float(ref.abs().max()))
    # bf16 rounding of the operands alone would be ~100x worse: the lo terms matter
    yb = F.conv2d(x.to(torch.bfloat16).double(), w.to(torch.bfloat16).double(), None, padding=1)
    assert _l2(yb, ref.detach()) > 30 * e_y
    old = c2.FP32_SPLIT
    c2.FP32_SPLIT = False
    try:
        with c2.fp32_island(True):
            assert not conv.split_eligible(xg)
            assert _l2(conv(xg).detach().cpu(), ref.detach()) < 1e-5
    finally:
        c2.FP32_SPLIT = old
