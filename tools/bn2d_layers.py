#!/usr/bin/env python3
"""Per distinct BatchNorm shape of the `full` workload (batch 4): forward and backward time of csrc/bn2d.hip, GB/s on the
algorithmic bytes, and how many layers of the step have that shape.  One process, HIP events, nothing else on the GPU."""
import collections
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
import torch  # noqa: E402
from bevfusion_amd import _lib, bn2d as b2  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.FullModel(dev, 4, 40000)
shapes = collections.Counter()
orig = b2._apply


def spy(x, residual, *a, **k):
    if x.is_cuda and x.dim() == 4:
        partial = (len(a) > 7 and a[7] is not None) or k.get("partial") is not None
        shapes[(tuple(x.shape), x.dtype, residual is not None, bool(a[6]) if len(a) > 6 else bool(k.get("relu")), partial)] += 1
    return orig(x, residual, *a, **k)


b2._apply = spy
with torch.autocast("cuda", dtype=torch.bfloat16):
    wl.step_model(wl.inputs, None, wl.gts)
b2._apply = orig


def timed(fn, op, iters=30):
    """GPU time of the library op per call (its own HIP-event scope: no host overhead in the number)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(2)
    _lib.profile_read(op, reset=True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    ms, cnt = _lib.profile_read(op, reset=True)
    _lib.profile_enable(0)
    return ms / max(cnt, 1)


rows = []
for (shape, dtype, res, relu, partial), n in sorted(shapes.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][0]).numel()):
    N, C, H, W = shape
    x = torch.randn(N, C, H, W, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn_like(x) if res else None
    w, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    g = torch.randn_like(x)
    y = b2._BN2dFunction.apply(x, r, w, b, rm, rv, 1e-5, 0.1, relu, None, None)

    def fwd():
        with torch.no_grad():
            b2._BN2dFunction.apply(x.detach(), r, w, b, rm, rv, 1e-5, 0.1, relu, None, None)

    def fwd_bwd():
        yy = b2._BN2dFunction.apply(x, r, w, b, rm, rv, 1e-5, 0.1, relu, None, None)
        yy.backward(g)
        x.grad = None

    tf = timed(fwd, "bn2d_fwd")
    tb = timed(fwd_bwd, "bn2d_bwd")
    nbytes = x.numel() * x.element_size()
    rows.append(dict(shape=list(shape), dtype=str(dtype).replace("torch.", ""), residual=res, relu=relu, layers=n, mbytes=round(nbytes / 1e6, 2),
                     fwd_us=round(tf * 1e3, 1), bwd_us=round(tb * 1e3, 1),
                     fwd_gbs=round(nbytes * (3 + res) / tf / 1e6, 0), bwd_gbs=round(nbytes * (5 + 2 * res) / tb / 1e6, 0),
                     step_fwd_ms=round(n * tf, 3), step_bwd_ms=round(n * tb, 3)))
for r in rows:
    print(json.dumps(r))
print(json.dumps(dict(total_fwd_ms=round(sum(r["step_fwd_ms"] for r in rows), 3), total_bwd_ms=round(sum(r["step_bwd_ms"] for r in rows), 3))))
