#!/usr/bin/env python3
"""Per distinct convolution shape of the `full` workload (batch 4): time of the dense weight gradient (csrc/conv2d.hip,
bfhip_conv2d_wgrad: main kernel + slab sum) from the library's own HIP-event scope, TFLOP/s, and its share of the step."""
import collections
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
import torch  # noqa: E402
from bevfusion_amd import _lib, conv2d as c2  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.FullModel(dev, 4, 40000)
shapes = collections.Counter()
o1, o2 = c2._Conv2dFunction.forward, c2._LibConvHipWgradFunction.forward


def spy1(ctx, x, w, b, s, p, d, e, *more):
    shapes[(tuple(x.shape), tuple(w.shape), s, p, d)] += 1
    return o1(ctx, x, w, b, s, p, d, e, *more)


def spy2(ctx, x, w, s, p, d, *more):
    shapes[(tuple(x.shape), tuple(w.shape), s, p, d)] += 1
    return o2(ctx, x, w, s, p, d, *more)


c2._Conv2dFunction.forward, c2._LibConvHipWgradFunction.forward = staticmethod(spy1), staticmethod(spy2)
c2.CONV_EXT = False  # the spies sit on the Python Functions
with torch.autocast("cuda", dtype=torch.bfloat16):
    wl.step_model(wl.inputs, None, wl.gts)
c2._Conv2dFunction.forward, c2._LibConvHipWgradFunction.forward = staticmethod(o1), staticmethod(o2)

lib = _lib.load()
rows = []
for (xs, ws, s, p, d), n in shapes.items():
    N, Cin, H, W = xs
    Cout, _, KH, KW = ws
    OH = (H + 2 * p - d * (KH - 1) - 1) // s + 1
    OW = (W + 2 * p - d * (KW - 1) - 1) // s + 1
    x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, OH, OW, Cout, device=dev).to(torch.bfloat16)
    dw = torch.empty(Cout, KH, KW, Cin, device=dev, dtype=torch.float32)
    wsb = lib.bfhip_conv2d_wgrad_workspace_bytes(N, OH, OW, Cin, Cout, KH, KW)
    ws_ = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
    st = _lib.stream_of(x)

    def run():
        _lib.call("bfhip_conv2d_wgrad", x.data_ptr(), Cin, dy.data_ptr(), Cout, dw.data_ptr(), N, H, W, Cin, Cout, KH, KW, s, p, d, 0,
                  ws_.data_ptr(), ws_.numel(), st)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    _lib.profile_enable(2)
    _lib.profile_read("conv2d_wgrad", reset=True)
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    ms, cnt = _lib.profile_read("conv2d_wgrad", reset=True)
    _lib.profile_enable(0)
    t = ms / max(cnt, 1)
    gf = 2.0 * N * OH * OW * Cout * Cin * KH * KW / 1e9
    rows.append(dict(x=list(xs), w=list(ws), stride=s, layers=n, us=round(t * 1e3, 1), tflops=round(gf / t, 0), step_ms=round(n * t, 3)))
rows.sort(key=lambda r: -r["step_ms"])
for r in rows:
    print(json.dumps(r))
print(json.dumps(dict(total_ms=round(sum(r["step_ms"] for r in rows), 3), launches=sum(r["layers"] for r in rows))))
