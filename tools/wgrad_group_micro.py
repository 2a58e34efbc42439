#!/usr/bin/env python3
"""Dense weight gradients of the `full` workload's 77 layers (shapes of profiles/r03_conv_wgrad_layers.txt): one launch pair per layer
(bfhip_conv2d_wgrad) against the grouped launch (bfhip_conv2d_wgrad_group_*) at several step targets; GPU time by HIP events around
20 back-to-back repeats, results compared layer by layer."""
import ctypes
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bevfusion_amd  # noqa: E402,F401
from bevfusion_amd import _lib, conv2d as c2  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
layers = []
flops = 0.0
for line in open(os.path.join(ROOT, "profiles", "r03_conv_wgrad_layers.txt")):
    d = json.loads(line)
    if "x" not in d:
        continue
    N, Cin, H, W = d["x"]
    Cout, _, KH, KW = d["w"]
    s, p = d["stride"], KH // 2
    OH, OW = (H + 2 * p - KH) // s + 1, (W + 2 * p - KW) // s + 1
    x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, OH, OW, Cout, device=dev).to(torch.bfloat16)
    for _ in range(d["layers"]):
        layers.append((x, dy, (N, H, W, Cin, Cout, KH, KW, s, p, 1), OH, OW))
        flops += 2.0 * N * OH * OW * Cout * KH * KW * Cin
n = len(layers)
st = _lib.stream_of(layers[0][0])
dws = [torch.empty(g[4], g[5], g[6], g[3], device=dev, dtype=torch.bfloat16) for _, _, g, _, _ in layers]
dws2 = [torch.empty_like(d) for d in dws]
wsb = max(lib.bfhip_conv2d_wgrad_workspace_bytes(g[0], oh, ow, g[3], g[4], g[5], g[6]) for _, _, g, oh, ow in layers)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)


def inline():
    for (x, dy, g, oh, ow), dw in zip(layers, dws):
        N, H, W, Cin, Cout, KH, KW, s, p, d = g
        _lib.call("bfhip_conv2d_wgrad", x.data_ptr(), Cin, dy.data_ptr(), Cout, dw.data_ptr(), N, H, W, Cin, Cout, KH, KW, s, p, d, 1,
                  ws.data_ptr(), ws.numel(), st)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {"layers": n, "tflop_per_pass": flops / 1e12}
t = timed(inline)
out["inline_ms"] = round(t, 4)
out["inline_tflops"] = round(flops / t / 1e9, 1)
rows = np.array([(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), g[3], g[4]) + g + (1, 0) for (x, dy, g, _, _), dw in zip(layers, dws2)],
                dtype=c2._layer_dtype())
nb = lib.bfhip_conv2d_wgrad_group_table_bytes(n)
host = torch.empty(nb, dtype=torch.uint8).pin_memory()
table = torch.empty(nb, dtype=torch.uint8, device=dev)
for target in (32, 48, 64, 96, 128, 192, 256):
    sb = ctypes.c_size_t(0)
    _lib.call("bfhip_conv2d_wgrad_group_plan", rows.ctypes.data, n, target, host.data_ptr(), nb, ctypes.byref(sb))
    slab = torch.empty(sb.value + 256, dtype=torch.uint8, device=dev)
    table.copy_(host, non_blocking=True)
    torch.cuda.synchronize()

    def group():
        _lib.call("bfhip_conv2d_wgrad_group_launch", host.data_ptr(), table.data_ptr(), slab.data_ptr(), slab.numel(), st)
    t = timed(group)
    hd = host.numpy()[:256].view(np.int32)
    err = max(float((a.float() - b.float()).norm() / a.float().norm()) for a, b in zip(dws, dws2))
    out["group_%d" % target] = dict(ms=round(t, 4), tflops=round(flops / t / 1e9, 1), slab_mb=round(sb.value / 1e6, 1),
                                    blocks=[int(hd[8]), int(hd[9])], worst_rel_l2_vs_inline=err)
    del slab
print(json.dumps(out, indent=1))
