#!/usr/bin/env python3
"""Would running the weight gradient of a dense convolution on a second stream beside its data gradient pay?
Times dgrad + wgrad of ResNet-50 / SECOND-shaped bf16 NHWC convolutions back to back on one stream and concurrently on two."""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: F401  (points MIOpen at the shipped find-db)

dev = torch.device("cuda:0")
side = torch.cuda.Stream()
shapes = [  # (N, Cin, H, W, Cout, k, stride)
    (24, 64, 64, 176, 64, 3, 1), (24, 256, 64, 176, 64, 1, 1), (24, 128, 32, 88, 128, 3, 1), (24, 512, 32, 88, 128, 1, 1),
    (24, 256, 16, 44, 256, 3, 1), (24, 1024, 16, 44, 256, 1, 1), (24, 512, 8, 22, 512, 3, 1),
    (4, 256, 180, 180, 128, 3, 1), (4, 128, 180, 180, 128, 3, 1), (4, 256, 90, 90, 256, 3, 1)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot = [0.0, 0.0]
for (N, ci, H, W, co, k, st) in shapes:
    x = torch.randn(N, ci, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, k, k, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    pad = k // 2
    y = torch.nn.functional.conv2d(x, w, None, st, pad)
    g = torch.randn_like(y)
    args = (g, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1)

    def serial():
        torch.ops.aten.convolution_backward(*args, [True, False, False])
        torch.ops.aten.convolution_backward(*args, [False, True, False])

    def overlapped():
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            torch.ops.aten.convolution_backward(*args, [False, True, False])
            done = torch.cuda.Event()
            done.record()
        torch.ops.aten.convolution_backward(*args, [True, False, False])
        torch.cuda.current_stream().wait_event(done)
    ts, to = timed(serial), timed(overlapped)
    tot[0] += ts
    tot[1] += to
    print("N%d %dx%dx%d -> %d k%d: serial %.1f us, two streams %.1f us" % (N, ci, H, W, co, k, ts, to), flush=True)
print("sum: serial %.1f us, two streams %.1f us" % tuple(tot))
