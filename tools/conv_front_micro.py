#!/usr/bin/env python3
"""Host time per convolution call (forward, forward + backward) through the two front-ends of conv2d.py: the Python autograd Functions
(BFHIP_CONV_EXT=0) and csrc/torch_binding.cpp.  A small layer, so the GPU is never the limit: wall time per call = issue time."""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bevfusion_amd  # noqa: E402,F401
from bevfusion_amd import conv2d as c2  # noqa: E402

dev = torch.device("cuda:0")
conv = c2.Conv2d(64, 64, 3, padding=1, bias=False).to(dev).train()
conv.weight.data = conv.weight.data.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
hyb = c2.Conv2dHipWgrad(64, 64, 3, padding=1, bias=False).to(dev).train()
hyb.weight.data = hyb.weight.data.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
tw = c2.TransposedWeights([conv, hyb])
x = torch.randn(4, 64, 32, 32, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
g = torch.randn(4, 64, 32, 32, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    return dt / n * 1e6


def fwd(m):
    with torch.no_grad():
        m(x)


def fb(m, layers=8):
    m.weight.grad = None
    h = x
    for _ in range(layers):  # several layers per pass: the end-of-pass group is shared, as in the model
        h = m(h)
    h.backward(g)


for grouped in (True, False):
    for ext in (False, True):
        c2.CONV_EXT, c2.WGRAD_GROUPED = ext, grouped
        with torch.autocast("cuda", dtype=torch.bfloat16):
            a = timed(lambda: fwd(conv))
            b = timed(lambda: fb(conv), 100) / 8
            c = timed(lambda: fb(hyb), 100) / 8
        print("grouped=%d ext=%d  Conv2d forward %.1f us   Conv2d fwd+bwd per layer %.1f us   library-forward layer fwd+bwd %.1f us" % (grouped, ext, a, b, c))
