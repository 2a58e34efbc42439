#!/usr/bin/env python3
"""1x1 convolution on channels-last bf16 activations: MIOpen conv (fwd + bwd) vs explicit GEMMs on the [N*H*W, C] matrix."""
import sys
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


shapes = [(24, 64, 256, 64, 176), (24, 256, 64, 64, 176), (24, 128, 512, 32, 88), (24, 512, 128, 32, 88),
          (24, 256, 1024, 16, 44), (24, 1024, 256, 16, 44), (24, 512, 2048, 8, 22), (24, 2048, 512, 8, 22)]
for (N, ci, co, H, W) in shapes:
    x = torch.randn(N, ci, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(co, ci, 1, 1, device=dev, dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn(N, co, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def conv_fwd():
        return F.conv2d(x, w)

    def conv_fb():
        y = F.conv2d(x, w)
        torch.autograd.grad(y, (x, w), g)

    xm = x.detach().permute(0, 2, 3, 1).reshape(-1, ci).requires_grad_(True)
    wm = w.detach().view(co, ci).requires_grad_(True)
    gm = g.permute(0, 2, 3, 1).reshape(-1, co)

    def mm_fwd():
        return xm @ wm.t()

    def mm_fb():
        y = xm @ wm.t()
        torch.autograd.grad(y, (xm, wm), gm)

    def mm_bwd_parts():
        dx = gm @ wm
        dw = gm.t() @ xm
        return dx, dw

    print("N%d %4d->%4d %3dx%3d | conv fwd %6.1f us  fwd+bwd %6.1f us | mm fwd %6.1f us  fwd+bwd %6.1f us  (bwd only %6.1f)" %
          (N, ci, co, H, W, timed(conv_fwd), timed(conv_fb), timed(mm_fwd), timed(mm_fb), timed(mm_bwd_parts)))
    sys.stdout.flush()
