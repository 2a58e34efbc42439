#!/usr/bin/env python3
"""csrc/pool.hip against torch's channels-last max_pool2d on the ResNet-50 stem's map [24, 64, 128, 352] (bf16): GPU ms of the
forward and of forward + backward, by graph replay."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import torch.nn.functional as F

import bevfusion_amd  # noqa: F401
from bevfusion_amd.dense_modules import MaxPool3x3s2
from resnet_conv_micro import timed

dev = torch.device("cuda:0")
x = torch.randn(24, 64, 128, 352, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
pool = MaxPool3x3s2()
gy = torch.randn(24, 64, 64, 176, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
for name, f in (("hip", pool), ("torch", lambda t: F.max_pool2d(t, 3, stride=2, padding=1))):
    fwd = timed(lambda: f(x.detach()))
    fb = timed(lambda: torch.autograd.grad(f(x), (x,), gy))
    print("%-6s fwd %.4f ms   fwd+bwd %.4f ms   bwd %.4f ms" % (name, fwd, fb, fb - fwd))
