#!/usr/bin/env python3
"""GPU time of the decoder's cross attention (200 queries x 32 400 keys, 8 heads x 16, batch 4, dropout 0.1), forward and
forward + backward, by graph replay."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

import bevfusion_amd  # noqa: F401
from bevfusion_amd import attention
from resnet_conv_micro import timed

dev = torch.device("cuda:0")
B, H, Lq, Lk, D = 4, 8, 200, 32400, 16
q = torch.randn(B, Lq, H * D, device=dev).to(torch.bfloat16).requires_grad_(True)
k = torch.randn(B, Lk, H * D, device=dev).to(torch.bfloat16).requires_grad_(True)
v = torch.randn(B, Lk, H * D, device=dev).to(torch.bfloat16).requires_grad_(True)
g = torch.randn(B, Lq, H * D, device=dev).to(torch.bfloat16)
for p in (0.1, 0.0):
    fwd = timed(lambda: attention.cross_attention(q.detach(), k.detach(), v.detach(), H, dropout_p=p, seed=7))
    fb = timed(lambda: torch.autograd.grad(attention.cross_attention(q, k, v, H, dropout_p=p, seed=7), (q, k, v), g))
    print("dropout %.1f: fwd %.4f ms   fwd+bwd %.4f ms   bwd %.4f ms" % (p, fwd, fb, fb - fwd))
