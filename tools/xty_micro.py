#!/usr/bin/env python3
"""x^T y over K = 129 600 rows (the decoder's K / V projection weight gradients): csrc/xty.hip vs the conv weight-gradient kernel."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

import bevfusion_amd  # noqa: F401
from bevfusion_amd import linear_rows as lr
from resnet_conv_micro import timed

dev = torch.device("cuda:0")
for K, M, N in ((129600, 128, 128), (129600, 256, 128), (32400, 128, 128), (800, 128, 128), (800, 64, 128), (800, 256, 128)):
    x = torch.randn(K, M, device=dev).to(torch.bfloat16)
    y = torch.randn(K, N, device=dev).to(torch.bfloat16)
    ref = x.float().t() @ y.float()
    out = {}
    for conv in (True, False):
        lr.XTY_CONV = conv
        r = lr.xty(x, y)
        out[conv] = (timed(lambda: lr.xty(x, y)), float((r - ref).norm() / ref.norm()))
    mm = timed(lambda: x.t() @ y)
    print((K, M, N), "conv wgrad %.4f ms (err %.1e)   xty %.4f ms (err %.1e)   torch mm %.4f ms" % (out[True] + out[False] + (mm,)))
