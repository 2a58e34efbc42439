#!/usr/bin/env python3
"""The ResNet stem (7x7 stride 2, 3 -> 64 on [24, 3, 256, 704]): library conv on 3 channels vs csrc/conv2d.hip on the image padded to
8 channels (channels-last); forward and weight gradient, GPU time by graph replay."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import torch.nn.functional as F

import bevfusion_amd  # noqa: F401
from bevfusion_amd import _lib
from resnet_conv_micro import timed

dev = torch.device("cuda:0")
N, H, W = 24, 256, 704
img = torch.randn(N, 3, H, W, device=dev)
w = (torch.randn(64, 3, 7, 7, device=dev) / 12).to(torch.bfloat16)
xb = img.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
wl = w.contiguous(memory_format=torch.channels_last)
y0 = F.conv2d(xb, wl, None, 2, 3)
gy = torch.randn_like(y0)
x8 = torch.zeros(N, H, W, 8, device=dev, dtype=torch.bfloat16)
x8[..., :3] = img.permute(0, 2, 3, 1)
w8 = torch.zeros(64, 7, 7, 8, device=dev, dtype=torch.bfloat16)
w8[..., :3] = w.permute(0, 2, 3, 1)
OH, OW = y0.shape[2:]
yh = torch.empty(N, OH, OW, 64, device=dev, dtype=torch.bfloat16)
lib = _lib.load()
part = torch.empty(lib.bfhip_conv2d_stat_rows(N, OH, OW), 2, 64, device=dev)
print("supported:", lib.bfhip_conv2d_supported(N, H, W, 8, 64, 7, 7, 2, 3, 1))
st = _lib.stream_of(x8)


def hip_fwd():
    _lib.call("bfhip_conv2d_fwd", x8.data_ptr(), 8, w8.data_ptr(), None, yh.data_ptr(), 64, N, H, W, 8, 64, 7, 7, 2, 3, 1, 0, part.data_ptr(), _lib.stream_of(x8))


hip_fwd()
err = float((yh.permute(0, 3, 1, 2).float() - y0.float()).abs().max() / y0.float().abs().max())
ws = torch.empty(lib.bfhip_conv2d_wgrad_workspace_bytes(N, OH, OW, 8, 64, 7, 7), dtype=torch.uint8, device=dev)
dw = torch.empty(64, 7, 7, 8, device=dev)
gyn = gy.permute(0, 2, 3, 1).contiguous()


def hip_wgrad():
    _lib.call("bfhip_conv2d_wgrad", x8.data_ptr(), 8, gyn.data_ptr(), 64, dw.data_ptr(), N, H, W, 8, 64, 7, 7, 2, 3, 1, 0, ws.data_ptr(), ws.numel(), _lib.stream_of(x8))


def lib_wgrad():
    return torch.ops.aten.convolution_backward(gy, xb, wl, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])[1]


def pad_image():
    t = torch.zeros(N, H, W, 8, device=dev, dtype=torch.bfloat16)
    t[..., :3] = img.permute(0, 2, 3, 1)
    return t


def lib_image():
    return img.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


hip_wgrad()
ref_dw = lib_wgrad().float()
errw = float((dw[..., :3].permute(0, 3, 1, 2) - ref_dw).abs().max() / ref_dw.abs().max())
print("fwd err %.2e  wgrad err %.2e" % (err, errw))
print("lib fwd %.4f  hip fwd %.4f  lib wgrad %.4f  hip wgrad %.4f  pad image %.4f  lib image conversion %.4f ms" % (
    timed(lambda: F.conv2d(xb, wl, None, 2, 3)), timed(hip_fwd), timed(lib_wgrad), timed(hip_wgrad), timed(pad_image), timed(lib_image)))
