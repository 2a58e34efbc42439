"""Linear layers over very many rows (every BEV cell of the batch): y = x W^T + b with x [K, in], K ~ 1e5.

Forward and dX are ordinary GEMMs; the weight gradient dW = dY^T X has a tiny output and a K-long reduction, which
csrc/xty.hip splits over the whole chip (library GEMMs walk K serially: 0.42 ms vs ~0.04 ms at K = 129 600, 128 x 128).
Used by the TransFusion decoder layer's K / V projections and position-embedding MLP (BF/transformer.py:10-23,60-105).
"""
import os

import torch
import torch.nn.functional as F

from . import _lib

MIN_ROWS = 8192 if os.environ.get("BFHIP_LINEAR_ROWS", "1") == "1" else 1 << 62  # below this the plain autograd path is as fast
_DT = {torch.float32: 0, torch.bfloat16: 1}
_WS = {}


# bf16 operands: x^T y over K rows IS the weight gradient of a 1x1 convolution over a [1, K, 1, C] map, and csrc/conv2d.hip's
# weight-gradient kernel (bf16 MFMA, LDS-DMA ring, split over the rows) does it in about a third of the time of csrc/xty.hip's
# fp32-MFMA kernel (K = 129 600, 128 x 128: ~30 vs 90 us).  BFHIP_XTY_CONV=0: always csrc/xty.hip.
XTY_CONV = os.environ.get("BFHIP_XTY_CONV", "1") == "1"
XTY_CONV_MIN_ROWS = int(os.environ.get("BFHIP_XTY_CONV_MIN_ROWS", "4096"))


def xty(x, y):
    """f32[M, N] = x^T y for row-major x [K, M], y [K, N] (same dtype: f32 or bf16)."""
    assert x.dim() == 2 and y.dim() == 2 and x.shape[0] == y.shape[0] and x.dtype == y.dtype and x.dtype in _DT
    x, y = x.contiguous(), y.contiguous()
    K, M = x.shape
    N = y.shape[1]
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    stream = _lib.stream_of(x)
    if (XTY_CONV and x.dtype == torch.bfloat16 and M % 8 == 0 and N % 8 == 0 and K >= XTY_CONV_MIN_ROWS and K * max(M, N) < (1 << 31)
            and x.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0 and _lib.load().bfhip_conv2d_supported(1, K, 1, N, M, 1, 1, 1, 0, 1)):
        nbytes = _lib.call_size("bfhip_conv2d_wgrad_workspace_bytes", 1, K, 1, N, M, 1, 1)
        key = (x.device, stream)
        ws = _WS.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = _WS[key] = torch.empty(max(nbytes, 1 << 24), dtype=torch.uint8, device=x.device)
        # dW[co = M][ci = N] = sum over pixels of dy[pixel][co] * x[pixel][ci]: "dy" = x, "x" = y
        _lib.call("bfhip_conv2d_wgrad", y.data_ptr(), N, x.data_ptr(), M, out.data_ptr(), 1, K, 1, N, M, 1, 1, 1, 0, 1, 0,
                  ws.data_ptr(), ws.numel(), stream)
        return out
    nbytes = _lib.call_size("bfhip_xty_workspace_bytes", K, M, N)
    key = (x.device, stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[key] = torch.empty(max(nbytes, 1 << 24), dtype=torch.uint8, device=x.device)
    _lib.call("bfhip_xty", x.data_ptr(), y.data_ptr(), K, M, N, _DT[x.dtype], out.data_ptr(), ws.data_ptr(), ws.numel(), stream)
    return out


class _LinearRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        y = F.linear(x, weight, bias)  # under autocast this is the bf16 GEMM torch would run anyway
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = dy @ weight.to(dy.dtype)
            if dx.dtype != x.dtype:
                dx = dx.to(x.dtype)
        if ctx.needs_input_grad[1]:
            xs = x if x.dtype == dy.dtype else x.to(dy.dtype)
            dw = xty(dy, xs).to(weight.dtype)           # [out, in] = dY^T X
        if ctx.has_bias and ctx.needs_input_grad[2]:
            from .bn2d import colsum
            db = colsum(dy).to(ctx.bias_dtype)
        return dx, dw, db


def linear_rows(x, weight, bias=None):
    """F.linear for a row-major matrix x [K, in]; for K >= MIN_ROWS on the GPU the weight gradient uses the split-K kernel."""
    if x.is_cuda and x.dim() == 2 and x.shape[0] >= MIN_ROWS and (x.dtype in _DT or torch.is_autocast_enabled("cuda")):
        return _LinearRows.apply(x, weight, bias)
    return F.linear(x, weight, bias)
