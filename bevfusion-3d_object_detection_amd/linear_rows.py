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


def xty(x, y):
    """f32[M, N] = x^T y for row-major x [K, M], y [K, N] (same dtype: f32 or bf16)."""
    assert x.dim() == 2 and y.dim() == 2 and x.shape[0] == y.shape[0] and x.dtype == y.dtype and x.dtype in _DT
    x, y = x.contiguous(), y.contiguous()
    K, M = x.shape
    N = y.shape[1]
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    stream = _lib.stream_of(x)
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
            db = dy.sum(0, dtype=torch.float32).to(ctx.bias_dtype)
        return dx, dw, db


def linear_rows(x, weight, bias=None):
    """F.linear for a row-major matrix x [K, in]; for K >= MIN_ROWS on the GPU the weight gradient uses the split-K kernel."""
    if x.is_cuda and x.dim() == 2 and x.shape[0] >= MIN_ROWS and (x.dtype in _DT or torch.is_autocast_enabled("cuda")):
        return _LinearRows.apply(x, weight, bias)
    return F.linear(x, weight, bias)
