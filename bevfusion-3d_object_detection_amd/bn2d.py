"""BatchNorm2d (+ residual) (+ ReLU) for the dense conv stacks, fused in csrc/bn2d.hip when the activation is
channels-last (the layout MIOpen's NHWC kernels produce) and the module is training.

`BatchNorm2dAct` IS an `nn.BatchNorm2d` (same parameters, buffers and state-dict keys as the reference's
`build_norm_layer(dict(type='BN'))` modules), so checkpoints load unchanged; `bn_act()` returns the
`[BatchNorm2dAct(act=True), nn.Identity()]` pair that replaces `[BatchNorm2d, ReLU]` inside an `nn.Sequential`
without shifting the indices of the following layers.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib

FUSED_BN2D = os.environ.get("BFHIP_FUSED_BN2D", "1") == "1"  # A/B switch; the torch path has identical semantics
# residual + ReLU layers fed by a HIP convolution (the bottlenecks' bn3): keep one bit per element for the backward instead of y
RELU_BITS = os.environ.get("BFHIP_BN2D_RELU_BITS", "1") == "1"
_DT = {torch.float32: 0, torch.bfloat16: 1}
_WS = {}


def _apply(x, residual, weight, bias, running_mean, running_var, eps, momentum, relu, partial=None, rows_dev=None):
    """rows_dev (optional int32[1] device tensor): number of ACTIVE rows of a capacity-sized matrix (the rest are zeros)."""
    ext = _lib.torch_ext()
    if (ext is not None and rows_dev is None and weight.dtype == torch.float32
            and (partial is None or partial.dtype == torch.float32)):  # C++ autograd front-end: ~3x less host time per call
        return ext.bn2d(x, residual, weight, bias, running_mean, running_var, eps, momentum, relu, partial, RELU_BITS)
    return _BN2dFunction.apply(x, residual, weight, bias, running_mean, running_var, eps, momentum, relu, partial, rows_dev)


_SIZES = {}


def _ws_bytes(M, C, dt):
    """bfhip_bn2d_workspace_bytes, memoised per shape (0 = unsupported shape)."""
    key = (M, C, dt)
    n = _SIZES.get(key)
    if n is None:
        n = _SIZES[key] = _lib.call_size("bfhip_bn2d_workspace_bytes", M, C, dt)
    return n


def _workspace(device, nbytes, stream):
    key = (device, stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 22), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


COLSUM = os.environ.get("BFHIP_COLSUM", "1") == "1"


def colsum(t):
    """f32[C] = sum over the rows of a dense row-major [M, C] matrix (f32 | bf16): bias gradients (sum of dy over pixels / rows) on
    the BatchNorm statistics kernel; torch's reduction elsewhere (CPU, small or unsupported shapes)."""
    M, C = t.shape
    if (COLSUM and t.is_cuda and t.dtype in _DT and t.is_contiguous() and M >= 4096 and t.data_ptr() % 16 == 0
            and _ws_bytes(M, C, _DT[t.dtype]) > 0):
        out = torch.empty(C, dtype=torch.float32, device=t.device)
        stream = _lib.stream_of(t)
        ws = _workspace(t.device, _ws_bytes(M, C, _DT[t.dtype]), stream)
        _lib.call("bfhip_colsum", t.data_ptr(), M, C, _DT[t.dtype], out.data_ptr(), ws.data_ptr(), ws.numel(), stream)
        return out
    return t.sum(0, dtype=torch.float32)


class _BN2dFunction(torch.autograd.Function):
    """y = act(BN_train(x) [+ residual]); x, residual, y channels-last [N, C, H, W], f32 or bf16."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, eps, momentum, relu, partial=None, rows_dev=None):
        N, C, H, W = x.shape
        M, dt = N * H * W, _DT[x.dtype]
        res = None
        if residual is not None:
            res = residual if residual.dtype == x.dtype else residual.to(x.dtype)
            res = res.contiguous(memory_format=torch.channels_last)
        y = torch.empty_like(x)  # keeps the channels-last strides
        stats = torch.empty(4 * C, dtype=torch.float32, device=x.device)
        stream = _lib.stream_of(x)
        mask = None
        if partial is not None and RELU_BITS and relu and res is not None and dt == 1 and C % 8 == 0:
            # residual + ReLU: the backward needs (y > 0) only -- one bit per element instead of the saved output
            mask = torch.empty((M, C // 8), dtype=torch.uint8, device=x.device)
            _lib.call("bfhip_bn2d_fwd_partials_mask", x.data_ptr(), res.data_ptr(), weight.data_ptr(), bias.data_ptr(), M, C, dt, eps,
                      momentum, 1, running_mean.data_ptr(), running_var.data_ptr(), stats.data_ptr(), y.data_ptr(),
                      partial.data_ptr(), partial.shape[0], _lib.ptr(rows_dev), mask.data_ptr(), stream)
        elif partial is not None:
            # the producing convolution accumulated the column sums in its epilogue (conv2d.py): no statistics pass
            _lib.call("bfhip_bn2d_fwd_partials", x.data_ptr(), _lib.ptr(res), weight.data_ptr(), bias.data_ptr(), M, C, dt, eps,
                      momentum, 1 if relu else 0, running_mean.data_ptr(), running_var.data_ptr(), stats.data_ptr(),
                      y.data_ptr(), partial.data_ptr(), partial.shape[0], _lib.ptr(rows_dev), stream)
        else:
            ws = _workspace(x.device, _ws_bytes(M, C, dt), stream)
            _lib.call("bfhip_bn2d_fwd", x.data_ptr(), _lib.ptr(res), weight.data_ptr(), bias.data_ptr(), M, C, dt, eps, momentum,
                      1 if relu else 0, running_mean.data_ptr(), running_var.data_ptr(), stats.data_ptr(), y.data_ptr(),
                      _lib.ptr(rows_dev), ws.data_ptr(), ws.numel(), stream)
        keep_y = relu and residual is not None and mask is None  # otherwise the ReLU mask is recomputed from x in the backward
        ctx.rows_dev = rows_dev
        ctx.save_for_backward(x, y if keep_y else None, stats, weight, mask)
        ctx.relu, ctx.has_res = relu, residual is not None
        ctx.res_dtype = residual.dtype if residual is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, stats, weight, mask = ctx.saved_tensors
        N, C, H, W = x.shape
        M, dt = N * H * W, _DT[x.dtype]
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dgb = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        stream = _lib.stream_of(x)
        ws = _workspace(x.device, _ws_bytes(M, C, dt), stream)
        if mask is not None:
            _lib.call("bfhip_bn2d_bwd_mask", dy.data_ptr(), x.data_ptr(), mask.data_ptr(), stats.data_ptr(), weight.data_ptr(), M, C,
                      dt, dx.data_ptr(), _lib.ptr(dres), dgb.data_ptr(), _lib.ptr(ctx.rows_dev), ws.data_ptr(), ws.numel(), stream)
        else:
            _lib.call("bfhip_bn2d_bwd", dy.data_ptr(), x.data_ptr(), _lib.ptr(y), stats.data_ptr(), weight.data_ptr(), M, C, dt,
                      1 if ctx.relu else 0, dx.data_ptr(), _lib.ptr(dres), dgb.data_ptr(), _lib.ptr(ctx.rows_dev), ws.data_ptr(),
                      ws.numel(), stream)
        if dres is not None and ctx.res_dtype != dres.dtype:
            dres = dres.to(ctx.res_dtype)
        return dx, dres, dgb[:C].to(weight.dtype), dgb[C:].to(weight.dtype), None, None, None, None, None, None, None


class _LazyBatchCounter:
    """`num_batches_tracked` advanced lazily (it only matters when momentum is None): the per-step `add_(1)` launch of every
    BN layer is folded into a host counter that is flushed into the buffer whenever the state dict is read."""

    _pending_batches = 0

    def _flush_batches(self):
        if self._pending_batches and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(self._pending_batches)
        self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_batches()
        super()._save_to_state_dict(destination, prefix, keep_vars)


class BatchNorm2dAct(_LazyBatchCounter, nn.BatchNorm2d):
    """nn.BatchNorm2d whose forward can also add a residual and apply ReLU (`act=True` makes ReLU the default, for use
    inside nn.Sequential)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, act=False, **kw):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine, track_running_stats=track_running_stats,
                         **kw)
        self.act = act

    def fusable(self, x):
        return (FUSED_BN2D and self.training and x.is_cuda and x.dim() == 4 and x.dtype in _DT and self.affine
                and self.track_running_stats and self.momentum is not None and x.numel() > x.shape[1]
                and x.is_contiguous(memory_format=torch.channels_last)
                and _ws_bytes(x.numel() // x.shape[1], x.shape[1], _DT[x.dtype]) > 0)

    def forward(self, x, residual=None, relu=None):
        relu = self.act if relu is None else relu
        if self.fusable(x):
            self._pending_batches += 1
            partial = getattr(x, "_bfhip_stat_partial", None)  # statistics from the producing conv's epilogue (conv2d.py)
            if partial is not None:
                partial, ptr, version = partial
                # the sums describe the conv's output as it left the kernel: an in-place edit of x since then (add_, mul_, relu_)
                # bumps the version counter, and the attribute must not survive it
                if (ptr != x.data_ptr() or version != x._version or partial.shape[2] != x.shape[1]
                        or partial.shape[0] != -(-(x.numel() // x.shape[1]) // 128)):
                    partial = None
            return _apply(x, residual, self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum,
                          relu, partial)
        self._flush_batches()
        out = super().forward(x)
        if residual is not None:
            out = out + residual
        return F.relu(out) if relu else out

    def extra_repr(self):
        return super().extra_repr() + (", act=ReLU" if self.act else "")


class BatchNormRows(_LazyBatchCounter, nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters / buffers / state-dict keys) for row-major feature matrices [M, C]: in training on a
    supported width the statistics, the affine map and an optional ReLU run in the fused kernels of csrc/bn2d.hip
    (an [M, C] matrix is the channels-last view [M, C, 1, 1]).  [B, C, L] inputs take the torch path."""

    def forward(self, x, relu=False):
        if (FUSED_BN2D and self.training and x.is_cuda and x.dim() == 2 and x.dtype in _DT and self.affine
                and self.track_running_stats and self.momentum is not None and x.shape[0] > 1 and x.is_contiguous()
                and _ws_bytes(x.shape[0], x.shape[1], _DT[x.dtype]) > 0):
            self._pending_batches += 1
            M, C = x.shape
            y = _apply(x.view(M, C, 1, 1), None, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                       self.momentum, relu)
            return y.view(M, C)
        self._flush_batches()
        out = super().forward(x)
        return F.relu(out) if relu else out


def bn_act(num_features, eps=1e-5, momentum=0.1):
    """[BN + ReLU fused, placeholder] in place of [BatchNorm2d, ReLU] of an nn.Sequential (indices preserved)."""
    return [BatchNorm2dAct(num_features, eps=eps, momentum=momentum, act=True), nn.Identity()]
