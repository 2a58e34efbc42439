"""Dense 2-D convolution on channels-last bf16 activations through csrc/conv2d.hip (implicit GEMM on the matrix cores,
forward + data gradient + weight gradient), behind `torch.nn.Conv2d`'s interface.

`Conv2d` IS an `nn.Conv2d` (same parameters / state-dict keys as the reference's `build_conv_layer(dict(type='Conv2d'))`
layers: ConvFuser BF/bevfusion_head.py:26-38, SECOND mmdet3d/models/backbones/second.py:27-95, SECONDFPN
necks/second_fpn.py:30-94, shared_conv BF/bevfusion_head.py:95-102, depthnet / downsample BF/depth_lss.py:592-620,
GeneralizedLSSFPN BF/bevfusion_necks.py:50-72).  Its forward takes the HIP kernels when the call is one they serve -- a
CUDA tensor under bf16 autocast (or already bf16), groups = 1, zero padding, channel counts that are multiples of 8 -- and
torch's own convolution otherwise (fp32 islands, CPU tensors, odd channel counts).  `BFHIP_CONV2D=0` switches the HIP path
off (A/B runs).

In training, a bias-free convolution also emits the per-row-block column sums / sums of squares of its fp32 accumulators;
they ride on the output tensor (`y._bfhip_stat_partial`) and the fused BatchNorm that follows (bn2d.BatchNorm2dAct) starts
from them instead of re-reading the activation for its statistics pass.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib

ENABLED = os.environ.get("BFHIP_CONV2D", "1") == "1"
# C++ autograd front-end of the convolution Functions (csrc/torch_binding.cpp: conv2d, lib_conv2d, defer_wgrad): the same entry
# points of the C ABI in the same order, without ~20-30 us of interpreter time per call and direction.  BFHIP_CONV_EXT=0 (or
# conv2d.CONV_EXT = False, what tools that spy on the Python Functions set) keeps the Python classes below.
CONV_EXT = os.environ.get("BFHIP_CONV_EXT", "1") == "1"


def _conv_ext():
    if not CONV_EXT or WGRAD_SIDE_STREAM:
        return None
    ext = _lib.torch_ext()
    if ext is None or not hasattr(ext, "conv2d"):
        return None
    ext.set_wgrad_grouped(bool(WGRAD_GROUPED))
    return ext


def _cached_wt(weight):
    """The transposed copy TransposedWeights keeps on a parameter, while the parameter has not changed since the refresh."""
    cached = getattr(weight, "_bfhip_wt", None)
    if cached is not None and cached[1] == weight._version and cached[2] == weight.data_ptr():
        return cached[0]
    return None
MIN_PIXELS = int(os.environ.get("BFHIP_CONV2D_MIN_PIXELS", "2048"))  # tiny maps: the library's small-problem kernels win
# layers with fewer input channels stay on the library: the one such layer of the model (dtransform 8 -> 32, 5x5 stride 4 on the
# 256 x 704 depth images) has an 8-column data gradient over 4.3 M rows -- 0.30 ms on 64-column tiles; 33.43 vs 33.70 ms per step
MIN_CIN = int(os.environ.get("BFHIP_CONV2D_MIN_CIN", "16"))
# residual blocks: the identity branch's gradient is added inside the data gradient of the block's first conv (forward_fork)
FORK = os.environ.get("BFHIP_CONV_FORK", "1") == "1"
_WS = {}

# Weight gradients on their own HIP stream (opt-in, BFHIP_WGRAD_SIDE_STREAM=1; bench.py switches it on and joins after the
# backward): dW of a layer is a leaf of the backward graph -- nothing downstream waits for it until the optimizer -- and its
# kernel is matrix-core / LDS bound, while the chain it would otherwise sit in (BatchNorm backward -> data gradient -> BatchNorm
# backward ...) is dominated by HBM-bound BatchNorm passes and small launches that leave most CUs idle.  The caller MUST call
# `wgrad_join()` after `backward()` and before anything reads a weight gradient (optimizer, clipping, gradient exchange).
WGRAD_SIDE_STREAM = os.environ.get("BFHIP_WGRAD_SIDE_STREAM", "0") == "1"
_SIDE = {}
_KEEP = []  # operands of in-flight side-stream launches (kept alive until the join instead of record_stream per tensor)


def _wgrad_stream(device):
    s = _SIDE.get(device)
    if s is None:
        s = _SIDE[device] = torch.cuda.Stream(device=device)
    return s


def wgrad_join():
    """Make the current stream wait for every weight gradient launched on the side stream; release their operands."""
    for dev, side in _SIDE.items():
        torch.cuda.current_stream(dev).wait_stream(side)
    _KEEP.clear()


# Grouped weight gradients (default on, BFHIP_WGRAD_GROUPED=0 / conv2d.WGRAD_GROUPED = False: every layer launches its own):
# inside a backward pass `_launch_wgrad` only COLLECTS (x, dy, weight) and returns no gradient; a callback queued on the autograd
# engine runs when the pass ends (before `backward()` returns, on the caller's streams) and computes dW of all collected layers
# with one launch per tile shape plus one slab-sum launch (csrc/conv2d.hip: conv_wgrad_group_kernel), then stores / accumulates
# `weight.grad` itself -- what AccumulateGrad would have done.  Consequences: x and dy of every layer live until the end of the
# pass (a few GB at batch 4 beside 288 GB of HBM); a weight with tensor hooks keeps the per-layer launch; so does a pass with an
# explicit input list (`torch.autograd.grad(..., inputs)`, `backward(inputs=...)`: the engine captures those gradients from the
# graph and must not touch .grad) when the C++ front-end is in use -- the Python-only path (BFHIP_TORCH_EXT=0) cannot see that and
# `torch.autograd.grad(..., weight)` then finds no gradient for the weight (it raises "appears to not have been used"): switch the
# grouping off for such calls there.  Switch it off, too, under torch's DistributedDataParallel, whose bucket all-reduce is driven by the AccumulateGrad
# hooks this path never reaches (grad_sync.FlatGradAllReduce, which runs after the pass, is fine).  Not used while a HIP graph is being captured (the table upload is host memory of this step) or
# with the side-stream option above.
WGRAD_GROUPED = os.environ.get("BFHIP_WGRAD_GROUPED", "1") == "1"
_PENDING = {}      # backward pass (graph task id) -> [(x, dy, weight, row of the layer table, producing stream)]
_GROUPABLE = {}    # geometry -> bool
_GROUP_STATE = {}  # device -> _WgradGroupState
_LAYER_DT = None


class _WgradGroupState:
    """Per device: two pinned images of the group table (alternating: a copy may still be in flight), its device copy, the slabs."""

    def __init__(self, device):
        self.device = device
        self.host = [None, None]
        self.dev = None
        self.slab = None
        self.flip = 0

    def tables(self, nbytes):
        if self.dev is None or self.dev.numel() < nbytes:
            cap = max(int(nbytes) * 2, 1 << 16)
            self.host = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(2)]
            self.dev = torch.empty(cap, dtype=torch.uint8, device=self.device)
        self.flip ^= 1
        return self.host[self.flip], self.dev

    def slabs(self, nbytes):
        if self.slab is None or self.slab.numel() < nbytes:
            self.slab = None
            self.slab = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return self.slab


def _layer_dtype():
    global _LAYER_DT
    if _LAYER_DT is None:
        import numpy as np
        _LAYER_DT = np.dtype([("x", "<u8"), ("dy", "<u8"), ("dw", "<u8")] + [(k, "<i4") for k in (
            "ldx", "ldg", "N", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad", "dil", "dw_bf16", "reserved")])
        assert _LAYER_DT.itemsize == 80  # include/bevfusion_hip.h: bfhip_wgrad_layer
    return _LAYER_DT


def _defer_wgrad(x, dy, weight, stride, pad, dil):
    """True when the layer's weight gradient was queued for the grouped launch at the end of the running backward pass."""
    N, Cin, H, W = x.shape
    Cout, _, KH, KW = weight.shape
    key = (N, H, W, Cin, Cout, KH, KW, stride, pad, dil)
    ok = _GROUPABLE.get(key)
    if ok is None:
        ok = _GROUPABLE[key] = bool(_lib.load().bfhip_conv2d_wgrad_groupable(*key))
    # only a leaf parameter without tensor hooks: the gradient of anything else (a cast copy of a master weight, a hooked tensor)
    # has to travel on through the graph
    if (not ok or not weight.is_leaf or weight._backward_hooks or weight.dtype not in (torch.bfloat16, torch.float32)
            or torch.cuda.is_current_stream_capturing()):
        return False
    ldx, ldg = _nhwc_view(x), _nhwc_view(dy)
    if ldx is None or ldg is None:
        return False
    ext = _conv_ext()
    if ext is not None:  # one list per pass, whichever front-end a layer went through
        return bool(ext.defer_wgrad(x, dy, weight, stride, pad, dil))
    # one list per backward pass (graph task): a pass that died with an exception leaves its list behind without ever running
    # its callback -- the next pass must neither inherit those records nor skip queuing its own callback; a re-entrant pass
    # (checkpointing) has its own id and its own callback
    tid = torch._C._current_graph_task_id()
    pend = _PENDING.get(tid)
    if pend is None:
        if tid < 0:
            return False  # not inside a backward pass of the engine
        try:
            torch.autograd.Variable._execution_engine.queue_callback(lambda: _flush_wgrads(tid))
        except RuntimeError:
            return False
        for old in [k for k in _PENDING if k < tid - 8]:
            del _PENDING[old]
        pend = _PENDING[tid] = []
    row = (x.data_ptr(), dy.data_ptr(), 0, ldx, ldg, N, H, W, Cin, Cout, KH, KW, stride, pad, dil,
           1 if weight.dtype == torch.bfloat16 else 0, 0)
    pend.append((x, dy, weight, row, _lib.stream_of(x)))
    return True


def _flush_wgrads(tid):
    """End of backward pass `tid`: dW of every collected layer in one group per device; stores / accumulates weight.grad."""
    import ctypes

    import numpy as np
    pend = _PENDING.pop(tid, None)
    if not pend:
        return
    by_dev = {}
    for e in pend:
        by_dev.setdefault(e[0].device, []).append(e)
    lib = _lib.load()
    with torch.no_grad():
        for dev, entries in by_dev.items():
            st = _GROUP_STATE.get(dev)
            if st is None:
                st = _GROUP_STATE[dev] = _WgradGroupState(dev)
            cur = torch.cuda.current_stream(dev)
            raw = cur.cuda_stream
            n = len(entries)
            dws = []
            for x, dy, weight, row, s in entries:
                if s != raw:  # produced on another stream than the one the group runs on (the engine has already joined them)
                    x.record_stream(cur)
                    dy.record_stream(cur)
                Cout, Cin, KH, KW = weight.shape
                dws.append(torch.empty((Cout, KH, KW, Cin), dtype=weight.dtype, device=dev).permute(0, 3, 1, 2))
            layers = np.array([e[3] for e in entries], dtype=_layer_dtype())
            layers["dw"] = [d.data_ptr() for d in dws]
            nbytes = int(lib.bfhip_conv2d_wgrad_group_table_bytes(n))
            host, table = st.tables(nbytes)
            slab_bytes = ctypes.c_size_t(0)
            _lib.call("bfhip_conv2d_wgrad_group_plan", layers.ctypes.data, n, 0, host.data_ptr(), nbytes, ctypes.byref(slab_bytes))
            slab = st.slabs(slab_bytes.value)
            with torch.cuda.device(dev):
                table[:nbytes].copy_(host[:nbytes], non_blocking=True)
                _lib.call("bfhip_conv2d_wgrad_group_launch", host.data_ptr(), table.data_ptr(), slab.data_ptr(), slab.numel(), raw)
            for (x, dy, weight, row, s), dw in zip(entries, dws):
                if weight.grad is None:
                    weight.grad = dw
                else:
                    weight.grad.add_(dw)


def _launch_wgrad(x, dy, weight, stride, pad, dil):
    """dW [Cout, Cin, KH, KW] (channels-last memory) of a convolution; on the side stream when WGRAD_SIDE_STREAM is set.
    (None, None) when the layer joined the grouped launch at the end of the backward pass (WGRAD_GROUPED)."""
    N, Cin, H, W = x.shape
    Cout, _, KH, KW = weight.shape
    OH, OW = dy.shape[2], dy.shape[3]
    if WGRAD_GROUPED and not WGRAD_SIDE_STREAM and _defer_wgrad(x, dy, weight, stride, pad, dil):
        return None, None
    lib = _lib.load()
    out_bf16 = weight.dtype == torch.bfloat16
    side = None
    # only when autograd will merely STORE the result (weight.grad is None: AccumulateGrad takes the tensor as it is, no
    # kernel); a gradient that is accumulated into an existing .grad is read by an add on the main stream right away
    if WGRAD_SIDE_STREAM and getattr(weight, "grad", None) is None:
        main = torch.cuda.current_stream(x.device)
        side = _wgrad_stream(x.device)
        side.wait_stream(main)   # dy and x were produced by work already queued on the main stream
        _KEEP.append((x, dy))
    with torch.cuda.stream(side) if side is not None else _NullCtx():
        stream = _lib.stream_of(x)
        dw = torch.empty((Cout, KH, KW, Cin), dtype=weight.dtype if out_bf16 else torch.float32, device=x.device).permute(0, 3, 1, 2)
        ws = _workspace(x.device, lib.bfhip_conv2d_wgrad_workspace_bytes(N, OH, OW, Cin, Cout, KH, KW), stream)
        _lib.call("bfhip_conv2d_wgrad", x.data_ptr(), _nhwc_view(x), dy.data_ptr(), _nhwc_view(dy), dw.data_ptr(), N, H, W, Cin,
                  Cout, KH, KW, stride, pad, dil, 1 if out_bf16 else 0, ws.data_ptr(), ws.numel(), stream)
        if dw.dtype != weight.dtype:
            dw = dw.to(weight.dtype)
    # (no reference to dw is kept here: AccumulateGrad only takes a gradient over without a copy when nobody else holds it)
    return dw, side


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _workspace(device, nbytes, stream):
    key = (device, stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _WS[key] = torch.empty(max(int(nbytes), 1 << 24), dtype=torch.uint8, device=device)
    return buf


def _nhwc_view(t):
    """(pointer tensor, pixel pitch) when t [N, C, H, W] is channels-last dense or a channel slice of such a tensor."""
    N, C, H, W = t.shape
    sn, sc, sh, sw = t.stride()
    if sc == 1 and sw >= C and sh == W * sw and (sn == H * W * sw or N == 1) and sw % 8 == 0 and t.data_ptr() % 16 == 0:
        return sw
    return None


def _as_nhwc_bf16(t):
    if t.dtype != torch.bfloat16:
        t = t.to(torch.bfloat16)
    if _nhwc_view(t) is None:
        t = t.contiguous(memory_format=torch.channels_last)
        if _nhwc_view(t) is None:  # C == 1 or W == 1 corner cases of torch's stride normalisation
            t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return t


def _weight_ohwi(w):
    """bf16 [Cout][KH][KW][Cin] memory of a conv weight [Cout, Cin, KH, KW]."""
    if w.dtype != torch.bfloat16:
        w = w.to(torch.bfloat16)
    p = w.permute(0, 2, 3, 1)
    return p if p.is_contiguous() else p.contiguous()


class _Conv2dFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil, emit_stats, dgrad_lib=False, fork=False):
        ctx.set_materialize_grads(False)  # no zero-filled gradient tensor for the (non-differentiable) statistics output
        x = _as_nhwc_bf16(x)
        N, Cin, H, W = x.shape
        Cout, _, KH, KW = weight.shape
        OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
        OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
        w = _weight_ohwi(weight)
        y = torch.empty((N, OH, OW, Cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
        partial = None
        if emit_stats:
            rows = _lib.load().bfhip_conv2d_stat_rows(N, OH, OW)
            partial = torch.empty((rows, 2, Cout), dtype=torch.float32, device=x.device)
        b32 = None
        if bias is not None:
            b32 = bias if bias.dtype == torch.float32 else bias.float()
        _lib.call("bfhip_conv2d_fwd", x.data_ptr(), _nhwc_view(x), w.data_ptr(), _lib.ptr(b32), y.data_ptr(), Cout, N, H, W, Cin,
                  Cout, KH, KW, stride, pad, dil, 0, _lib.ptr(partial), _lib.stream_of(x))
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, pad, dil)
        ctx.dgrad_lib = dgrad_lib
        ctx.bias_dtype = bias.dtype if bias is not None else None
        if partial is not None:
            ctx.mark_non_differentiable(partial)
        ctx.fork = int(fork)
        if fork == 2:
            # third output: x at its even pixels, [N, C, ceil(H/2), ceil(W/2)] (what a stride-2 1x1 shortcut reads).  The gradient
            # of that compact tensor comes back to THIS node and is added at the even pixels by the data gradient's epilogue
            return y, partial, x[:, :, ::2, ::2].contiguous(memory_format=torch.channels_last)
        if fork:
            # third output: the input again, as a second consumer's handle (the identity branch of a residual block).  Its gradient
            # arrives in THIS node's backward, where the data gradient's epilogue adds it -- instead of autograd summing the two
            # gradient paths into x with a separate pass over the tensor
            return y, partial, x.view_as(x)
        return y, partial

    @staticmethod
    def backward(ctx, dy, _dpartial, d_alias=None):
        x, weight = ctx.saved_tensors
        if dy is None:
            if d_alias is not None and ctx.fork == 2:
                full = torch.zeros_like(x)
                full[:, :, ::2, ::2] = d_alias
                d_alias = full
            return d_alias, None, None, None, None, None, None, None, None
        stride, pad, dil = ctx.geom
        N, Cin, H, W = x.shape
        Cout, _, KH, KW = weight.shape
        dy = _as_nhwc_bf16(dy)
        OH, OW = dy.shape[2], dy.shape[3]
        stream = _lib.stream_of(x)
        lib = _lib.load()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if ctx.dgrad_lib:
                dx = _add_grad(_lib_dgrad(dy, x, weight, stride, pad, dil), d_alias, ctx.fork)
            else:
                dx = _hip_dgrad(dy, x, weight, stride, pad, dil, d_alias, 2 if ctx.fork == 2 else 1)
        side = None
        if ctx.needs_input_grad[1]:
            dw, side = _launch_wgrad(x, dy, weight, stride, pad, dil)
        if ctx.bias_dtype is not None and ctx.needs_input_grad[2]:
            if side is not None:  # a leaf like dW: behind the weight gradient on the side stream (dy is kept alive by _KEEP)
                with torch.cuda.stream(side):
                    db = _bias_grad(dy).to(ctx.bias_dtype)
            else:
                db = _bias_grad(dy).to(ctx.bias_dtype)
        return dx, dw, db, None, None, None, None, None, None


def _add_grad(dx, addend, fork):
    """dx + the second gradient path: on dx's own grid (fork 1) or on the grid of its even pixels (fork 2)."""
    if addend is None:
        return dx
    if fork == 2:
        dx[:, :, ::2, ::2] += addend
        return dx
    return dx + addend


def _bias_grad(dy):
    """f32[C] = sum of dy [N, C, H, W] over batch and pixels; a dense channels-last dy is a row-major [N*H*W, C] matrix."""
    N, C, H, W = dy.shape
    if _nhwc_view(dy) == C:
        from .bn2d import colsum
        return colsum(dy.permute(0, 2, 3, 1).reshape(N * H * W, C))
    return dy.sum(dim=(0, 2, 3), dtype=torch.float32)


def _hip_dgrad(dy, x, weight, stride, pad, dil, addend=None, addend_stride=1):
    """dx of a convolution on csrc/conv2d.hip; `addend` (a second gradient into x: same shape, or for addend_stride 2 the shape of
    x[:, :, ::2, ::2]) is added in the kernel's epilogue when the call is one that fuses it (pointwise layer, transposed weight at
    hand, dense bf16 channels-last addend), otherwise by torch."""
    N, Cin, H, W = x.shape
    Cout, _, KH, KW = weight.shape
    stream = _lib.stream_of(x)
    dx = torch.empty((N, H, W, Cin), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    cached = getattr(weight, "_bfhip_wt", None)  # TransposedWeights: (wt, weight._version, weight.data_ptr()) at refresh time
    if cached is not None and cached[1] == weight._version and cached[2] == weight.data_ptr():
        want = (N, Cin, (H + 1) // 2, (W + 1) // 2) if addend_stride == 2 else (N, Cin, H, W)
        fuse = (addend is not None and addend.dtype == torch.bfloat16 and tuple(addend.shape) == want
                and addend.is_contiguous(memory_format=torch.channels_last) and addend.data_ptr() % 16 == 0
                and _lib.load().bfhip_conv2d_dgrad_fuses_addend(KH, KW, stride, pad, 0))
        _lib.call("bfhip_conv2d_dgrad_wt", dy.data_ptr(), _nhwc_view(dy), cached[0].data_ptr(), _lib.ptr(addend) if fuse else None,
                  addend_stride, dx.data_ptr(), Cin, N, H, W, Cin, Cout, KH, KW, stride, pad, dil, 0, stream)
        return dx if fuse else _add_grad(dx, addend, addend_stride)
    ws = _workspace(x.device, _lib.load().bfhip_conv2d_dgrad_workspace_bytes(Cin, Cout, KH, KW), stream)
    _lib.call("bfhip_conv2d_dgrad", dy.data_ptr(), _nhwc_view(dy), _weight_ohwi(weight).data_ptr(), dx.data_ptr(), Cin, N, H,
              W, Cin, Cout, KH, KW, stride, pad, dil, 0, ws.data_ptr(), ws.numel(), stream)
    return _add_grad(dx, addend, addend_stride)


def _lib_dgrad(dy, x, weight, stride, pad, dil):
    w = weight if weight.dtype == torch.bfloat16 else weight.to(torch.bfloat16)
    return torch.ops.aten.convolution_backward(dy, x, w, None, [stride] * 2, [pad] * 2, [dil] * 2, False, [0, 0], 1,
                                               [True, False, False])[0]


class _LibConvHipWgradFunction(torch.autograd.Function):
    """Forward and data gradient by the library convolution (MIOpen / CK through torch), weight gradient by csrc/conv2d.hip:
    for layers where the library's forward is ahead (ResNet-50's 1x1 and 3x3 convolutions) but its weight gradient brings an
    fp32 zero-fill and a cast launch per call (atomic split-K) and is no faster than the HIP one."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, dil, dgrad_hip=False):
        x = _as_nhwc_bf16(x)
        w = weight if weight.dtype == torch.bfloat16 else weight.to(torch.bfloat16)
        with torch.autocast("cuda", enabled=False):
            y = F.conv2d(x, w, None, stride, pad, dil)
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, pad, dil)
        ctx.dgrad_hip = dgrad_hip
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad, dil = ctx.geom
        N, Cin, H, W = x.shape
        Cout, _, KH, KW = weight.shape
        dy = _as_nhwc_bf16(dy)
        OH, OW = dy.shape[2], dy.shape[3]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _hip_dgrad(dy, x, weight, stride, pad, dil) if ctx.dgrad_hip else _lib_dgrad(dy, x, weight, stride, pad, dil)
        if ctx.needs_input_grad[1]:
            dw, _ = _launch_wgrad(x, dy, weight, stride, pad, dil)
        return dx, dw, None, None, None, None


# fp32 convolutions outside autocast (the reference's fp32 islands: the heat-map head, BF/bevfusion_head.py:218) as THREE bf16
# products per multiply on the matrix cores: a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with a = a_hi + a_lo (a_hi = bf16(a),
# a_lo = bf16(a - a_hi)), fp32 accumulation, fp32 result.  Relative error of a product 2^-16 (the dropped a_lo*b_lo term and the
# rounding of the lo parts) -- between exact fp32 and the TF32 (2^-11) the reference's fp32 convolutions run in on its own
# hardware by torch's default (torch.backends.cudnn.allow_tf32).  The fp32 MFMA path of the library convolution peaks at
# 157 TFLOP/s (1.0 ms for the head's 128 -> 128 3x3 layer, forward + both gradients); the three products run at ~3 x the bf16
# kernels' time.  BFHIP_FP32_CONV=lib: the library's exact fp32 (also what `bench.py`'s reference-numerics region uses).
FP32_SPLIT = os.environ.get("BFHIP_FP32_CONV", "split") == "split"
_SPLIT_SCOPE = [0]


class fp32_island:
    """Marks an fp32 island of a MIXED-PRECISION step (`with torch.autocast(enabled=False)` inside a bf16-autocast forward, as
    BF/bevfusion_head.py:218): only inside such a scope do fp32 convolutions take the three-product path.  A model that runs in
    fp32 throughout keeps the library's exact fp32 convolution everywhere (its gradients are held to fp32-grade tolerances by
    tests/test_dense_modules_gpu.py: through a deep BN + ReLU stack a 5e-6 forward error flips enough ReLU masks to show)."""

    def __init__(self, mixed):
        self.mixed = bool(mixed)

    def __enter__(self):
        _SPLIT_SCOPE[0] += self.mixed
        return self

    def __exit__(self, *a):
        _SPLIT_SCOPE[0] -= self.mixed
        return False


def _split3(t_nhwc, P, C, chan_order=None, batch_order=None):
    """fp32 [P, C] dense -> (bf16 [P, 3C] | None, bf16 [3, P, C] | None), blocks hi / lo by the order words (bit k: block k = lo)."""
    chan = torch.empty((P, 3 * C), dtype=torch.bfloat16, device=t_nhwc.device) if chan_order is not None else None
    batch = torch.empty((3, P, C), dtype=torch.bfloat16, device=t_nhwc.device) if batch_order is not None else None
    _lib.call("bfhip_split_bf16x3", t_nhwc.data_ptr(), P, C, _lib.ptr(chan), chan_order or 0, _lib.ptr(batch), batch_order or 0,
              _lib.stream_of(t_nhwc))
    return chan, batch


_HHL, _HLH = 0b100, 0b010   # block orders [hi, hi, lo] and [hi, lo, hi]


class _Conv2dSplitFunction(torch.autograd.Function):
    """fp32 in, fp32 out, three bf16 products per multiply (see FP32_SPLIT).  Concatenation does the bookkeeping: along the
    channels for forward / data gradient (x' = [x_hi, x_hi, x_lo], w' = [w_hi, w_lo, w_hi] is ONE convolution with 3 x the
    channels), along the batch for the weight gradient (a sum over pixels: x'' = [x_hi; x_lo; x_hi], dy'' = [dy_hi; dy_hi; dy_lo])."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil, emit_stats):
        ctx.set_materialize_grads(False)
        N, Cin, H, W = x.shape
        Cout, _, KH, KW = weight.shape
        OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
        OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
        xn = x.permute(0, 2, 3, 1)
        if not xn.is_contiguous():
            xn = xn.contiguous()
        need_w = ctx.needs_input_grad[1]
        x_chan, x_batch = _split3(xn, N * H * W, Cin, _HHL, _HLH if need_w else None)
        wn = weight.detach().permute(0, 2, 3, 1).contiguous().view(Cout * KH * KW, Cin)
        w_chan, _ = _split3(wn, Cout * KH * KW, Cin, _HLH)          # [Cout*KH*KW, 3 Cin] = the OHWI weight of a conv with 3 Cin channels
        y = torch.empty((N, OH, OW, Cout), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
        partial = None
        if emit_stats:
            partial = torch.empty((_lib.load().bfhip_conv2d_stat_rows(N, OH, OW), 2, Cout), dtype=torch.float32, device=x.device)
        b32 = None if bias is None else (bias if bias.dtype == torch.float32 else bias.float())
        _lib.call("bfhip_conv2d_fwd", x_chan.data_ptr(), 3 * Cin, w_chan.data_ptr(), _lib.ptr(b32), y.data_ptr(), Cout, N, H, W, 3 * Cin,
                  Cout, KH, KW, stride, pad, dil, 1, _lib.ptr(partial), _lib.stream_of(x))
        ctx.save_for_backward(x_batch, weight)
        ctx.geom = (stride, pad, dil, N, Cin, H, W)
        ctx.has_bias = bias is not None
        if partial is not None:
            ctx.mark_non_differentiable(partial)
        return y, partial

    @staticmethod
    def backward(ctx, dy, _dpartial):
        x_batch, weight = ctx.saved_tensors
        if dy is None:
            return None, None, None, None, None, None, None
        stride, pad, dil, N, Cin, H, W = ctx.geom
        Cout, _, KH, KW = weight.shape
        OH, OW = dy.shape[2], dy.shape[3]
        lib = _lib.load()
        gn = dy.float().permute(0, 2, 3, 1)
        if not gn.is_contiguous():
            gn = gn.contiguous()
        stream = _lib.stream_of(gn)
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_chan, g_batch = _split3(gn, N * OH * OW, Cout, _HHL if need_x else None, 0b100 if need_w else None)
        dx = dw = db = None
        if need_x:
            # dx = sum over (co, tap) of dy'[.., 3 Cout] * w'' with w'' = [w_hi; w_lo; w_hi] stacked along the output channels
            wn = weight.detach().permute(0, 2, 3, 1).contiguous().view(Cout * KH * KW, Cin)
            _, w_stack = _split3(wn, Cout * KH * KW, Cin, None, _HLH)       # [3, Cout*KH*KW, Cin] = OHWI weight with 3 Cout outputs
            dx = torch.empty((N, H, W, Cin), dtype=torch.float32, device=gn.device).permute(0, 3, 1, 2)
            ws = _workspace(gn.device, lib.bfhip_conv2d_dgrad_workspace_bytes(Cin, 3 * Cout, KH, KW), stream)
            _lib.call("bfhip_conv2d_dgrad", g_chan.data_ptr(), 3 * Cout, w_stack.data_ptr(), dx.data_ptr(), Cin, N, H, W, Cin, 3 * Cout,
                      KH, KW, stride, pad, dil, 1, ws.data_ptr(), ws.numel(), stream)
        if need_w:
            dw = torch.empty((Cout, KH, KW, Cin), dtype=torch.float32, device=gn.device).permute(0, 3, 1, 2)
            ws = _workspace(gn.device, lib.bfhip_conv2d_wgrad_workspace_bytes(3 * N, OH, OW, Cin, Cout, KH, KW), stream)
            _lib.call("bfhip_conv2d_wgrad", x_batch.data_ptr(), Cin, g_batch.data_ptr(), Cout, dw.data_ptr(), 3 * N, H, W, Cin, Cout, KH,
                      KW, stride, pad, dil, 0, ws.data_ptr(), ws.numel(), stream)
            if dw.dtype != weight.dtype:
                dw = dw.to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = gn.sum(dim=(0, 1, 2))
        return dx, dw, db, None, None, None, None


class TransposedWeights:
    """bf16 [Cin][KH][KW][Cout] copies of the weights of every convolution whose data gradient runs on csrc/conv2d.hip,
    refreshed for ALL layers in one launch (`refresh()`: after the optimizer step).  Without it every data gradient starts with
    its own transpose launch (~5 us, 69 per `full` step).  A copy rides on its parameter (`weight._bfhip_wt`) with the
    parameter's version counter and address at refresh time; `_hip_dgrad` uses it only while both still match, so a weight
    edited through torch since (load_state_dict, an in-place op) falls back to the per-call transpose until the next refresh.
    Kernels that write parameters behind torch's back (amp.MasterWeightAdamW's one-table update) must call `refresh()` after."""

    def __init__(self, modules):
        import numpy as np
        self.items = []
        recs = []
        blk = 0
        seg_bytes = _lib.load().bfhip_conv2d_wt_segment_bytes()
        assert seg_bytes == 40, seg_bytes
        for m in modules:
            if not isinstance(m, Conv2d) or m.groups != 1 or not m.weight.is_cuda:
                continue
            if isinstance(m, Conv2dHipWgrad) and m.dgrad != "hip" and not m.cache_wt:
                continue
            w = m.weight
            if w.dtype not in (torch.bfloat16, torch.float32) or not w.permute(0, 2, 3, 1).is_contiguous():
                continue
            Cout, Cin, KH, KW = w.shape
            wt = torch.empty(Cin * KH * KW * Cout, dtype=torch.bfloat16, device=w.device)
            self.items.append((w, wt))
            recs.append((w.data_ptr(), wt.data_ptr(), Cout, KH * KW, Cin, 1 if w.dtype == torch.float32 else 0, blk))
            blk += -(-Cin // 32) * -(-Cout // 32) * KH * KW
        self.blocks = blk
        self.table = None
        if recs:
            dt = np.dtype([("src", "<u8"), ("dst", "<u8"), ("Cout", "<i4"), ("taps", "<i4"), ("Cin", "<i4"), ("f32", "<i4"), ("blk0", "<i8")])
            arr = np.array(recs, dtype=dt)
            self.table = torch.from_numpy(arr.view(np.uint8).copy()).to(self.items[0][0].device)
            self._ptrs = [r[0] for r in recs]
        self.refresh()

    def refresh(self):
        if self.table is None:
            return
        # a parameter whose storage moved (``.to()``, ``.data = ...``) invalidates the table: drop its copy instead of reading a
        # stale address
        if any(w.data_ptr() != p for (w, _), p in zip(self.items, self._ptrs)):
            for w, _ in self.items:
                w._bfhip_wt = None
            self.table = None
            return
        _lib.call("bfhip_conv2d_weight_transpose_batched", self.table.data_ptr(), len(self.items), self.blocks,
                  _lib.stream_of(self.table))
        for w, wt in self.items:
            w._bfhip_wt = (wt, w._version, w.data_ptr())


def _one(v):
    if isinstance(v, (tuple, list)):
        return v[0] if all(a == v[0] for a in v) else None
    return v


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, emit_stats=False, dgrad_lib=False, fork=False):
    """y = conv2d(x, weight, bias) on the HIP path (bf16, channels-last); returns (y, stat_partial | None) and, with `fork`,
    a third output: x again, for a second consumer whose gradient the data gradient's epilogue adds (see _Conv2dFunction)."""
    ext = _conv_ext() if x.is_cuda and x.dim() == 4 and not weight._backward_hooks else None
    if ext is not None:
        out = ext.conv2d(x, weight, bias, int(stride), int(padding), int(dilation), bool(emit_stats), bool(dgrad_lib), int(fork),
                         None if dgrad_lib else _cached_wt(weight))
        partial = out[1] if emit_stats else None
        return (out[0], partial, out[2]) if fork else (out[0], partial)
    return _Conv2dFunction.apply(x, weight, bias, int(stride), int(padding), int(dilation), bool(emit_stats), bool(dgrad_lib),
                                 int(fork))


def lib_conv2d(x, weight, stride, padding, dilation, dgrad_hip=False):
    """Library forward (and data gradient unless `dgrad_hip`), weight gradient on csrc/conv2d.hip (_LibConvHipWgradFunction)."""
    ext = _conv_ext() if x.is_cuda and x.dim() == 4 and not weight._backward_hooks else None
    if ext is not None:
        return ext.lib_conv2d(x, weight, int(stride), int(padding), int(dilation), bool(dgrad_hip),
                              _cached_wt(weight) if dgrad_hip else None)
    return _LibConvHipWgradFunction.apply(x, weight, stride, padding, dilation, dgrad_hip)


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose forward runs csrc/conv2d.hip when it can (module docstring)."""

    def hip_eligible(self, x):
        if not (ENABLED and x.is_cuda and x.dim() == 4 and self.groups == 1 and self.padding_mode == "zeros"
                and not isinstance(self.padding, str)):
            return False
        low = x.dtype == torch.bfloat16 or (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
                                            and x.dtype in (torch.float32, torch.bfloat16))
        if not low or self.weight.dtype not in (torch.bfloat16, torch.float32):
            return False
        s, p, d = _one(self.stride), _one(self.padding), _one(self.dilation)
        if s is None or p is None or d is None:
            return False
        N, Cin, H, W = x.shape
        if N * H * W < MIN_PIXELS or Cin < MIN_CIN:
            return False
        return bool(_lib.load().bfhip_conv2d_supported(N, H, W, Cin, self.out_channels, self.kernel_size[0], self.kernel_size[1],
                                                       s, p, d))

    def split_eligible(self, x):
        """An fp32 convolution outside autocast that the three-product path serves (FP32_SPLIT)."""
        if not (FP32_SPLIT and _SPLIT_SCOPE[0] > 0 and ENABLED and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and self.weight.dtype == torch.float32
                and not torch.is_autocast_enabled("cuda") and self.groups == 1 and self.padding_mode == "zeros"
                and not isinstance(self.padding, str)):
            return False
        s, p, d = _one(self.stride), _one(self.padding), _one(self.dilation)
        if s is None or p is None or d is None:
            return False
        N, Cin, H, W = x.shape
        Cout, KH, KW = self.out_channels, self.kernel_size[0], self.kernel_size[1]
        if N * H * W < MIN_PIXELS or Cin % 8 or Cout % 8 or x.numel() * 3 >= (1 << 31):
            return False
        sup = _lib.load().bfhip_conv2d_supported
        return bool(sup(N, H, W, 3 * Cin, Cout, KH, KW, s, p, d) and sup(N, H, W, Cin, 3 * Cout, KH, KW, s, p, d)
                    and sup(3 * N, H, W, Cin, Cout, KH, KW, s, p, d))

    def forward(self, x):
        if not self.hip_eligible(x):
            if self.split_eligible(x):
                emit = self.training and self.bias is None and torch.is_grad_enabled()
                y, partial = _Conv2dSplitFunction.apply(x, self.weight, self.bias, _one(self.stride), _one(self.padding),
                                                        _one(self.dilation), emit)
                if partial is not None:
                    y._bfhip_stat_partial = (partial, y.data_ptr(), y._version)
                return y
            return super().forward(x)
        emit = self.training and self.bias is None and torch.is_grad_enabled()
        y, partial = conv2d(x, self.weight, self.bias, _one(self.stride), _one(self.padding), _one(self.dilation), emit)
        if partial is not None:
            # picked up by the fused BatchNorm that follows (bn2d.BatchNorm2dAct), which checks that y is still the tensor the
            # sums were taken from: same storage and no in-place edit since (tensor version counter)
            y._bfhip_stat_partial = (partial, y.data_ptr(), y._version)
        return y


class Conv2dHipWgrad(Conv2d):
    """nn.Conv2d (bias-free) whose weight gradient always runs on csrc/conv2d.hip and whose forward / data gradient take the
    library (MIOpen / CK through torch) or the HIP kernel, each on its own (`fwd`, `dgrad`: "lib" | "hip"; default: both on the
    library): for layers where one side's kernel is ahead in one direction only (dense_modules.ResNet50).  A HIP forward
    emits the BatchNorm statistics like `Conv2d`.  Falls back to nn.Conv2d when the call is not one the HIP kernels serve."""

    fwd = "lib"
    dgrad = "lib"
    cache_wt = False  # keep a transposed copy of the weight although `dgrad` is "lib" (a stride-2 shortcut run through forward_unstrided)

    def forward(self, x):
        if self.bias is not None or not self.training or not self.hip_eligible(x):
            return nn.Conv2d.forward(self, x)
        s, p, d = _one(self.stride), _one(self.padding), _one(self.dilation)
        if self.fwd == "hip":
            y, partial = conv2d(x, self.weight, None, s, p, d, torch.is_grad_enabled(), self.dgrad != "hip")
            if partial is not None:
                y._bfhip_stat_partial = (partial, y.data_ptr(), y._version)
            return y
        return lib_conv2d(x, self.weight, s, p, d, self.dgrad == "hip")

    def forward_fork(self, x, subsample=1):
        """(conv(x), x'): x' is x for a second consumer (the identity branch of a residual block); when this layer runs forward and
        data gradient on the HIP kernels, the gradient that comes back through x' is added in the data gradient's epilogue
        instead of by a separate pass of autograd's.  subsample=2: x' is x[:, :, ::2, ::2] (dense), the input of a stride-2 1x1
        shortcut; its (compact) gradient is added at the even pixels.  Any other case: (self(x), x) / (self(x), None)."""
        if (FORK and self.fwd == "hip" and self.dgrad == "hip" and self.bias is None and self.training and torch.is_grad_enabled()
                and x.requires_grad and x.dtype == torch.bfloat16 and x.dim() == 4 and _nhwc_view(x) is not None
                and self.hip_eligible(x)):
            s, p, d = _one(self.stride), _one(self.padding), _one(self.dilation)
            y, partial, alias = conv2d(x, self.weight, None, s, p, d, True, False, 2 if subsample == 2 else 1)
            y._bfhip_stat_partial = (partial, y.data_ptr(), y._version)
            return y, alias
        return self(x), (x if subsample == 1 else None)

    def forward_unstrided(self, x_sub):
        """This (1x1, stride-s, unpadded) layer applied to its input ALREADY subsampled (x[:, :, ::s, ::s], dense): the same values
        as self(x), computed as a stride-1 pointwise convolution entirely on the HIP kernels (forward with BatchNorm statistics,
        data gradient on the compact grid, weight gradient)."""
        assert self.kernel_size == (1, 1) and _one(self.padding) == 0 and self.bias is None
        y, partial = conv2d(x_sub, self.weight, None, 1, 0, 1, torch.is_grad_enabled())
        if partial is not None:
            y._bfhip_stat_partial = (partial, y.data_ptr(), y._version)
        return y
