"""Sparse convolution modules under the names the reference builds by registry
('SubMConv3d', 'SparseConv3d'; mmdet3d/models/layers/spconv/overwrite_spconv/write_spconv2.py:13-39)
and the container types it imports from spconv.pytorch (`SparseConvTensor`, `SparseModule`,
`SparseSequential`; mmdet3d/models/layers/sparse_block.py:11-14, BF/sparse_encoder.py:133).

The arithmetic runs in csrc/spconv.hip.  API surface kept from traveller59/spconv 2.x as the
reference uses it: `SparseConvTensor(features, indices, spatial_shape, batch_size)` with
`.features .indices .spatial_shape .batch_size .indice_dict .replace_feature() .dense()
.find_indice_pair() .shadow_copy()`; convs take (in_channels, out_channels, kernel_size,
stride=, padding=, dilation=, bias=, indice_key=); weight layout (out, k0, k1, k2, in).
"""
import math
import os

import torch
from torch import nn

from . import _lib
from .registry import MODELS


def _triple(v):
    return [int(v)] * 3 if isinstance(v, int) else [int(a) for a in v]


_WS = {}


def _workspace(device, nbytes, tag="ws"):
    """Grow-only scratch, one buffer per (device, current stream, tag): a buffer is only ever used by kernels of the stream
    it was allocated under, so growing it (which frees the old one back to that stream's allocator pool) cannot hand
    memory that side-stream kernels still use to a main-stream tensor."""
    key = (device, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


class IndiceData:
    """Rulebook of one convolution (what spconv keeps in `indice_dict[indice_key]`)."""

    def __init__(self, out_indices, pair_fwd, pair_bwd, n_pairs, is_subm, out_spatial_shape, ksize, stride, padding,
                 dilation, n_out_dev=None, sorted_rows=None):
        self.n_out_dev = n_out_dev          # static capacity mode: true N_out (device i32[1]); out_indices has capacity rows
        self.out_indices = out_indices      # i32[N_out, 4]
        self.pair_fwd = pair_fwd            # i32[KV, N_out]
        self.pair_bwd = pair_bwd            # i32[KV, N_in] (None for SubM: pair_fwd with flipped offsets)
        self.n_pairs = n_pairs              # device i32[64] spread counters; total pairs = n_pairs.sum()
        self.is_subm = is_subm
        self.out_spatial_shape = out_spatial_shape
        self.ksize, self.stride, self.padding, self.dilation = ksize, stride, padding, dilation
        # mask-sorted row orders (tile-level offset skipping): forward table and, for strided convs, backward table.  The
        # rulebook builders hand them over (made by the launches that fill the tables); sort_rows is the stand-alone form.
        if sorted_rows is not None:
            self.mask_fwd, self.perm_fwd, self.mask_bwd, self.perm_bwd = sorted_rows
        else:
            self.mask_fwd, self.perm_fwd = sort_rows(pair_fwd)
            self.mask_bwd, self.perm_bwd = sort_rows(pair_bwd) if pair_bwd is not None else (None, None)


# Mask-sorted row order for the gather-GEMM (tile-level offset skipping); the masks alone are always computed.  Measured at
# the benchmark's sizes (batch 4): the 12 sorts of a pass cost 0.52 ms of launch-bound work and save 0.15 ms of gather-GEMM
# time (forward 1.02 -> 0.98, dgrad 1.00 -> 0.89 ms).  ALONE, the LiDAR pass is faster without them (6.35 vs 7.06 ms,
# tools/sparse_micro.py); inside the full training step, where the branch runs beside the camera stream and the sorts'
# small launches fill gaps while the GEMM savings free CUs, the step is 0.25 ms faster WITH them (34.83 vs 35.08 and 36.03 vs
# 36.28 ms, tools/ab_step.sh, same box) -- hence on by default; BFHIP_SPCONV_SORT=0 for LiDAR-only deployments.
SORT_ROWS = os.environ.get("BFHIP_SPCONV_SORT", "1") == "1"
# weight gradient of layers whose input channel count is not a multiple of 4: zero-pad the input for the MFMA kernel
PAD_WGRAD_INPUT = os.environ.get("BFHIP_SPCONV_PAD_WGRAD", "1") == "1"
# forward of layers with fewer than 16 fp32 input channels: zero-pad to 16 for the MFMA kernels (see _SparseConvFunction)
PAD_NARROW_INPUT = os.environ.get("BFHIP_SPCONV_PAD_INPUT", "1") == "1"


def sort_rows(pairs):
    """(row_mask u32[n], perm i32[n] or None) of a pair table i32[KV, n] (bfhip_rulebook_sort_rows)."""
    kv, n = pairs.shape
    dev = pairs.device
    mask = torch.empty(n, dtype=torch.int32, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev) if SORT_ROWS else None
    if n == 0:
        return mask, perm
    lib = _lib.load()
    ws = _workspace(dev, lib.bfhip_rulebook_sort_rows_workspace_bytes(n, kv), "rule")
    with torch.cuda.device(dev):
        rc = lib.bfhip_rulebook_sort_rows(_lib.ptr(pairs), n, kv, n, _lib.ptr(mask), _lib.ptr(perm), _lib.ptr(ws), ws.numel(),
                                          _lib.stream_of(pairs))
    _lib.check(rc, "rulebook_sort_rows")
    return mask, perm


def _mask_perm(n, dev):
    mask = torch.empty(n, dtype=torch.int32, device=dev)
    return mask, (torch.empty(n, dtype=torch.int32, device=dev) if SORT_ROWS else None)


def build_subm_rulebook(indices, batch_size, spatial_shape, ksize, dilation):
    N = indices.shape[0]
    kv = ksize[0] * ksize[1] * ksize[2]
    dev = indices.device
    pair_fwd = torch.empty((kv, N), dtype=torch.int32, device=dev)
    n_pairs = torch.empty(64, dtype=torch.int32, device=dev)  # 64 spread counters (zeroed by the call); total = sum
    fused = kv <= 32  # row masks + sorted order from the launch that fills the table
    mask, perm = _mask_perm(N, dev) if fused else (None, None)
    lib = _lib.load()
    ws = _workspace(dev, lib.bfhip_rulebook_subm_workspace_bytes(N), "rule")
    with torch.cuda.device(dev):
        rc = lib.bfhip_rulebook_subm(_lib.ptr(indices), N, batch_size, _lib.host_i32(spatial_shape), _lib.host_i32(ksize),
                                     _lib.host_i32(dilation), _lib.ptr(pair_fwd), _lib.ptr(n_pairs), _lib.ptr(mask),
                                     _lib.ptr(perm), _lib.ptr(ws), ws.numel(), _lib.stream_of(indices))
    _lib.check(rc, "rulebook_subm")
    return IndiceData(indices, pair_fwd, None, n_pairs, True, list(spatial_shape), ksize, [1, 1, 1], None, dilation,
                      sorted_rows=(mask, perm, None, None) if fused else None)


def conv_out_shape(spatial_shape, ksize, stride, padding, dilation):
    """(in + 2p - d(k-1) - 1)//s + 1 (projects/SparseConvolution/sparse_conv.py:88-90)."""
    return [(spatial_shape[i] + 2 * padding[i] - dilation[i] * (ksize[i] - 1) - 1) // stride[i] + 1 for i in range(3)]


class StridedPlan:
    """What `prepare_strided_rulebooks` leaves for one strided layer: its geometry, N_out (host int) and the workspace
    holding the counted bitmap / prefix sums that `bfhip_rulebook_sparse_fill` needs."""

    def __init__(self, geo_key, n_out, counts, ws):
        self.geo_key, self.n_out, self.counts, self.ws = geo_key, n_out, counts, ws
        self.n_out_dev = None  # static capacity mode: n_out is a capacity, the true count is this device scalar


def _geo_key(spatial_shape, ksize, stride, padding, dilation):
    return (tuple(spatial_shape), tuple(ksize), tuple(stride), tuple(padding), tuple(dilation))


def prepare_strided_rulebooks(indices, batch_size, spatial_shape, specs, hints=None, static_caps=None, n_in_dev=None):
    """Count the outputs of a CHAIN of strided sparse convs (specs: [(ksize, stride, padding, dilation), ...], each applied to
    the previous one's output coordinates; SubM layers in between do not change coordinates) with ONE host read for all of
    them instead of one per layer.  Output coordinates of a level go into a capped buffer whose true length
    stays on the device and feeds the next level's count.  Returns {geometry key: StridedPlan}; a level whose N_out
    exceeded its cap (and everything after it) is left out and takes the per-layer path.  `hints` (the N_out values of the
    previous forward, if any) tighten the caps to 1.5x: consecutive frames have similar occupancy, and the launches of the
    following level are sized by the cap.

    Static capacity mode (`static_caps`: one row capacity per level, `n_in_dev`: active input rows on the device): NO host
    read at all -- every level's outputs live in a buffer of its capacity, the true counts stay on the device
    (`plan.n_out_dev`) and rows beyond them are inactive (batch index -1).  Returns ({key: plan}, true-count tensor) then."""
    dev = indices.device
    lib = _lib.load()
    stream = _lib.stream_of(indices)
    N0 = indices.shape[0]
    cur_idx, cur_cap, cur_n_dev, shape = indices, N0, n_in_dev, list(spatial_shape)
    levels = []
    if N0 == 0:
        return {}
    with torch.cuda.device(dev):
        for ksize, stride, padding, dilation in specs:
            geo = [_lib.host_i32(shape), _lib.host_i32(ksize), _lib.host_i32(stride), _lib.host_i32(padding),
                   _lib.host_i32(dilation)]
            nbytes = lib.bfhip_rulebook_sparse_workspace_bytes(batch_size, *geo)
            if nbytes == 0:
                break
            out_shape = conv_out_shape(shape, ksize, stride, padding, dilation)
            cells = batch_size * out_shape[0] * out_shape[1] * out_shape[2]
            cap = int(min(cells, 8 * cur_cap, max(4 * N0, 1 << 16)))
            if static_caps is not None:
                cap = int(min(cells, static_caps[len(levels)]))
            elif hints is not None and len(levels) < len(hints) and hints[len(levels)] > 0:
                cap = int(min(cap, hints[len(levels)] * 3 // 2 + 4096))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)  # kept until the layer's fill
            counts = torch.zeros(65, dtype=torch.int32, device=dev)
            _lib.check(lib.bfhip_rulebook_sparse_count(_lib.ptr(cur_idx), cur_cap, _lib.ptr(cur_n_dev), batch_size, *geo,
                                                       _lib.ptr(counts), _lib.ptr(ws), ws.numel(), stream), "rulebook_sparse_count")
            tmp_idx = torch.empty((cap, 4), dtype=torch.int32, device=dev)
            _lib.check(lib.bfhip_rulebook_sparse_out_indices(batch_size, *geo, cap, _lib.ptr(tmp_idx), _lib.ptr(ws), ws.numel(),
                                                             stream), "rulebook_sparse_out_indices")
            levels.append((_geo_key(shape, ksize, stride, padding, dilation), counts, ws, cap))
            cur_idx, cur_cap, cur_n_dev, shape = tmp_idx, cap, counts[0:1], out_shape
    if static_caps is not None:
        plans = {}
        for key, counts, ws, cap in levels:
            plan = StridedPlan(key, cap, counts, ws)
            plan.n_out_dev = counts[0:1]
            plans[key] = plan
        return plans, torch.stack([c[0] for _, c, _, _ in levels]) if levels else None
    if not levels:
        return {}
    n_outs = torch.stack([c[0] for _, c, _, _ in levels]).tolist()  # the one host read
    plans = {}
    for (key, counts, ws, cap), n_out in zip(levels, n_outs):
        if n_out > cap:
            break  # this level (and its successors, counted from a truncated input) falls back to the per-layer path
        plans[key] = StridedPlan(key, int(n_out), counts, ws)
    if hints is not None:
        # the count of the first overflowing level is still exact (its input was complete); the levels after it were counted
        # from a truncated input and say nothing about the frame: no hint (0) for them, the next forward uses the loose caps
        good = min(len(plans) + 1, len(n_outs))
        hints[:] = n_outs[:good] + [0] * (len(n_outs) - good)
    return plans


class CapacityMonitor:
    """Static capacity mode keeps every row count on the device; this watches them WITHOUT a host stall: the true counts of
    a forward are copied to pinned memory behind the forward's kernels, and read at the start of a later forward once
    their event has completed (event.query() is not a synchronising call)."""

    def __init__(self):
        self.pinned = self.event = None
        self.pending = False

    def submit(self, counts_dev):
        if torch.cuda.is_current_stream_capturing():
            return
        n = counts_dev.numel()
        if self.pinned is None or self.pinned.numel() != n:
            self.pinned = torch.empty(n, dtype=torch.int32).pin_memory()
            self.event = torch.cuda.Event()
        self.pinned.copy_(counts_dev, non_blocking=True)
        self.event.record()
        self.pending = True

    def poll(self):
        """The counts of the last submitted forward if they have arrived, else None."""
        if torch.cuda.is_current_stream_capturing():
            return None
        if self.pending and self.event.query():
            self.pending = False
            return self.pinned.tolist()
        return None


CAPACITY_SLACK = float(os.environ.get("BFHIP_CAPACITY_SLACK", "1.25"))


def round_capacity(n):
    """Row capacity for an observed count n: slack x n + 2048, rounded up to 2048 (grow-only at the call sites).  Frame-to-
    frame occupancy of one drive varies by well under 25 %; an overflow is detected (CapacityMonitor), reported and grown."""
    return (int(int(n) * CAPACITY_SLACK) + 2048 + 2047) // 2048 * 2048


def build_sparse_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, plan=None):
    N = indices.shape[0]
    kv = ksize[0] * ksize[1] * ksize[2]
    dev = indices.device
    lib = _lib.load()
    geo = [_lib.host_i32(spatial_shape), _lib.host_i32(ksize), _lib.host_i32(stride), _lib.host_i32(padding),
           _lib.host_i32(dilation)]
    stream = _lib.stream_of(indices)
    with torch.cuda.device(dev):
        if plan is not None:
            ws, counts, n_out = plan.ws, plan.counts, plan.n_out  # counted by prepare_strided_rulebooks: no host read here
        else:
            nbytes = lib.bfhip_rulebook_sparse_workspace_bytes(batch_size, *geo)
            if nbytes == 0:
                raise RuntimeError("SparseConv3d: unsupported geometry")
            ws = _workspace(dev, nbytes, "rule")
            counts = torch.zeros(65, dtype=torch.int32, device=dev)  # [N_out, 64 spread pair counters]
            rc = lib.bfhip_rulebook_sparse_count(_lib.ptr(indices), N, None, batch_size, *geo, _lib.ptr(counts), _lib.ptr(ws),
                                                 ws.numel(), stream)
            _lib.check(rc, "rulebook_sparse_count")
            n_out = int(counts[0].item())  # the one host sync of a strided layer (spconv returns num_act_out too)
        out_indices = torch.empty((n_out, 4), dtype=torch.int32, device=dev)
        pair_fwd = torch.empty((kv, n_out), dtype=torch.int32, device=dev)
        pair_bwd = torch.empty((kv, N), dtype=torch.int32, device=dev)
        fused = kv <= 32 and N > 0 and n_out > 0
        mask_f, perm_f = _mask_perm(n_out, dev) if fused else (None, None)
        mask_b, perm_b = _mask_perm(N, dev) if fused else (None, None)
        sws = _workspace(dev, lib.bfhip_rulebook_sort_rows_workspace_bytes(max(N, n_out), kv), "sort") if fused else None
        rc = lib.bfhip_rulebook_sparse_fill(_lib.ptr(indices), N, batch_size, *geo, n_out, _lib.ptr(out_indices),
                                            _lib.ptr(pair_fwd), _lib.ptr(pair_bwd), _lib.ptr(counts), _lib.ptr(mask_f),
                                            _lib.ptr(perm_f), _lib.ptr(mask_b), _lib.ptr(perm_b), _lib.ptr(sws),
                                            sws.numel() if fused else 0, _lib.ptr(ws), ws.numel(), stream)
    _lib.check(rc, "rulebook_sparse_fill")
    return IndiceData(out_indices, pair_fwd, pair_bwd, counts[1:], False,
                      conv_out_shape(spatial_shape, ksize, stride, padding, dilation), ksize, stride, padding, dilation,
                      n_out_dev=plan.n_out_dev if plan is not None else None,
                      sorted_rows=(mask_f, perm_f, mask_b, perm_b) if fused else None)


def _bf16_ok(weight, transpose):
    cout, cin = weight.shape[0], weight.shape[-1]
    kdim, ndim = (cout, cin) if transpose else (cin, cout)
    return kdim % 8 == 0 and ndim in (16, 32, 64, 128) or (kdim % 8 == 0 and ndim <= 128 and (ndim + 15) // 16 in (1, 2, 4, 8))


# features stored in bf16 between the sparse layers under bf16 autocast (half the gather traffic; the reference's spconv
# keeps fp16 features under AMP).  BFHIP_SPCONV_BF16_FEATURES=0 keeps fp32 storage with bf16-rounded MFMA inputs.
BF16_FEATURES = os.environ.get("BFHIP_SPCONV_BF16_FEATURES", "1") == "1"


def _io16_ok(weight, transpose):
    """bf16 in / bf16 out needs the MFMA path (K % 8 == 0) and 16-byte aligned rows on both sides."""
    cout, cin = weight.shape[0], weight.shape[-1]
    kdim, ndim = (cout, cin) if transpose else (cin, cout)
    return _bf16_ok(weight, transpose) and kdim % 8 == 0 and ndim % 8 == 0


# Debug guard (the round-2 aperture violation): the gather kernels trust pairs / perm / row_mask.  With
# BFHIP_SPCONV_VALIDATE=1 every gather launch is preceded by bfhip_rulebook_validate and a HOST READ of its four counters.
VALIDATE = os.environ.get("BFHIP_SPCONV_VALIDATE", "0") == "1"


def validate_rulebook(pairs, n_rows, n_src, perm=None, row_mask=None):
    """[bad pair entries, perm entries out of range, rows not exactly once in perm, row-mask mismatches] (host list; syncs)."""
    lib = _lib.load()
    dev = pairs.device
    status = torch.empty(4, dtype=torch.int32, device=dev)
    ws = _workspace(dev, lib.bfhip_rulebook_validate_workspace_bytes(n_rows), "validate")
    with torch.cuda.device(dev):
        rc = lib.bfhip_rulebook_validate(_lib.ptr(pairs), pairs.shape[1], pairs.shape[0], n_rows, n_src, _lib.ptr(perm),
                                         _lib.ptr(row_mask), _lib.ptr(status), _lib.ptr(ws), ws.numel(), _lib.stream_of(pairs))
    _lib.check(rc, "rulebook_validate")
    return status.tolist()


def _gemm(inp, weight, pairs, n_rows, transpose, flip, perm=None, row_mask=None, bf16=False, io16=False):
    cout, cin = weight.shape[0], weight.shape[-1]
    kv = pairs.shape[0]
    if VALIDATE:
        bad = validate_rulebook(pairs, n_rows, inp.shape[0], perm, row_mask)
        if any(bad):
            raise RuntimeError("spconv gather operands out of range: [pairs, perm range, perm multiplicity, masks] = %s "
                               "(n_rows %d, ld %d, n_src %d)" % (bad, n_rows, pairs.shape[1], inp.shape[0]))
    out = torch.empty((n_rows, cin if transpose else cout), dtype=torch.bfloat16 if io16 else torch.float32,
                      device=inp.device)
    lib = _lib.load()
    ws = _workspace(inp.device, lib.bfhip_spconv_workspace_bytes(kv, cin, cout), "gemm")
    args = (_lib.ptr(inp), _lib.ptr(weight), _lib.ptr(pairs), pairs.shape[1], kv, n_rows, cin, cout, 1 if transpose else 0,
            1 if flip else 0, _lib.ptr(perm), _lib.ptr(row_mask), _lib.ptr(out))
    tail = (_lib.ptr(ws), ws.numel(), _lib.stream_of(inp))
    with torch.cuda.device(inp.device):
        if bf16 and _bf16_ok(weight, transpose):
            rc = lib.bfhip_spconv_gemm_bf16(*args, 1 if io16 else 0, *tail)
        else:
            assert not io16
            rc = lib.bfhip_spconv_gemm(*args, *tail)
    _lib.check(rc, "spconv_gemm")
    return out


class _SparseConvFunction(torch.autograd.Function):
    """features [N_in, Cin], weight (Cout,k0,k1,k2,Cin) -> [N_out, Cout]."""

    @staticmethod
    def forward(ctx, features, weight, data, n_in):
        # under bf16 autocast the MFMA inputs are bf16 (fp32 accumulate), as the reference's spconv runs in half precision
        # under AMP; with BF16_FEATURES the activations between the layers are stored in bf16 too.  Index paths and the
        # weight gradient's accumulation stay fp32.
        ctx.bf16 = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
        w = weight.contiguous().float()
        io16 = ctx.bf16 and BF16_FEATURES and _io16_ok(w, False) and _io16_ok(w, True)
        ctx.io16 = io16
        features = features.contiguous()
        features = features.to(torch.bfloat16) if io16 else features.float()
        # a narrow fp32 input (the 5 point features of the first layer) is zero-padded to 16 channels so that the layer runs on
        # the fp32 MFMA kernels instead of the one-thread-per-output scalar kernel (forward 82 -> 25 us at batch 4, weight
        # gradient on the 16-channel tiles); the input stays fp32: absolute coordinates do not survive 8 mantissa bits
        ctx.cin = w.shape[-1]
        ctx.gemm_bf16 = ctx.bf16
        if not io16 and PAD_NARROW_INPUT and w.shape[-1] < 16:
            pad = 16 - w.shape[-1]
            features = torch.nn.functional.pad(features, (0, pad))
            w = torch.nn.functional.pad(w, (0, pad))
            ctx.gemm_bf16 = False  # exact-fp32 MFMA, as the scalar kernel it replaces
        out = _gemm(features, w, data.pair_fwd, data.pair_fwd.shape[1], False, False, data.perm_fwd, data.mask_fwd,
                    bf16=ctx.gemm_bf16, io16=io16)
        if ctx.bf16 and BF16_FEATURES and not io16:
            out = out.to(torch.bfloat16)  # narrow first layer (5 -> 16): fp32 kernel, bf16 hand-over
        ctx.save_for_backward(features, w)
        ctx.data = data
        ctx.n_in = n_in
        return out

    @staticmethod
    def backward(ctx, grad_out):
        features, w = ctx.saved_tensors
        data = ctx.data
        io16 = ctx.io16
        grad_out = grad_out.contiguous()
        grad_out = grad_out.to(torch.bfloat16) if io16 else grad_out.float()
        d_feat = d_w = None
        if ctx.needs_input_grad[0]:
            if data.is_subm:
                # SubM: pair_fwd doubles as the backward table with flipped offsets; rows with equal masks stay
                # adjacent under perm_fwd (the flip permutes mask bits), the per-wave masks are recomputed
                d_feat = _gemm(grad_out, w, data.pair_fwd, ctx.n_in, True, True, data.perm_fwd, data.mask_fwd, bf16=ctx.gemm_bf16,
                               io16=io16)
            else:
                d_feat = _gemm(grad_out, w, data.pair_bwd, ctx.n_in, True, False, data.perm_bwd, data.mask_bwd, bf16=ctx.gemm_bf16,
                               io16=io16)
        if ctx.needs_input_grad[1]:
            cout, cin = w.shape[0], w.shape[-1]
            kv = data.pair_fwd.shape[0]
            n_out = data.pair_fwd.shape[1]
            d_w = torch.empty_like(w)
            lib = _lib.load()
            # the MFMA kernel loads 4-channel pieces: an input with 5 point features (the first layer) is padded to 8
            # zero-filled channels here rather than sent down the scalar-load kernel (backward of that layer 0.164 -> 0.130 ms at batch 4)
            cin_k, feats_k, dw_k = cin, features, d_w
            if cin % 4 and cout % 4 == 0 and not io16 and PAD_WGRAD_INPUT:
                cin_k = (cin + 3) // 4 * 4
                feats_k = torch.nn.functional.pad(features, (0, cin_k - cin))
                dw_k = torch.empty(w.shape[:-1] + (cin_k,), dtype=w.dtype, device=w.device)
            ws = _workspace(w.device, lib.bfhip_spconv_wgrad_workspace_bytes(kv, cin_k, cout, n_out), "wgrad")
            with torch.cuda.device(w.device):
                rc = lib.bfhip_spconv_wgrad(_lib.ptr(feats_k), _lib.ptr(grad_out), _lib.ptr(data.pair_fwd), n_out, kv,
                                            n_out, cin_k, cout, None, _lib.ptr(dw_k), 1 if io16 else 0, _lib.ptr(ws),
                                            ws.numel(), _lib.stream_of(w))
            _lib.check(rc, "spconv_wgrad")
            if dw_k is not d_w:
                d_w.copy_(dw_k[..., :cin])
        if ctx.cin != w.shape[-1]:  # padded narrow input: drop the pad channels
            d_feat = d_feat[:, :ctx.cin] if d_feat is not None else None
            d_w = d_w[..., :ctx.cin].contiguous() if d_w is not None else None
        return d_feat, d_w, None, None


class _ToBevFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, indices, B, X, Y, Z):
        features = features.contiguous().float()
        N, C = features.shape
        out = torch.empty((B, C * Z, X, Y), dtype=torch.float32, device=features.device)
        with torch.cuda.device(features.device):
            rc = _lib.load().bfhip_sparse_to_bev(_lib.ptr(features), _lib.ptr(indices), N, C, B, X, Y, Z, _lib.ptr(out),
                                                 _lib.stream_of(features))
        _lib.check(rc, "sparse_to_bev")
        ctx.save_for_backward(indices)
        ctx.shape = (N, C, B, X, Y, Z)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (indices,) = ctx.saved_tensors
        N, C, B, X, Y, Z = ctx.shape
        grad_out = grad_out.contiguous()
        g = torch.empty((N, C), dtype=torch.float32, device=grad_out.device)
        with torch.cuda.device(grad_out.device):
            rc = _lib.load().bfhip_bev_to_sparse(_lib.ptr(grad_out), _lib.ptr(indices), N, C, B, X, Y, Z, _lib.ptr(g),
                                                 _lib.stream_of(grad_out))
        _lib.check(rc, "bev_to_sparse")
        return g, None, None, None, None, None


class _ToBevChannelsLastFunction(torch.autograd.Function):
    """[N, C] features -> [B, C*Z, X, Y] with channels-last strides (NHWC memory), f32 or bf16; the gradient (a channel
    slice of a wider channels-last tensor included) is gathered in place."""

    @staticmethod
    def forward(ctx, features, indices, B, X, Y, Z, dtype):
        features = features.contiguous().float()
        N, C = features.shape
        out = torch.empty((B, X, Y, C * Z), dtype=dtype, device=features.device)
        _lib.call("bfhip_sparse_to_bev_nhwc", _lib.ptr(features), _lib.ptr(indices), N, C, B, X, Y, Z,
                  1 if dtype == torch.bfloat16 else 0, _lib.ptr(out), _lib.stream_of(features))
        ctx.save_for_backward(indices)
        ctx.shape = (N, C, B, X, Y, Z)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_out):
        (indices,) = ctx.saved_tensors
        N, C, B, X, Y, Z = ctx.shape
        if grad_out.dtype not in (torch.float32, torch.bfloat16):
            grad_out = grad_out.float()
        if grad_out.stride(1) != 1:
            grad_out = grad_out.contiguous(memory_format=torch.channels_last)
        g = torch.empty((N, C), dtype=torch.float32, device=grad_out.device)
        _lib.call("bfhip_bev_nhwc_to_sparse", _lib.ptr(grad_out), grad_out.stride(0), grad_out.stride(2), grad_out.stride(3),
                  1 if grad_out.dtype == torch.bfloat16 else 0, _lib.ptr(indices), N, C, Z, _lib.ptr(g),
                  _lib.stream_of(grad_out))
        return g, None, None, None, None, None, None


class SparseConvTensor:
    """features f32[N,C], indices i32[N,4] = (batch, x, y, z) in this fork's axis order
    (BF/sparse_encoder.py:133), spatial_shape [X,Y,Z], batch_size."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None, indice_dict=None,
                 benchmark=False):
        assert features.dim() == 2 and indices.dim() == 2 and indices.shape[1] == 4
        assert indices.dtype == torch.int32, "indices must be int32"
        self.features = features
        self.indices = indices.contiguous()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = indice_dict if indice_dict is not None else {}
        self.benchmark = benchmark
        self._auto_rulebooks = {}  # SubM rulebooks keyed by (ksize, dilation); valid while indices are unchanged
        # static capacity mode: device i32[1] number of ACTIVE rows (a prefix); the other rows have batch index -1 and zero
        # features and are skipped by every index-driven kernel.  None: every row is active (exact sizes).
        self.n_valid = None

    def replace_feature(self, feature):
        new = self.shadow_copy()
        new.features = feature
        return new

    def shadow_copy(self):
        t = SparseConvTensor(self.features, self.indices, self.spatial_shape, self.batch_size,
                             indice_dict=self.indice_dict, benchmark=self.benchmark)
        t._auto_rulebooks = self._auto_rulebooks
        t.n_valid = self.n_valid
        return t

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    @property
    def spatial_size(self):
        return math.prod(self.spatial_shape)

    def dense(self, channels_first=True):
        """[B, C, X, Y, Z] (channels_first) like spconv's .dense()."""
        bev = self.to_bev()  # [B, C*Z, X, Y]
        B, X, Y, Z = self.batch_size, *self.spatial_shape
        C = self.features.shape[1]
        out = bev.view(B, C, Z, X, Y).permute(0, 1, 3, 4, 2)
        return out if channels_first else out.permute(0, 2, 3, 4, 1)

    def to_bev(self, channels_last=False, dtype=None):
        """dense() + permute(0,1,4,2,3) + view(B, C*Z, X, Y) in one kernel (BF/sparse_encoder.py:147-151).
        channels_last=True returns the same [B, C*Z, X, Y] tensor with channels-last strides (what the NHWC conv
        kernels of the BEV backbone consume), optionally already in bf16."""
        X, Y, Z = self.spatial_shape
        if channels_last:
            return _ToBevChannelsLastFunction.apply(self.features, self.indices, self.batch_size, X, Y, Z,
                                                    dtype or torch.float32)
        return _ToBevFunction.apply(self.features, self.indices, self.batch_size, X, Y, Z)


FUSED_BN1D = os.environ.get("BFHIP_FUSED_BN1D", "1") == "1"  # A/B switch; the torch path has identical semantics


class _BN1dFunction(torch.autograd.Function):
    """y = act(BN_train(x) [+ residual]) on f32[N, C] (csrc/bn1d.hip)."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, eps, momentum, relu, rows_dev=None):
        x = x.contiguous()
        N, C = x.shape
        ctx.rows_dev = rows_dev
        res = residual.contiguous() if residual is not None else None
        y = torch.empty_like(x)
        stats = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        lib = _lib.load()
        ws = _workspace(x.device, lib.bfhip_bn1d_workspace_bytes(N, C), "bn1d")
        with torch.cuda.device(x.device):
            rc = lib.bfhip_bn1d_fwd(_lib.ptr(x), _lib.ptr(res), _lib.ptr(weight), _lib.ptr(bias), N, C, float(eps),
                                    float(momentum), 1 if relu else 0, _lib.ptr(running_mean), _lib.ptr(running_var),
                                    _lib.ptr(stats), _lib.ptr(y), _lib.ptr(rows_dev), _lib.ptr(ws), ws.numel(),
                                    _lib.stream_of(x))
        _lib.check(rc, "bn1d_fwd")
        ctx.save_for_backward(x, y, stats, weight)
        ctx.relu = relu
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, stats, weight = ctx.saved_tensors
        N, C = x.shape
        dy = dy.contiguous().float()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dgb = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        lib = _lib.load()
        ws = _workspace(x.device, lib.bfhip_bn1d_workspace_bytes(N, C), "bn1d")
        with torch.cuda.device(x.device):
            rc = lib.bfhip_bn1d_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(x), _lib.ptr(stats), _lib.ptr(weight), N, C,
                                    1 if ctx.relu else 0, _lib.ptr(dx), _lib.ptr(dres), _lib.ptr(dgb), _lib.ptr(ctx.rows_dev),
                                    _lib.ptr(ws), ws.numel(), _lib.stream_of(x))
        _lib.check(rc, "bn1d_bwd")
        return dx, dres, dgb[:C], dgb[C:], None, None, None, None, None, None


class BatchNorm1dAct(nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters / buffers / state_dict keys) whose forward can also add a residual and apply
    ReLU; in training mode on fp32 CUDA features of a supported width it runs the fused HIP kernels.  `num_batches_tracked`
    is advanced lazily like bn2d._LazyBatchCounter's (a host counter flushed into the buffer when the state dict is read)."""

    _pending_batches = 0

    def _flush_batches(self):
        if self._pending_batches and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(self._pending_batches)
        self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_batches()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def forward(self, x, residual=None, relu=False, rows_dev=None):
        """rows_dev (static capacity mode): device i32[1] count of ACTIVE rows; the rows beyond it are zeros on entry."""
        C = x.shape[1] if x.dim() == 2 else 0
        if (FUSED_BN1D and self.training and x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 and x.shape[0] > 1
                and self.affine and self.track_running_stats and self.momentum is not None and C % 8 == 0):
            # bf16 feature matrices: the channels-last BN kernels (csrc/bn2d.hip) on the [N, C, 1, 1] view
            from . import bn2d
            self._pending_batches += 1
            N = x.shape[0]
            res = residual.contiguous().view(N, C, 1, 1) if residual is not None else None
            y = bn2d._apply(x.contiguous().view(N, C, 1, 1), res, self.weight, self.bias, self.running_mean, self.running_var,
                            self.eps, self.momentum, relu, rows_dev=rows_dev)
            return y.view(N, C)
        fused = (FUSED_BN1D and self.training and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] > 1
                 and self.affine and self.track_running_stats and self.momentum is not None
                 and C % 4 == 0 and C <= 256 and 256 % C == 0)
        if fused:
            self._pending_batches += 1
            ext = _lib.torch_ext()
            if ext is not None and rows_dev is None:
                return ext.bn1d(x, residual, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                self.momentum, relu)
            return _BN1dFunction.apply(x, residual, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                       self.momentum, relu, rows_dev)
        if rows_dev is not None:
            raise RuntimeError("capacity-sized feature matrices need the fused BatchNorm kernels (training mode, supported width)")
        self._flush_batches()
        out = super().forward(x)
        if residual is not None:
            out = out + residual
        return torch.relu(out) if relu else out


class SparseModule(nn.Module):
    """Base class of modules that take and return a SparseConvTensor.

    Checkpoint import (write_spconv2.py:33-34,43-74): state dicts written by spconv 2.x carry module version 2 and
    store conv kernels as (out, k0, k1, k2, in) — this package's layout; anything older (MMCV spconv 1.x,
    version None/1) stores (k0, k1, k2, in, out) and is rotated last-dim-first on load.
    """

    _version = 2

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        if local_metadata.get("version", None) != 2:
            for name, param in self._parameters.items():
                key = prefix + name
                if param is None or key not in state_dict:
                    continue
                src = state_dict[key]
                if src.dim() >= 1:  # same rule for every tensor of the module, as the reference applies it
                    state_dict[key] = src.permute(src.dim() - 1, *range(src.dim() - 1))
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)


def is_spconv_module(m):
    return isinstance(m, SparseModule)


class SparseSequential(SparseModule):
    """Sequential that applies plain nn modules to `.features` (as spconv's SparseSequential does)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], dict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        self.add_module(name if name is not None else str(len(self._modules)), module)

    def forward(self, input):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            if is_spconv_module(module):
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    if isinstance(module, BatchNorm1dAct) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                        input = input.replace_feature(module(input.features, relu=True, rows_dev=input.n_valid))  # BN + ReLU
                        i += 1
                    elif isinstance(module, BatchNorm1dAct):
                        input = input.replace_feature(module(input.features, rows_dev=input.n_valid))
                    else:
                        input = input.replace_feature(module(input.features))
            else:
                input = module(input)
            i += 1
        return input


class SparseConvolution(SparseModule):
    """Common base of SubMConv3d / SparseConv3d (spconv.pytorch.conv.SparseConvolution)."""

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 algo=None, fp32_accum=None, name=None):
        super().__init__()
        assert ndim == 3 and groups == 1 and not transposed and not inverse, "only 3-D forward convs are implemented"
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.padding = _triple(padding)
        self.dilation = _triple(dilation)
        self.subm = subm
        self.indice_key = indice_key
        self.conv1x1 = all(k == 1 for k in self.kernel_size)
        # (out, k0, k1, k2, in): the on-disk layout of spconv 2.x checkpoints (write_spconv2.py:50-51)
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # spconv: kaiming_uniform_(a=sqrt(5)) with fan_in = in_channels * kernel volume
        fan_in = self.in_channels * math.prod(self.kernel_size)
        bound = math.sqrt(6.0 / ((1 + 5.0) * fan_in))
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            b = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -b, b)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, subm={self.subm}, indice_key={self.indice_key}")

    def _rulebook(self, input):
        data = input.find_indice_pair(self.indice_key)
        if data is not None:
            return data
        if self.subm:
            auto_key = (tuple(self.kernel_size), tuple(self.dilation))
            data = input._auto_rulebooks.get(auto_key)
            if data is None:
                data = build_subm_rulebook(input.indices, input.batch_size, input.spatial_shape, self.kernel_size,
                                           self.dilation)
                input._auto_rulebooks[auto_key] = data
        else:
            plans = input.indice_dict.get("_strided_plans") or {}
            plan = plans.get(_geo_key(input.spatial_shape, self.kernel_size, self.stride, self.padding, self.dilation))
            data = build_sparse_rulebook(input.indices, input.batch_size, input.spatial_shape, self.kernel_size,
                                         self.stride, self.padding, self.dilation, plan=plan)
        if self.indice_key is not None:
            input.indice_dict[self.indice_key] = data
        return data

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        assert input.features.shape[1] == self.in_channels, "channel size mismatch"
        _lib.require_cuda(input.features, "features")
        if input.n_valid is not None and self.bias is not None:
            raise RuntimeError("static capacity mode needs bias-free sparse convolutions (inactive rows must stay zero)")
        data = self._rulebook(input)
        n_in = input.features.shape[0]
        out_features = _SparseConvFunction.apply(input.features, self.weight, data, n_in)
        if self.bias is not None:
            out_features = out_features + self.bias
        if self.subm:
            return input.replace_feature(out_features)
        out = SparseConvTensor(out_features, data.out_indices, data.out_spatial_shape, input.batch_size,
                               indice_dict=input.indice_dict, benchmark=input.benchmark)
        out.n_valid = data.n_out_dev
        return out


@MODELS.register_module()
class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key)


@MODELS.register_module()
class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, False,
                         indice_key=indice_key)


def replace_feature(out, new_features):
    """mmdet3d/models/layers/sparse_block.py:17-24"""
    return out.replace_feature(new_features)
