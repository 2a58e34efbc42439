"""`bev_pool_ext` -- same two functions as the reference's pybind module
(BF/ops/bev_pool/src/bev_pool.cpp:22-87,89-94), backed by the gfx950 kernels in
csrc/bev_pool.hip through the C ABI (bfhip_bev_pool_fwd / bfhip_bev_pool_bwd).

Argument order follows the reference's Python call sites (BF/ops/bev_pool/bev_pool.py:56-65):
(x, geom_feats, interval_lengths, interval_starts, b, d, h, w).  Unlike the reference (raw
data_ptr casts, no validation, bev_pool.cpp:33-36) dtype / contiguity / device are checked
and a RuntimeError is raised instead of silently misreading memory.
"""
import torch

from ... import _lib


def _check(t, name, dtype):
    _lib.require_cuda(t, name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))


def bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, b, d, h, w, _m_dev=None):
    """x f32[n,c], geom_feats i32[n,4]=(x,y,z,b), lengths/starts i32[m] -> out f32[b,d,h,w,c]."""
    _check(x, "x", torch.float32)
    _check(geom_feats, "geom_feats", torch.int32)
    _check(interval_lengths, "interval_lengths", torch.int32)
    _check(interval_starts, "interval_starts", torch.int32)
    n, c = x.shape
    m = interval_lengths.shape[0]
    b, d, h, w = int(b), int(d), int(h), int(w)
    out = torch.empty((b, d, h, w, c), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().bfhip_bev_pool_fwd(_lib.ptr(x), _lib.ptr(geom_feats), _lib.ptr(interval_starts),
                                            _lib.ptr(interval_lengths), _lib.ptr(out), n, c, m, b, d, h, w,
                                            _lib.ptr(_m_dev), _lib.stream_of(x))
    _lib.check(rc, "bev_pool_forward")
    return out


def bev_pool_backward(out_grad, geom_feats, interval_lengths, interval_starts, b, d, h, w,
                      _cover_all=False, _m_dev=None):
    """out_grad f32[b,d,h,w,c] -> x_grad f32[n,c]; rows outside every interval get 0."""
    _check(out_grad, "out_grad", torch.float32)
    _check(geom_feats, "geom_feats", torch.int32)
    _check(interval_lengths, "interval_lengths", torch.int32)
    _check(interval_starts, "interval_starts", torch.int32)
    n = geom_feats.shape[0]
    c = out_grad.shape[4]
    m = interval_lengths.shape[0]
    b, d, h, w = int(b), int(d), int(h), int(w)
    x_grad = torch.empty((n, c), dtype=out_grad.dtype, device=out_grad.device)
    with torch.cuda.device(out_grad.device):
        rc = _lib.load().bfhip_bev_pool_bwd(_lib.ptr(out_grad), _lib.ptr(geom_feats), _lib.ptr(interval_starts),
                                            _lib.ptr(interval_lengths), _lib.ptr(x_grad), n, c, m, b, d, h, w,
                                            1 if _cover_all else 0, _lib.ptr(_m_dev), _lib.stream_of(out_grad))
    _lib.check(rc, "bev_pool_backward")
    return x_grad
