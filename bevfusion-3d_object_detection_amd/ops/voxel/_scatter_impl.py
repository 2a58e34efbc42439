"""dynamic_point_to_voxel_forward / _backward on top of csrc/scatter.hip (C ABI)."""
import torch

from ... import _lib

_REDUCE = {"sum": 0, "mean": 1, "max": 2}
_WS = {}


def _workspace(device, nbytes, tag):
    key = (device, torch.cuda.current_stream(device).cuda_stream, tag)  # per stream: see spconv._workspace
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _reduce_id(reduce_type):
    if reduce_type not in _REDUCE:
        # same message as the reference's convert_reduce_type (voxelization.h:98-107)
        raise RuntimeError("do not support reduce type " + str(reduce_type))
    return _REDUCE[reduce_type]


def forward(feats, coors, reduce_type):
    """-> [voxel_feats f32[M,C], voxel_coors i32[M,3], point2voxel_map i32[N], voxel_points_count i32[M]]"""
    rt = _reduce_id(reduce_type)
    _lib.require_cuda(feats, "feats")
    _lib.require_cuda(coors, "coors")
    if feats.dtype != torch.float32:
        raise RuntimeError("feats must be float32")
    coors = coors.int() if coors.dtype != torch.int32 else coors
    N, C = feats.shape
    if N == 0:  # scatter_points_cuda.cu:193-197
        return [feats.clone().detach(), coors.clone().detach(), coors.new_empty((0,), dtype=torch.int32),
                coors.new_empty((0,), dtype=torch.int32)]
    dev = feats.device
    voxel_feats = torch.empty((N, C), dtype=torch.float32, device=dev)
    voxel_coors = torch.empty((N, 3), dtype=torch.int32, device=dev)
    p2v = torch.empty((N,), dtype=torch.int32, device=dev)
    count = torch.empty((N,), dtype=torch.int32, device=dev)
    counts = torch.empty(3, dtype=torch.int32, device=dev)
    lib = _lib.load()
    ws = _workspace(dev, lib.bfhip_dynamic_scatter_workspace_bytes(N), "fwd")
    with torch.cuda.device(dev):
        rc = lib.bfhip_dynamic_scatter_fwd(_lib.ptr(feats), _lib.ptr(coors), N, C, rt, _lib.ptr(voxel_feats),
                                           _lib.ptr(voxel_coors), _lib.ptr(p2v), _lib.ptr(count), _lib.ptr(counts),
                                           _lib.ptr(ws), ws.numel(), _lib.stream_of(feats))
    _lib.check(rc, "dynamic_point_to_voxel_forward")
    m, _, overflow = counts.tolist()  # output shapes depend on M: one host read (unique_dim syncs too)
    if overflow:
        raise RuntimeError("dynamic_point_to_voxel_forward: voxel coordinates must be < 2^21")
    return [voxel_feats[:m], voxel_coors[:m], p2v, count[:m]]


def backward(grad_feats, grad_voxel_feats, feats, voxel_feats, point2voxel_map, voxel_points_count, reduce_type):
    rt = _reduce_id(reduce_type)
    for t, name in ((grad_feats, "grad_feats"), (grad_voxel_feats, "grad_reduced_feats"), (feats, "feats"),
                    (voxel_feats, "reduced_feats"), (point2voxel_map, "coors_idx"), (voxel_points_count, "reduce_count")):
        _lib.require_cuda(t, name)
    N, C = feats.shape
    M = voxel_feats.shape[0]
    lib = _lib.load()
    ws = _workspace(feats.device, lib.bfhip_dynamic_scatter_bwd_workspace_bytes(M, C), "bwd")
    with torch.cuda.device(feats.device):
        rc = lib.bfhip_dynamic_scatter_bwd(_lib.ptr(grad_feats), _lib.ptr(grad_voxel_feats), _lib.ptr(feats),
                                           _lib.ptr(voxel_feats), _lib.ptr(point2voxel_map),
                                           _lib.ptr(voxel_points_count), N, M, C, rt, _lib.ptr(ws), ws.numel(),
                                           _lib.stream_of(feats))
    _lib.check(rc, "dynamic_point_to_voxel_backward")
