"""`voxel_layer` -- same four functions as the reference's pybind module
(BF/ops/voxel/src/voxelization.cpp:6-11, dispatch in voxelization.h:58-140), backed by
csrc/voxelize.hip and csrc/scatter.hip through the C ABI.

The reference dispatches CPU tensors to its (buggy) CPU path; this build is
device-only: CPU tensors raise RuntimeError like the reference built without WITH_CUDA does for
GPU tensors (voxelization.h:75).
"""
import torch

from ... import _lib

_WS = {}  # grow-only workspace cache; one per (device, stream): see spconv._workspace


def _workspace(device, nbytes):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _check_points(points):
    _lib.require_cuda(points, "points")
    if points.dtype != torch.float32:
        raise RuntimeError("points must be float32, got %s" % points.dtype)
    if points.dim() != 2 or points.shape[1] < 3:
        raise RuntimeError("points must be [N, >=3]")


def dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3):
    """coors i32[N,3] <- voxel index of every point, (-1,-1,-1) outside the range (in place)."""
    if NDim != 3:
        raise RuntimeError("only NDim=3 is supported")
    _check_points(points)
    _lib.require_cuda(coors, "coors")
    if coors.dtype != torch.int32 or coors.shape != (points.shape[0], 3):
        raise RuntimeError("coors must be int32 [N,3]")
    with torch.cuda.device(points.device):
        rc = _lib.load().bfhip_dynamic_voxelize(_lib.ptr(points), _lib.ptr(coors), points.shape[0], points.shape[1],
                                                _lib.host_f32(voxel_size), _lib.host_f32(coors_range),
                                                _lib.stream_of(points))
    _lib.check(rc, "dynamic_voxelize")


def hard_voxelize_async(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range, max_points,
                        max_voxels, voxel_num_dev=None):
    """Sync-free form: returns the device int32[1] tensor holding the voxel count."""
    _check_points(points)
    for t, name in ((voxels, "voxels"), (coors, "coors"), (num_points_per_voxel, "num_points_per_voxel")):
        _lib.require_cuda(t, name)
    n, f = points.shape
    if voxels.dtype != torch.float32 or coors.dtype != torch.int32 or num_points_per_voxel.dtype != torch.int32:
        raise RuntimeError("voxels must be float32, coors / num_points_per_voxel int32")
    if voxels.shape[0] < max_voxels or voxels.shape[1] != max_points or voxels.shape[2] != f:
        raise RuntimeError("voxels must be [>=max_voxels, max_points, F]")
    if coors.shape[0] < max_voxels or num_points_per_voxel.shape[0] < max_voxels:
        raise RuntimeError("coors / num_points_per_voxel must hold max_voxels rows")
    lib = _lib.load()
    if voxel_num_dev is None:
        voxel_num_dev = torch.empty(1, dtype=torch.int32, device=points.device)
    nbytes = lib.bfhip_hard_voxelize_workspace_bytes(n, max_points, max_voxels)
    ws = _workspace(points.device, nbytes)
    with torch.cuda.device(points.device):
        rc = lib.bfhip_hard_voxelize(_lib.ptr(points), n, f, _lib.ptr(voxels), _lib.ptr(coors),
                                     _lib.ptr(num_points_per_voxel), _lib.host_f32(voxel_size),
                                     _lib.host_f32(coors_range), int(max_points), int(max_voxels), _lib.ptr(ws),
                                     ws.numel(), _lib.ptr(voxel_num_dev), _lib.stream_of(points))
    _lib.check(rc, "hard_voxelize")
    return voxel_num_dev


def hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range, max_points, max_voxels,
                  NDim=3, deterministic=True):
    """Fills voxels/coors/num_points_per_voxel in place and returns voxel_num as a Python int
    (one D2H read, as the reference does at voxelization_cuda.cu:369-370).  `deterministic` is
    accepted for signature parity; the HIP path is always deterministic."""
    if NDim != 3:
        raise RuntimeError("only NDim=3 is supported")
    return int(hard_voxelize_async(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range,
                                   max_points, max_voxels).item())


def dynamic_point_to_voxel_forward(feats, coors, reduce_type):
    from . import _scatter_impl
    return _scatter_impl.forward(feats, coors, reduce_type)


def dynamic_point_to_voxel_backward(grad_feats, grad_voxel_feats, feats, voxel_feats, point2voxel_map,
                                    voxel_points_count, reduce_type):
    from . import _scatter_impl
    return _scatter_impl.backward(grad_feats, grad_voxel_feats, feats, voxel_feats, point2voxel_map,
                                  voxel_points_count, reduce_type)
