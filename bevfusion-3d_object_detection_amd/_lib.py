"""ctypes binding of csrc/libbevfusion_hip.so (the C ABI declared in include/bevfusion_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, a RuntimeError is
raised (the reference raises RuntimeError from its C++ exceptions as well,
BF/ops/voxel/src/voxelization.h:75).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbevfusion_hip.so")

_c_int = ctypes.c_int
_c_vp = ctypes.c_void_p
_c_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol of include/bevfusion_hip.h
SIGNATURES = {
    "bfhip_abi_version": (_c_int, []),
    "bfhip_last_error": (ctypes.c_char_p, []),
    "bfhip_profile_enable": (None, [_c_int]),
    "bfhip_profile_read": (_c_int, [_c_int, _c_vp, _c_vp, _c_int]),
    "bfhip_bev_pool_fwd": (_c_int, [_c_vp] * 5 + [_c_int] * 7 + [_c_vp, _c_vp]),
    "bfhip_bev_pool_bwd": (_c_int, [_c_vp] * 5 + [_c_int] * 8 + [_c_vp, _c_vp]),
    "bfhip_dynamic_voxelize": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp]),
    "bfhip_hard_voxelize_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "bfhip_hard_voxelize": (_c_int, [_c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                     _c_int, _c_int, _c_vp, _c_sz, _c_vp, _c_vp]),
    "bfhip_bev_plan_workspace_bytes": (_c_sz, [ctypes.c_longlong, ctypes.c_longlong]),
    "bfhip_bev_plan": (_c_int, [_c_vp] * 7 + [_c_int] * 4 + [_c_vp] * 3 + [_c_vp] * 11 + [_c_int, _c_vp, _c_sz, _c_vp]),
    "bfhip_lift_splat_fwd": (_c_int, [_c_vp, _c_int, _c_vp, _c_int, _c_int] + [_c_vp] * 6 + [_c_int, _c_int, ctypes.c_longlong,
                                       _c_vp, _c_int, _c_vp]),
    "bfhip_lift_splat_bwd": (_c_int, [_c_vp, _c_int, _c_vp, _c_int, _c_vp, _c_int, _c_int, _c_vp] + [_c_int] * 4 +
                             [_c_vp, _c_int, _c_vp, _c_int, _c_vp]),
    "bfhip_conv_out_shape": (_c_int, [_c_vp] * 6),
    "bfhip_rulebook_subm_workspace_bytes": (_c_sz, [_c_int]),
    "bfhip_rulebook_subm": (_c_int, [_c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz,
                                     _c_vp]),
    "bfhip_rulebook_sparse_workspace_bytes": (_c_sz, [_c_int] + [_c_vp] * 5),
    "bfhip_rulebook_sparse_count": (_c_int, [_c_vp, _c_int, _c_vp, _c_int] + [_c_vp] * 5 + [_c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_rulebook_sparse_out_indices": (_c_int, [_c_int] + [_c_vp] * 5 + [_c_int, _c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_rulebook_sparse_fill": (_c_int, [_c_vp, _c_int, _c_int] + [_c_vp] * 5 + [_c_int] + [_c_vp] * 4 +
                                   [_c_vp] * 4 + [_c_vp, _c_sz] + [_c_vp, _c_sz, _c_vp]),
    "bfhip_spconv_workspace_bytes": (_c_sz, [_c_int] * 3),
    "bfhip_rulebook_sort_rows_workspace_bytes": (_c_sz, [_c_int, _c_int]),
    "bfhip_rulebook_sort_rows": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_rulebook_validate_workspace_bytes": (_c_sz, [_c_int]),
    "bfhip_rulebook_validate": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_spconv_gemm": (_c_int, [_c_vp] * 3 + [_c_int] * 7 + [_c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_spconv_gemm_bf16": (_c_int, [_c_vp] * 3 + [_c_int] * 7 + [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_sz, _c_vp]),
    "bfhip_spconv_wgrad_workspace_bytes": (_c_sz, [_c_int] * 4),
    "bfhip_spconv_wgrad": (_c_int, [_c_vp] * 3 + [_c_int] * 5 + [_c_vp, _c_vp, _c_int, _c_vp, _c_sz, _c_vp]),
    "bfhip_sparse_to_bev": (_c_int, [_c_vp, _c_vp] + [_c_int] * 6 + [_c_vp, _c_vp]),
    "bfhip_bev_to_sparse": (_c_int, [_c_vp, _c_vp] + [_c_int] * 6 + [_c_vp, _c_vp]),
    "bfhip_dynamic_scatter_workspace_bytes": (_c_sz, [_c_int]),
    "bfhip_dynamic_scatter_fwd": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_vp, _c_sz, _c_vp]),
    "bfhip_dynamic_scatter_bwd_workspace_bytes": (_c_sz, [_c_int, _c_int]),
    "bfhip_dynamic_scatter_bwd": (_c_int, [_c_vp] * 6 + [_c_int] * 4 + [_c_vp, _c_sz, _c_vp]),
    "bfhip_voxel_mean": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "bfhip_voxel_compact_mean": (_c_int, [_c_vp] * 4 + [_c_int] * 5 + [_c_vp] * 4),
    "bfhip_depth_lift_bwd_workspace_bytes": (_c_sz, []),
    "bfhip_depth_lift_bwd": (_c_int, [_c_vp, _c_vp, ctypes.c_longlong, _c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_rasterise_depth_workspace_bytes": (_c_sz, [_c_int] * 3),
    "bfhip_rasterise_depth": (_c_int, [_c_vp, _c_int, _c_int] + [_c_vp] * 4 + [_c_int] * 3 + [_c_vp, _c_vp] + [_c_int] * 3 +
                              [_c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_depth_histogram": (_c_int, [_c_vp] + [_c_int] * 6 + [_c_vp, _c_vp, _c_vp, _c_vp]),
    "bfhip_bn1d_workspace_bytes": (_c_sz, [_c_int, _c_int]),
    "bfhip_bn1d_fwd": (_c_int, [_c_vp] * 4 + [_c_int, _c_int, ctypes.c_float, ctypes.c_float, _c_int] + [_c_vp] * 5 +
                       [_c_vp, _c_sz, _c_vp]),
    "bfhip_bn1d_bwd": (_c_int, [_c_vp] * 5 + [_c_int] * 3 + [_c_vp] * 4 + [_c_vp, _c_sz, _c_vp]),
    "bfhip_sparse_to_bev_nhwc": (_c_int, [_c_vp, _c_vp] + [_c_int] * 7 + [_c_vp, _c_vp]),
    "bfhip_bev_nhwc_to_sparse": (_c_int, [_c_vp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _c_int, _c_vp, _c_int,
                                          _c_int, _c_int, _c_vp, _c_vp]),
    "bfhip_bn2d_supported": (_c_int, [ctypes.c_longlong, _c_int, _c_int]),
    "bfhip_bn2d_workspace_bytes": (_c_sz, [ctypes.c_longlong, _c_int, _c_int]),
    "bfhip_colsum": (_c_int, [_c_vp, ctypes.c_longlong, _c_int, _c_int, _c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_bn2d_fwd": (_c_int, [_c_vp] * 4 + [ctypes.c_longlong, _c_int, _c_int, ctypes.c_float, ctypes.c_float, _c_int] +
                       [_c_vp] * 5 + [_c_vp, _c_sz, _c_vp]),
    "bfhip_bn2d_bwd": (_c_int, [_c_vp] * 5 + [ctypes.c_longlong, _c_int, _c_int, _c_int] + [_c_vp] * 4 + [_c_vp, _c_sz, _c_vp]),
    "bfhip_xty_workspace_bytes": (_c_sz, [ctypes.c_longlong, _c_int, _c_int]),
    "bfhip_xty": (_c_int, [_c_vp, _c_vp, ctypes.c_longlong, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_attn_workspace_bytes": (_c_sz, [_c_int] * 4),
    "bfhip_attn_fwd": (_c_int, [_c_vp] * 3 + [_c_int] * 4 + [ctypes.c_float, ctypes.c_float, ctypes.c_ulonglong, _c_vp, _c_vp,
                                                             _c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_attn_bwd": (_c_int, [_c_vp] * 6 + [_c_int] * 4 + [ctypes.c_float, ctypes.c_float, ctypes.c_ulonglong] + [_c_vp] * 4 +
                       [_c_vp, _c_sz, _c_vp]),
    "bfhip_attn_dropout_mask": (_c_int, [_c_int] * 4 + [ctypes.c_float, ctypes.c_ulonglong, _c_vp, _c_vp, _c_vp]),
    "bfhip_upsample2x_nhwc": (_c_int, [_c_vp, _c_vp] + [_c_int] * 6 + [_c_vp]),
    "bfhip_maxpool3x3s2_fwd": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp, _c_vp]),
    "bfhip_maxpool3x3s2_bwd": (_c_int, [_c_vp, _c_vp] + [_c_int] * 4 + [_c_vp, _c_vp]),
    "bfhip_circle_nms": (_c_int, [_c_vp, _c_int, ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp]),
    "bfhip_rotate_nms_workspace_bytes": (_c_sz, [_c_int, _c_int]),
    "bfhip_rotate_nms": (_c_int, [_c_vp, _c_vp, _c_int, ctypes.c_float, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_decode_boxes": (_c_int, [_c_vp] * 5 + [_c_int] * 4 + [_c_vp, _c_vp, _c_vp]),
    "bfhip_assign_cost": (_c_int, [_c_vp, _c_int, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_int, _c_vp, _c_vp] + [_c_int] * 3 +
                          [_c_vp] * 4),
    "bfhip_hungarian": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp]),
    "bfhip_assign_targets": (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp] + [_c_int] * 5 + [_c_vp] * 7),
    "bfhip_draw_heatmap": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp] + [_c_int] * 5 + [_c_vp, ctypes.c_double, _c_int, _c_vp, _c_vp]),
    "bfhip_gaussian_focal_loss_workspace_bytes": (_c_sz, [ctypes.c_longlong]),
    "bfhip_gaussian_focal_loss": (_c_int, [_c_vp, _c_vp, ctypes.c_longlong, ctypes.c_float, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_conv2d_supported": (_c_int, [_c_int] * 10),
    "bfhip_conv2d_stat_rows": (_c_int, [_c_int] * 3),
    "bfhip_conv2d_fwd": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int] + [_c_int] * 11 + [_c_vp, _c_vp]),
    "bfhip_conv2d_dgrad_workspace_bytes": (_c_sz, [_c_int] * 4),
    "bfhip_conv2d_dgrad": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_int] + [_c_int] * 11 + [_c_vp, _c_sz, _c_vp]),
    "bfhip_conv2d_dgrad_wt": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_int, _c_vp, _c_int] + [_c_int] * 11 + [_c_vp]),
    "bfhip_conv2d_dgrad_fuses_addend": (_c_int, [_c_int] * 5),
    "bfhip_split_bf16x3": (_c_int, [_c_vp, ctypes.c_longlong, _c_int, _c_vp, _c_int, _c_vp, _c_int, _c_vp]),
    "bfhip_conv2d_wt_segment_bytes": (_c_int, []),
    "bfhip_conv2d_weight_transpose_batched": (_c_int, [_c_vp, _c_int, ctypes.c_longlong, _c_vp]),
    "bfhip_conv2d_wgrad_workspace_bytes": (_c_sz, [_c_int] * 7),
    "bfhip_conv2d_wgrad": (_c_int, [_c_vp, _c_int, _c_vp, _c_int, _c_vp] + [_c_int] * 11 + [_c_vp, _c_sz, _c_vp]),
    "bfhip_conv2d_wgrad_group_table_bytes": (_c_sz, [_c_int]),
    "bfhip_conv2d_wgrad_groupable": (_c_int, [_c_int] * 10),
    "bfhip_conv2d_wgrad_group_plan": (_c_int, [_c_vp, _c_int, _c_int, _c_vp, _c_sz, _c_vp]),
    "bfhip_conv2d_wgrad_group_launch": (_c_int, [_c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "bfhip_bn2d_fwd_partials": (_c_int, [_c_vp] * 4 + [ctypes.c_longlong, _c_int, _c_int, ctypes.c_float, ctypes.c_float, _c_int] +
                                [_c_vp] * 4 + [_c_vp, _c_int, _c_vp, _c_vp]),
    "bfhip_bn2d_fwd_partials_mask": (_c_int, [_c_vp] * 4 + [ctypes.c_longlong, _c_int, _c_int, ctypes.c_float, ctypes.c_float, _c_int] +
                                     [_c_vp] * 4 + [_c_vp, _c_int, _c_vp, _c_vp, _c_vp]),
    "bfhip_bn2d_bwd_mask": (_c_int, [_c_vp] * 5 + [ctypes.c_longlong, _c_int, _c_int] + [_c_vp] * 4 + [_c_vp, _c_sz, _c_vp]),
    "bfhip_adamw_segment_bytes": (_c_int, []),
    "bfhip_adamw_chunk_elems": (_c_int, []),
    "bfhip_adamw_step": (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp] + [ctypes.c_float] * 6 + [_c_vp]),
    "bfhip_query_losses": (_c_int, [_c_vp] * 7 + [_c_int] * 6 + [ctypes.c_float, ctypes.c_float] + [_c_vp] * 4),
}

_lib = None


def load():
    """Load the HIP library once; raise loudly if it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libbevfusion_hip.so not found at %s: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


_torch_ext = False


def torch_ext():
    """csrc/bfhip_torch_ext.so (C++ autograd front-ends, csrc/torch_binding.cpp) or None when it has not been built or
    BFHIP_TORCH_EXT=0.  Same kernels either way: it only replaces the ctypes + Python autograd plumbing."""
    global _torch_ext
    if _torch_ext is False:
        _torch_ext = None
        path = os.path.join(_HERE, "csrc", "bfhip_torch_ext.so")
        if os.environ.get("BFHIP_TORCH_EXT", "1") == "1" and os.path.exists(path):
            import importlib.util
            load()
            try:
                spec = importlib.util.spec_from_file_location("bfhip_torch_ext", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                if mod.abi_version() == load().bfhip_abi_version():
                    _torch_ext = mod
            except (ImportError, OSError) as e:  # binding built against another torch: keep the ctypes path
                import warnings
                warnings.warn("bfhip_torch_ext not usable (%s); using the ctypes binding" % e)
    return _torch_ext


def check(rc, what):
    if rc != 0:
        msg = load().bfhip_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg))


_fn_cache = {}


def call(name, *args):
    """Invoke an `int`-returning entry point and raise RuntimeError(bfhip_last_error()) on failure."""
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        check(rc, name)


def call_size(name, *args):
    """Invoke a *_workspace_bytes() query."""
    return int(getattr(load(), name)(*args))


def ptr(t):
    """Device pointer of a tensor (None -> NULL); a plain int is what ctypes converts fastest for a void* parameter."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(t):
    """hipStream_t of torch's current stream on t's device (raw handle; torch.cuda.current_stream() costs ~5 us a call)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def host_i32(values):
    return (ctypes.c_int32 * len(values))(*[int(v) for v in values])


def host_f32(values):
    arr = (ctypes.c_float * len(values))(*[float(v) for v in values])
    return arr


def require_cuda(t, name):
    # mirrors CHECK_INPUT (BF/ops/voxel/src/voxelization_cuda.cu:8-14): device + contiguous
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA(HIP) tensor; this build has no CPU path" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


OPS = dict(bev_pool_fwd=0, bev_pool_bwd=1, hard_voxelize=2, dynamic_voxelize=3, lift_splat_fwd=4, lift_splat_bwd=5,
           spconv_fwd=6, spconv_bwd=7, rulebook=8, bev_aux=9, scatter_fwd=10, scatter_bwd=11, spconv_wgrad=12, raster=13,
           spconv_wgrad_main=14, conv2d_fwd=15, conv2d_dgrad=16, conv2d_wgrad=17, bn2d_fwd=18, bn2d_bwd=19,
           conv2d_pw_fwd=20, conv2d_pw_dgrad=21)


def profile_enable(on=True):
    """0 / False: off; 1 / True: the sparse / lift-splat / voxel ops; 2: also the dense ops (conv2d_*, bn2d_*)."""
    load().bfhip_profile_enable(int(on))


def profile_read(op, reset=True):
    """(sum_ms, count) of the HIP-event pairs recorded around op's dominant kernel."""
    s = ctypes.c_double(0.0)
    c = ctypes.c_longlong(0)
    check(load().bfhip_profile_read(OPS[op] if isinstance(op, str) else op, ctypes.byref(s), ctypes.byref(c),
                                    1 if reset else 0), "profile_read")
    return s.value, c.value
