"""numpy front-end of the C oracle (oracle/bevfusion_oracle.c, oracle/spconv_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Every function takes/returns numpy
arrays; the reference lines each one restates are cited in the C sources.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbevfusion_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("bevfusion_oracle.c", "spconv_oracle.c")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "libbevfusion_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_hard_voxelize.restype = ctypes.c_int
        _lib.oracle_intervals_from_ranks.restype = ctypes.c_int
        _lib.oracle_dynamic_scatter_fwd.restype = ctypes.c_int
        _lib.oracle_rulebook_subm.restype = ctypes.c_int64
        _lib.oracle_rulebook_sparse.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


REDUCE = {"sum": 0, "mean": 1, "max": 2}


# ------------------------------------------------------------------ voxelization
def grid_size(voxel_size, coors_range):
    vs, cr = _f32(voxel_size), _f32(coors_range)
    g = np.zeros(3, np.int32)
    lib().oracle_grid_size(_p(vs), _p(cr), 3, _p(g))
    return g


def dynamic_voxelize(points, voxel_size, coors_range):
    points = _f32(points)
    vs, cr = _f32(voxel_size), _f32(coors_range)
    coors = np.zeros((points.shape[0], 3), np.int32)
    lib().oracle_dynamic_voxelize(_p(points), points.shape[0], points.shape[1], _p(vs), _p(cr), 3, _p(coors))
    return coors


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """returns (voxels[M,P,F], coors[M,3] xyz, num_points[M]) like Voxelization.forward."""
    points = _f32(points)
    vs, cr = _f32(voxel_size), _f32(coors_range)
    n, f = points.shape
    voxels = np.zeros((max_voxels, max_points, f), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    m = lib().oracle_hard_voxelize(_p(points), n, f, _p(vs), _p(cr), int(max_points), int(max_voxels),
                                   _p(voxels), _p(coors), _p(num))
    if m < 0:
        raise MemoryError("oracle_hard_voxelize")
    return voxels[:m], coors[:m], num[:m]


def voxel_mean(voxels, num_points):
    voxels, num_points = _f32(voxels), _i32(num_points)
    m, p, f = voxels.shape
    out = np.zeros((m, f), np.float32)
    lib().oracle_voxel_mean(_p(voxels), _p(num_points), m, p, f, _p(out))
    return out


# ------------------------------------------------------------------ bev_pool
def intervals_from_ranks(ranks):
    ranks = _i64(ranks)
    n = ranks.shape[0]
    starts = np.zeros(max(n, 1), np.int32)
    lengths = np.zeros(max(n, 1), np.int32)
    m = lib().oracle_intervals_from_ranks(_p(ranks), n, _p(starts), _p(lengths))
    return starts[:m].copy(), lengths[:m].copy()


def bev_pool_fwd(x, geom, starts, lengths, b, d, h, w):
    x, geom, starts, lengths = _f32(x), _i32(geom), _i32(starts), _i32(lengths)
    n, c = x.shape
    out = np.empty((b, d, h, w, c), np.float32)
    lib().oracle_bev_pool_fwd(_p(x), _p(geom), _p(starts), _p(lengths), n, c, starts.shape[0], b, d, h, w, _p(out))
    return out


def bev_pool_bwd(out_grad, geom, starts, lengths, n):
    out_grad, geom, starts, lengths = _f32(out_grad), _i32(geom), _i32(starts), _i32(lengths)
    b, d, h, w, c = out_grad.shape
    xg = np.empty((n, c), np.float32)
    lib().oracle_bev_pool_bwd(_p(out_grad), _p(geom), _p(starts), _p(lengths), n, c, starts.shape[0], b, d, h, w, _p(xg))
    return xg


def quickcumsum_f64(x, ranks):
    x, ranks = _f32(x), _i64(ranks)
    n, c = x.shape
    m = int((np.diff(ranks) != 0).sum() + 1) if n else 0
    out = np.zeros((m, c), np.float64)
    lib().oracle_quickcumsum_f64(_p(x), _p(ranks), n, c, _p(out))
    return out


def lift_splat_fwd(depth, feat, src, geom, starts, lengths, b, d, h, w):
    """depth [BN,D,fH,fW], feat [BN,C,fH,fW], src i32[nk] flat frustum row of each sorted kept point."""
    depth, feat = _f32(depth), _f32(feat)
    src, geom, starts, lengths = _i32(src), _i32(geom), _i32(starts), _i32(lengths)
    BN, D = depth.shape[:2]
    HW = depth.shape[2] * depth.shape[3]
    C = feat.shape[1]
    out = np.empty((b, d, h, w, C), np.float32)
    lib().oracle_lift_splat_fwd(_p(depth), _p(feat), _p(src), _p(geom), _p(starts), _p(lengths),
                                src.shape[0], starts.shape[0], BN, D, HW, C, b, d, h, w, _p(out))
    return out


def lift_splat_bwd(out_grad, depth, feat, src, geom, starts, lengths):
    out_grad, depth, feat = _f32(out_grad), _f32(depth), _f32(feat)
    src, geom, starts, lengths = _i32(src), _i32(geom), _i32(starts), _i32(lengths)
    b, d, h, w, C = out_grad.shape
    BN, D = depth.shape[:2]
    HW = depth.shape[2] * depth.shape[3]
    dd = np.empty_like(depth)
    df = np.empty_like(feat)
    lib().oracle_lift_splat_bwd(_p(out_grad), _p(depth), _p(feat), _p(src), _p(geom), _p(starts), _p(lengths),
                                src.shape[0], starts.shape[0], BN, D, HW, C, b, d, h, w, _p(dd), _p(df))
    return dd, df


def frustum_geometry(frustum, post_trans, post_rots_inv, combine, c2l_trans, extra_rots, extra_trans):
    """frustum [D,fH,fW,3]; per camera [B,N,...]; per sample [B,...] -> [B,N,D,fH,fW,3]."""
    frustum = _f32(frustum)
    B, N = post_trans.shape[:2]
    DHW = frustum.size // 3
    args = [_f32(a) for a in (post_trans, post_rots_inv, combine, c2l_trans, extra_rots, extra_trans)]
    out = np.empty((B, N) + frustum.shape, np.float32)
    lib().oracle_frustum_geometry(_p(frustum), B, N, DHW, *[_p(a) for a in args], _p(out))
    return out


def bev_cells(geom, B, origin, dx, nx):
    """geom f32[..., 3] with leading dim B-major.  returns cell i32[N',4] (x,y,z,b), kept bool[N'], rank i64[N']."""
    geom = _f32(geom).reshape(-1, 3)
    n = geom.shape[0]
    origin, dx, nx = _f32(origin), _f32(dx), _i32(nx)
    cell = np.empty((n, 4), np.int32)
    kept = np.empty((n,), np.uint8)
    rank = np.empty((n,), np.int64)
    lib().oracle_bev_cells(_p(geom), ctypes.c_int64(n), int(B), _p(origin), _p(dx), _p(nx), _p(cell), _p(kept), _p(rank))
    return cell, kept.astype(bool), rank


def bev_pool_aux(geom, B, origin, dx, nx):
    """BF/depth_lss.py:118-176 end to end with a STABLE sort (the reference's argsort is unstable,
    so within-interval order is unspecified there).  returns geom_feats i32[nk,4], kept, ranks, indices
    where `indices` index the kept subset exactly like the reference's."""
    cell, kept, rank = bev_cells(geom, B, origin, dx, nx)
    gk, rk = cell[kept], rank[kept]
    indices = np.argsort(rk, kind="stable")
    return gk[indices], kept, rk[indices], indices


# ------------------------------------------------------------------ dynamic scatter
def dynamic_scatter_fwd(feats, coors, reduce_type):
    feats, coors = _f32(feats), _i32(coors)
    n, c = feats.shape
    vf = np.zeros((max(n, 1), c), np.float32)
    vc = np.zeros((max(n, 1), 3), np.int32)
    p2v = np.zeros((max(n, 1),), np.int32)
    cnt = np.zeros((max(n, 1),), np.int32)
    m = lib().oracle_dynamic_scatter_fwd(_p(feats), _p(coors), n, c, REDUCE[reduce_type], _p(vf), _p(vc), _p(p2v), _p(cnt))
    return vf[:m].copy(), vc[:m].copy(), p2v[:n].copy(), cnt[:m].copy()


def dynamic_scatter_bwd(grad_voxel_feats, feats, voxel_feats, point2voxel, voxel_count, reduce_type):
    g, feats, vf = _f32(grad_voxel_feats), _f32(feats), _f32(voxel_feats)
    p2v, cnt = _i32(point2voxel), _i32(voxel_count)
    n, c = feats.shape
    out = np.empty((n, c), np.float32)
    lib().oracle_dynamic_scatter_bwd(_p(g), _p(feats), _p(vf), _p(p2v), _p(cnt), n, vf.shape[0], c, REDUCE[reduce_type], _p(out))
    return out


# ------------------------------------------------------------------ sparse conv
def _i3(v):
    v = [v] * 3 if np.isscalar(v) else list(v)
    return np.asarray(v, np.int32)


def conv_out_shape(in_shape, ksize, stride, padding, dilation=1):
    out = np.zeros(3, np.int32)
    lib().oracle_conv_out_shape(_p(_i3(in_shape)), _p(_i3(ksize)), _p(_i3(stride)), _p(_i3(padding)), _p(_i3(dilation)), _p(out))
    return out


def rulebook_subm(indices, shape, ksize, dilation=1):
    indices = _i32(indices)
    n = indices.shape[0]
    ks = _i3(ksize)
    kv = int(np.prod(ks))
    pair = np.empty((kv, n), np.int32)
    lib().oracle_rulebook_subm(_p(indices), n, _p(_i3(shape)), _p(ks), _p(_i3(dilation)), _p(pair))
    return pair


def rulebook_sparse(indices, shape, ksize, stride, padding, dilation=1):
    """returns out_indices[N_out,4] (canonical ascending), pair_fwd[KV,N_out], pair_bwd[KV,N_in], out_shape."""
    indices = _i32(indices)
    n = indices.shape[0]
    ks = _i3(ksize)
    kv = int(np.prod(ks))
    max_out = max(n * kv, 1)
    out_idx = np.empty((max_out, 4), np.int32)
    pf = np.empty((kv, max_out), np.int32)
    pb = np.empty((kv, max(n, 1)), np.int32)
    npairs = ctypes.c_int64(0)
    m = lib().oracle_rulebook_sparse(_p(indices), n, _p(_i3(shape)), _p(ks), _p(_i3(stride)), _p(_i3(padding)),
                                     _p(_i3(dilation)), max_out, _p(out_idx), _p(pf), _p(pb), ctypes.byref(npairs))
    if m < 0:
        raise MemoryError("oracle_rulebook_sparse")
    return out_idx[:m].copy(), pf[:, :m].copy(), pb[:, :n].copy(), conv_out_shape(shape, ksize, stride, padding, dilation)


def spconv_fwd(feat_in, weight, pair_fwd):
    """weight (Cout, k0,k1,k2, Cin); pair_fwd [KV, N_out]."""
    feat_in, weight, pair_fwd = _f32(feat_in), _f32(weight), _i32(pair_fwd)
    cout, cin = weight.shape[0], weight.shape[-1]
    kv, n_out = pair_fwd.shape
    out = np.empty((n_out, cout), np.float32)
    lib().oracle_spconv_fwd(_p(feat_in), _p(weight), _p(pair_fwd), n_out, n_out, kv, cin, cout, _p(out))
    return out


def spconv_bwd(feat_in, weight, d_out, pair_fwd):
    feat_in, weight, d_out, pair_fwd = _f32(feat_in), _f32(weight), _f32(d_out), _i32(pair_fwd)
    cout, cin = weight.shape[0], weight.shape[-1]
    kv, n_out = pair_fwd.shape
    n_in = feat_in.shape[0]
    d_in = np.empty((n_in, cin), np.float32)
    d_w = np.empty(weight.shape, np.float32)
    lib().oracle_spconv_bwd(_p(feat_in), _p(weight), _p(d_out), _p(pair_fwd), n_out, n_in, n_out, kv, cin, cout, _p(d_in), _p(d_w))
    return d_in, d_w


def sparse_to_bev(feats, indices, B, X, Y, Z):
    feats, indices = _f32(feats), _i32(indices)
    n, c = feats.shape
    out = np.empty((B, c * Z, X, Y), np.float32)
    lib().oracle_sparse_to_bev(_p(feats), _p(indices), n, c, B, X, Y, Z, _p(out))
    return out


# ------------------------------------------------------------------ sparse depth rasteriser + GT histogram
def rasterise_depth(points, inv_rot, aug_trans, lidar2image, img_aug, iH, iW):
    """points f32[n,F]; inv_rot f32[3,3]; aug_trans f32[3]; lidar2image, img_aug f32[ncam,4,4] -> depth f32[ncam,iH,iW]."""
    points = _f32(points)
    l2i, ia = _f32(lidar2image), _f32(img_aug)
    ncam = l2i.shape[0]
    out = np.empty((ncam, iH, iW), np.float32)
    lib().oracle_rasterise_depth(_p(points), points.shape[0], points.shape[1], _p(_f32(inv_rot)), _p(_f32(aug_trans)),
                                 _p(l2i), _p(ia), ncam, int(iH), int(iW), _p(out))
    return out


def depth_histogram(depth, fH, fW, D, dbound):
    """depth f32[BN,h,w] -> (counts, distr) f32[BN,fH,fW,D]."""
    depth = _f32(depth)
    BN, h, w = depth.shape
    counts = np.empty((BN, fH, fW, D), np.float32)
    distr = np.empty((BN, fH, fW, D), np.float32)
    lib().oracle_depth_histogram(_p(depth), BN, h, w, int(fH), int(fW), int(D), ctypes.c_float(dbound[0]),
                                 ctypes.c_float(dbound[1]), ctypes.c_float(dbound[2]), _p(counts), _p(distr))
    return counts, distr
