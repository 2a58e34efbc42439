"""Host mirror of projects/BEVFusion/bevfusion/depth_lss.py (view transform of the camera branch).

Same class and method names as the reference (`BaseViewTransform`, `LSSTransform`,
`BaseDepthTransform`, `DepthLSSTransform`; `create_frustum`, `get_geometry`, `bev_pool_aux`,
`bev_pool`, `bev_pool_precomputed`, `get_cam_feats`, `forward`).  What differs is where the work
happens:

  * geometry -> cell -> rank -> stable sort -> intervals run in ONE sync-free device plan
    (`BevPlan`, csrc/lift_splat.hip: bfhip_bev_plan) instead of torch glue with an int64 argsort,
    boolean-mask syncs and a materialised [B,N,D,H,W,3] tensor (reference :68-176);
  * the depth (x) feature outer product and the two gathers are fused into the pooling kernel
    (`lift_splat`, bfhip_lift_splat_fwd/bwd); the [N',C] tensor (638 MB/sample fp32) never exists.

`bev_pool(x, geom)` / `bev_pool_precomputed(...)` keep the reference's op-boundary semantics for
callers that hand over a materialised x.
"""
import os
from typing import Tuple

import torch
from torch import nn

from . import _lib
from .bn2d import bn_act
from .conv2d import Conv2d
from .ops import bev_pool
from .registry import MODELS


# Camera-major processing order of the BEV intervals in lift_splat_fwd (csrc/lift_splat.hip: plan_group_key_kernel).  It cuts
# the kernel's memory-side traffic 11x (rocprofv3 FETCH_SIZE 1.1 -> 0.1 GB per batch-4 launch: the gathered feature rows then
# hit the per-XCD L2) but not its time (0.350 vs 0.374 ms alone, 0.379 vs 0.386 ms inside the step): the kernel is bound by
# the per-CU rate of 320-byte row gathers through the vector L1, and the extra key sort costs 0.12 ms per plan.  Off by
# default for that reason; on, it frees ~1 GB of Infinity-Cache / HBM traffic per step for whatever runs beside it.
BF16_BEV_OUT = os.environ.get("BFHIP_LIFT_SPLAT_BF16_OUT", "1") == "1"  # A/B switch for lift_splat_bev's bf16 hand-over
CAMERA_MAJOR = os.environ.get("BFHIP_LIFT_SPLAT_ORDER", "0") == "1"
# bf16 feature rows gathered as the bf16 depthnet stored them (bit-identical to widening them first: BF/depth_lss.py:467-468)
BF16_FEAT = os.environ.get("BFHIP_LIFT_SPLAT_BF16_FEAT", "1") == "1"


def _inverse(m):
    """torch.inverse without its device -> host read of the LU status (same factorisation; the reference's calibration
    matrices are never singular): keeps the camera branch free of synchronising calls."""
    return torch.linalg.inv_ex(m, check_errors=False).inverse


def gen_dx_bx(xbound, ybound, zbound):
    """(reference :14-18) dx = step, bx = first cell centre, nx = int((hi-lo)/step)."""
    rows = [xbound, ybound, zbound]
    dx = torch.Tensor([row[2] for row in rows])
    bx = torch.Tensor([row[0] + row[2] / 2.0 for row in rows])
    nx = torch.LongTensor([int((row[1] - row[0]) / row[2]) for row in rows])
    return dx, bx, nx


class BevPlan:
    """Device-resident result of bfhip_bev_plan for one batch of calibrations.

    sorted_pd u32[N'], starts/lengths/cell_of_interval i32[mmax], counts i32[2]={n_kept, m},
    cell_of_point i32[N'] and, when `with_reference_outputs`, the arrays the reference's
    bev_pool_aux returns (geom_feats i32[N',4], ranks i64[N'], kept bool[N']).
    """

    def __init__(self, B, N, D, HW, nx, device, with_reference_outputs=False, with_geometry=False):
        self.B, self.N, self.D, self.HW = B, N, D, HW
        self.nx = [int(v) for v in nx]
        self.nprime = B * N * D * HW
        self.out_cells = B * self.nx[2] * self.nx[0] * self.nx[1]
        self.mmax = min(self.nprime, self.out_cells)
        i32 = dict(dtype=torch.int32, device=device)
        self.sorted_pd = torch.empty(self.nprime, **i32)
        self.starts = torch.empty(self.mmax, **i32)
        self.lengths = torch.empty(self.mmax, **i32)
        self.cell_of_interval = torch.empty(self.mmax, **i32)
        # camera-major processing order of the intervals (lift_splat_fwd locality); BFHIP_LIFT_SPLAT_ORDER=0: rank order
        self.interval_order = torch.empty(self.mmax, **i32) if CAMERA_MAJOR else None  # read at construction (tests patch it)
        self.counts = torch.zeros(2, **i32)
        self.cell_of_point = torch.empty(self.nprime, **i32)
        self.geom_sorted = torch.empty((self.nprime, 4), **i32) if with_reference_outputs else None
        self.ranks_sorted = torch.empty(self.nprime, dtype=torch.int64, device=device) if with_reference_outputs else None
        self.kept = torch.empty(self.nprime, dtype=torch.uint8, device=device) if with_reference_outputs else None
        self.geom_xyz = torch.empty((self.nprime, 3), dtype=torch.float32, device=device) if with_geometry else None
        nbytes = _lib.load().bfhip_bev_plan_workspace_bytes(self.nprime, self.out_cells)
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=device)

    def build(self, frustum, post_trans, post_rots_inv, combine, c2l_trans, extra_rots, extra_trans, origin, dx):
        """All tensor arguments on the device, fp32, contiguous; origin/dx python floats (host)."""
        f32 = lambda t: t.contiguous().float()
        args = [f32(frustum), f32(post_trans), f32(post_rots_inv), f32(combine), f32(c2l_trans), f32(extra_rots),
                f32(extra_trans)]
        import ctypes
        nx_host = (ctypes.c_int32 * 3)(*self.nx)
        with torch.cuda.device(self.sorted_pd.device):
            rc = _lib.load().bfhip_bev_plan(
                *[_lib.ptr(a) for a in args], self.B, self.N, self.D, self.HW, _lib.host_f32(origin), _lib.host_f32(dx),
                nx_host, _lib.ptr(self.sorted_pd), _lib.ptr(self.starts), _lib.ptr(self.lengths),
                _lib.ptr(self.cell_of_interval), _lib.ptr(self.interval_order), _lib.ptr(self.counts), _lib.ptr(self.cell_of_point),
                _lib.ptr(self.geom_sorted), _lib.ptr(self.ranks_sorted), _lib.ptr(self.kept), _lib.ptr(self.geom_xyz),
                self.mmax, _lib.ptr(self.workspace), self.workspace.numel(), _lib.stream_of(self.sorted_pd))
        _lib.check(rc, "bev_plan")
        self._keepalive = args
        return self


class _LiftSplat(torch.autograd.Function):
    """out[b,z,x,y,:] = sum over the cell's frustum points of depth[p,d] * feat[p,:]."""

    @staticmethod
    def forward(ctx, depth, feat, plan, out_dtype=torch.float32):
        # depth [P, D] pixel-major, feat [P, C] pixel-major (any pitch, last dim contiguous); out_dtype bf16: the fp32 sums are
        # rounded once on store (what a consumer's cast would do) and the backward takes the bf16 gradient as it is
        assert depth.dim() == 2 and feat.dim() == 2 and depth.stride(1) == 1 and feat.stride(1) == 1
        # feat bf16: the depthnet's bf16 output gathered as stored (same values as the reference's x.float(), half the bytes)
        assert depth.dtype == torch.float32 and feat.dtype in (torch.float32, torch.bfloat16)
        f16 = 1 if feat.dtype == torch.bfloat16 else 0
        P, C = feat.shape
        assert P == plan.B * plan.N * plan.HW and depth.shape == (P, plan.D)
        assert out_dtype in (torch.float32, torch.bfloat16)
        out = torch.empty((plan.B, plan.nx[2], plan.nx[0], plan.nx[1], C), dtype=out_dtype, device=feat.device)
        with torch.cuda.device(feat.device):
            rc = _lib.load().bfhip_lift_splat_fwd(
                _lib.ptr(depth), depth.stride(0), _lib.ptr(feat), f16, feat.stride(0), _lib.ptr(plan.sorted_pd),
                _lib.ptr(plan.starts), _lib.ptr(plan.lengths), _lib.ptr(plan.cell_of_interval), _lib.ptr(plan.interval_order),
                _lib.ptr(plan.counts), plan.mmax, C, plan.out_cells, _lib.ptr(out), 1 if out_dtype == torch.bfloat16 else 0,
                _lib.stream_of(feat))
        _lib.check(rc, "lift_splat_fwd")
        ctx.save_for_backward(depth, feat)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, out_grad):
        depth, feat = ctx.saved_tensors
        plan = ctx.plan
        out_grad = out_grad.contiguous()
        if out_grad.dtype not in (torch.float32, torch.bfloat16):
            out_grad = out_grad.float()
        P, C = feat.shape
        d_depth = torch.empty((P, plan.D), dtype=torch.float32, device=feat.device)
        d_feat = torch.empty((P, C), dtype=feat.dtype, device=feat.device)  # bf16 features: bf16 gradient, rounded once
        with torch.cuda.device(feat.device):
            rc = _lib.load().bfhip_lift_splat_bwd(
                _lib.ptr(out_grad), 1 if out_grad.dtype == torch.bfloat16 else 0, _lib.ptr(depth), depth.stride(0), _lib.ptr(feat),
                1 if feat.dtype == torch.bfloat16 else 0, feat.stride(0),
                _lib.ptr(plan.cell_of_point), plan.B * plan.N, plan.D, plan.HW, C, _lib.ptr(d_depth), plan.D,
                _lib.ptr(d_feat), C, _lib.stream_of(feat))
        _lib.check(rc, "lift_splat_bwd")
        return d_depth, d_feat, None, None


class _SplitDepthFeat(torch.autograd.Function):
    """x bf16[P, >= D + C] (the depthnet's pixel-major output) -> (depth logits f32[P, D], features bf16[P, C], both contiguous):
    two strided copies forward, two strided copies into ONE bf16 gradient backward -- instead of widening all D + C channels,
    slicing, re-packing the feature slice and, in the backward, two zero-padded slice gradients plus their sum."""

    @staticmethod
    def forward(ctx, x, D, C):
        ctx.shape, ctx.D, ctx.C = x.shape, D, C
        return x[:, :D].float(), x[:, D:D + C].contiguous()

    @staticmethod
    def backward(ctx, g_logits, g_feat):
        D, C = ctx.D, ctx.C
        dx = torch.empty(ctx.shape, dtype=torch.bfloat16, device=g_logits.device)
        dx[:, :D].copy_(g_logits)          # fp32 -> bf16, rounded once (what the backward of x.float() does)
        dx[:, D:D + C].copy_(g_feat)
        if ctx.shape[1] > D + C:
            dx[:, D + C:].zero_()
        return dx, None, None


def lift_splat(depth, feat, plan, out_dtype=torch.float32):
    """depth f32[P,D], feat f32 | bf16 [P,C] (pixel-major) -> BEV f32 (or bf16) [B, nz, nx, ny, C]."""
    return _LiftSplat.apply(depth, feat, plan, out_dtype)


class BaseViewTransform(nn.Module):

    def __init__(self, in_channels: int, out_channels: int, image_size: Tuple[int, int],
                 feature_size: Tuple[int, int], xbound, ybound, zbound, dbound) -> None:
        super().__init__()
        self.in_channels = in_channels
        self.image_size = image_size
        self.feature_size = feature_size
        self.xbound, self.ybound, self.zbound, self.dbound = xbound, ybound, zbound, dbound
        dx, bx, nx = gen_dx_bx(xbound, ybound, zbound)
        self.dx = nn.Parameter(dx, requires_grad=False)
        self.bx = nn.Parameter(bx, requires_grad=False)
        self.nx = nn.Parameter(nx, requires_grad=False)
        self.C = out_channels
        # None = the reference's behaviour (whole view transform in fp32, BF/bevfusion.py:177).  torch.bfloat16 runs
        # only the dense conv stacks (dtransform / depthnet / downsample) under autocast; geometry, ranks, the
        # softmax output and the lift-splat pooling stay fp32 ("bf16 with fp32 index paths", BASELINE configs[3]).
        self.conv_dtype = None
        self.frustum = self.create_frustum()
        self.D = self.frustum.shape[0]
        self.fp16_enabled = False
        # host copies for the plan (no per-step D2H)
        self._nx_host = [int(v) for v in nx]
        self._dx_host = [float(v) for v in dx]
        self._origin_host = [float(v) for v in (bx - dx / 2.0)]  # fp32 tensor arithmetic, as the reference

    def create_frustum(self):
        """(reference :53-66) [D, fH, fW, 3] = (pixel x, pixel y, depth)."""
        iH, iW = self.image_size
        fH, fW = self.feature_size
        ds = torch.arange(*self.dbound, dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
        D = ds.shape[0]
        xs = torch.linspace(0, iW - 1, fW, dtype=torch.float).view(1, 1, fW).expand(D, fH, fW)
        ys = torch.linspace(0, iH - 1, fH, dtype=torch.float).view(1, fH, 1).expand(D, fH, fW)
        return nn.Parameter(torch.stack((xs, ys, ds), -1), requires_grad=False)

    # ------------------------------------------------------------------ device plan
    def make_plan(self, camera2lidar_rots, camera2lidar_trans, intrins_inverse, post_rots_inverse, post_trans,
                  extra_rots=None, extra_trans=None, with_reference_outputs=False, with_geometry=False):
        B, N, _ = camera2lidar_trans.shape
        dev = camera2lidar_trans.device
        fH, fW = self.feature_size
        plan = BevPlan(B, N, self.D, fH * fW, self._nx_host, dev, with_reference_outputs, with_geometry)
        combine = camera2lidar_rots.matmul(intrins_inverse)  # (reference :93)
        if extra_rots is None:
            extra_rots = torch.eye(3, device=dev).expand(B, 3, 3)
        if extra_trans is None:
            extra_trans = torch.zeros(B, 3, device=dev)
        return plan.build(self.frustum.reshape(-1, 3), post_trans.reshape(B * N, 3), post_rots_inverse.reshape(B * N, 9),
                          combine.reshape(B * N, 9), camera2lidar_trans.reshape(B * N, 3), extra_rots.reshape(B, 9),
                          extra_trans.reshape(B, 3), self._origin_host, self._dx_host)

    def get_geometry(self, camera2lidar_rots, camera2lidar_trans, intrins_inverse, post_rots_inverse, post_trans,
                     **kwargs):
        """(reference :68-112) materialised lidar-frame frustum [B,N,D,fH,fW,3]; computed by the plan kernel."""
        B, N, _ = camera2lidar_trans.shape
        plan = self.make_plan(camera2lidar_rots, camera2lidar_trans, intrins_inverse, post_rots_inverse, post_trans,
                              kwargs.get("extra_rots"), kwargs.get("extra_trans"), with_geometry=True)
        fH, fW = self.feature_size
        return plan.geom_xyz.view(B, N, self.D, fH, fW, 3)

    def get_cam_feats(self, x):
        raise NotImplementedError

    def bev_pool_aux(self, geom_feats):
        """(reference :118-176) geom [B,N,D,H,W,3] -> (geom_feats int[nk,4], kept bool[N'], ranks int64[nk],
        indices int64[nk]).  Kept for API parity (materialised geometry in, one host sync for nk);
        the sort is stable, so `indices` is deterministic (the reference's argsort is not)."""
        B, N, D, H, W, C = geom_feats.shape
        assert C == 3
        cells = ((geom_feats - (self.bx - self.dx / 2.0)) / self.dx).long().view(-1, 3)
        nprime = cells.shape[0]
        batch_ix = torch.arange(nprime, device=cells.device) // (nprime // B)
        cells = torch.cat((cells, batch_ix[:, None]), 1)
        kept = ((cells[:, :3] >= 0) & (cells[:, :3] < self.nx)).all(1)
        cells = cells[kept]
        Dz, Hx, Wy = self.nx[2], self.nx[0], self.nx[1]
        ranks = cells[:, 0] * (Wy * Dz * B) + cells[:, 1] * (Dz * B) + cells[:, 2] * B + cells[:, 3]
        ranks, indices = torch.sort(ranks, stable=True)
        return cells[indices], kept, ranks, indices

    def bev_pool(self, x, geom_feats):
        """(reference :179-204) op-boundary path: x [B,N,D,H,W,C] materialised, geom [B,N,D,H,W,3]."""
        B, N, D, H, W, C = x.shape
        x = x.reshape(B * N * D * H * W, C)
        geom_feats, kept, ranks, indices = self.bev_pool_aux(geom_feats)
        return self.bev_pool_precomputed(x.reshape(B, N, D, H, W, C), geom_feats, kept, ranks, indices)

    def bev_pool_precomputed(self, x, geom_feats, kept, ranks, indices):
        """(reference :206-223)"""
        B, N, D, H, W, C = x.shape
        x = x.reshape(B * N * D * H * W, C)[kept]
        assert x.shape[0] == geom_feats.shape[0]
        x = x[indices]
        x = bev_pool(x, geom_feats, ranks, B, self.nx[2], self.nx[0], self.nx[1], self.training)
        return torch.cat(x.unbind(dim=2), 1)  # collapse Z: [B, C*nz, nx, ny]

    # ------------------------------------------------------------------ fused path
    def lift_splat_bev(self, depth, feat, plan):
        """depth [BN, D, fH, fW], feat [BN, C, fH, fW] (the reference's layouts, :699-701) ->
        [B, C*nz, nx, ny], identical to bev_pool(depth (x) feat, geom)."""
        BN, D, fH, fW = depth.shape
        C = feat.shape[1]
        depth_pm = depth.permute(0, 2, 3, 1).reshape(BN * fH * fW, D)  # free if channels_last
        feat_pm = feat.permute(0, 2, 3, 1).reshape(BN * fH * fW, C)
        if depth_pm.stride(1) != 1:
            depth_pm = depth_pm.contiguous()
        # bf16 features (a bf16 depthnet, get_depth_and_feat) are gathered as stored when C allows 16-byte pieces
        f16 = BF16_FEAT and feat_pm.dtype == torch.bfloat16 and C % 8 == 0
        if not f16:
            feat_pm = feat_pm.float()
        per16 = 16 // feat_pm.element_size()
        if feat_pm.stride(1) != 1 or feat_pm.stride(0) % per16 or feat_pm.data_ptr() % 16:
            feat_pm = feat_pm.contiguous()
        # with bf16 conv stacks behind it (conv_dtype) the BEV map leaves the kernel in bf16: the downsample convolution's cast
        # and its backward (two 40 MB copies per step) disappear, the values are the ones that cast would have produced
        # (bit-identical model results; step time unchanged within noise: 33.79 vs 33.72 ms over three runs each)
        out_dtype = torch.bfloat16 if (BF16_BEV_OUT and getattr(self, "conv_dtype", None) == torch.bfloat16) else torch.float32
        out = lift_splat(depth_pm.float(), feat_pm, plan, out_dtype)  # [B, nz, nx, ny, C]
        if out.shape[1] == 1:
            # nz == 1 (every BEVFusion config): [B, nx, ny, C] IS the channels-last memory of [B, C, nx, ny] -> no copy
            # (a reshape, not out[:, 0]: the backward of a select is a zero-fill + copy)
            B_, _, X_, Y_, C_ = out.shape
            return out.view(B_, X_, Y_, C_).permute(0, 3, 1, 2)
        out = out.permute(0, 4, 1, 2, 3)  # [B, C, nz, nx, ny]
        return torch.cat(out.unbind(dim=2), 1).contiguous()

    def _calibration(self, camera_intrinsics, camera2lidar, img_aug_matrix, lidar_aug_matrix):
        intrins = camera_intrinsics[..., :3, :3]
        post_rots = img_aug_matrix[..., :3, :3]
        return dict(camera2lidar_rots=camera2lidar[..., :3, :3], camera2lidar_trans=camera2lidar[..., :3, 3],
                    intrins_inverse=_inverse(intrins), post_rots_inverse=_inverse(post_rots),
                    post_trans=img_aug_matrix[..., :3, 3], extra_rots=lidar_aug_matrix[..., :3, :3],
                    extra_trans=lidar_aug_matrix[..., :3, 3])

    def forward(self, img, points, lidar2image, camera_intrinsics, camera2lidar, img_aug_matrix, lidar_aug_matrix,
                metas=None, camera_intrinsics_inverse=None, img_aug_matrix_inverse=None,
                lidar_aug_matrix_inverse=None, geom_feats_precomputed=None):
        """(reference :225-270) geom_feats_precomputed may be a BevPlan (fused path) or the reference's
        (geom_feats, kept, ranks, indices) tuple (op-boundary path)."""
        if isinstance(geom_feats_precomputed, BevPlan):
            plan = geom_feats_precomputed
        elif geom_feats_precomputed is not None:
            geom_feats, kept, ranks, indices = geom_feats_precomputed[:4]
            return self.bev_pool_precomputed(self.get_cam_feats(img), geom_feats, kept, ranks, indices)
        else:
            plan = self.make_plan(**self._calibration(camera_intrinsics, camera2lidar, img_aug_matrix, lidar_aug_matrix))
        depth, feat = self.get_depth_and_feat(img)
        return self.lift_splat_bev(depth, feat, plan)


@MODELS.register_module()
class LSSTransform(BaseViewTransform):
    """(reference :273-339) depthnet = 1x1 conv producing D depth logits + C features."""

    def __init__(self, in_channels, out_channels, image_size, feature_size, xbound, ybound, zbound, dbound,
                 downsample: int = 1) -> None:
        super().__init__(in_channels, out_channels, image_size, feature_size, xbound, ybound, zbound, dbound)
        self.depthnet = nn.Conv2d(in_channels, self.D + self.C, 1)
        self.downsample = _make_downsample(out_channels, downsample)

    def get_depth_and_feat(self, x):
        B, N, C, fH, fW = x.shape
        x = self.depthnet(x.reshape(B * N, C, fH, fW))
        return x[:, :self.D].softmax(dim=1), x[:, self.D:self.D + self.C]

    def get_cam_feats(self, x):
        """Materialised [B,N,D,fH,fW,C] (reference :320-334), for the op-boundary path."""
        B, N = x.shape[:2]
        depth, feat = self.get_depth_and_feat(x)
        fH, fW = depth.shape[-2:]
        out = depth.unsqueeze(1) * feat.unsqueeze(2)
        return out.reshape(B, N, self.C, self.D, fH, fW).permute(0, 1, 3, 4, 5, 2)

    def forward(self, *args, **kwargs):
        return self.downsample(super().forward(*args, **kwargs))


def _make_downsample(out_channels, downsample):
    if downsample <= 1:
        return nn.Identity()
    assert downsample == 2, downsample
    return nn.Sequential(
        Conv2d(out_channels, out_channels, 3, padding=1, bias=False), *bn_act(out_channels),
        Conv2d(out_channels, out_channels, 3, stride=downsample, padding=1, bias=False),
        *bn_act(out_channels),
        Conv2d(out_channels, out_channels, 3, padding=1, bias=False), *bn_act(out_channels))


class BaseDepthTransform(BaseViewTransform):
    """(reference :342-551) adds the sparse LiDAR depth image as an input of get_cam_feats."""

    def rasterise_depth(self, img, points, lidar2image, img_aug_matrix, lidar_aug_matrix, lidar_aug_matrix_inverse=None,
                        with_histogram=False):
        """LiDAR points -> sparse depth images [B, N, 1, iH, iW] (reference :363-449), one HIP launch pair per
        sample (csrc/raster.hip).  Points are not modified (the reference mutates its argument in place, SURVEY
        appendix 10).  A pixel hit by several points keeps the last one (unspecified in the reference, :410-417).
        with_histogram: also return the un-normalised GT depth-bin counts [B, N, fH, fW, D] of :636-670,
        accumulated in the same pass."""
        if lidar_aug_matrix_inverse is None:
            lidar_aug_matrix_inverse = _inverse(lidar_aug_matrix)
        B = len(points)
        N = img if isinstance(img, int) else img.shape[1]   # only the camera count is needed from the image tensor
        iH, iW = self.image_size
        fH, fW = self.feature_size
        dev = points[0].device
        depth = torch.empty(B, N, 1, iH, iW, device=dev, dtype=torch.float32)
        counts = torch.zeros(B, N, fH, fW, self.D, device=dev, dtype=torch.float32) if with_histogram else None
        inv_rot = lidar_aug_matrix_inverse[:, :3, :3].contiguous().float()
        aug_t = lidar_aug_matrix[:, :3, 3].contiguous().float()
        l2i = lidar2image.contiguous().float()
        ia = img_aug_matrix.contiguous().float()
        lib = _lib.load()
        nbytes = lib.bfhip_rasterise_depth_workspace_bytes(N, iH, iW)
        if getattr(self, "_raster_ws", None) is None or self._raster_ws.numel() < nbytes or self._raster_ws.device != dev:
            self._raster_ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dbound = _lib.host_f32(self.dbound)
        with torch.cuda.device(dev):
            stream = _lib.stream_of(depth)
            for b in range(B):
                p = points[b].contiguous().float()
                rc = lib.bfhip_rasterise_depth(_lib.ptr(p), p.shape[0], p.shape[1], _lib.ptr(inv_rot[b]), _lib.ptr(aug_t[b]),
                                               _lib.ptr(l2i[b]), _lib.ptr(ia[b]), N, iH, iW, _lib.ptr(depth[b]),
                                               _lib.ptr(counts[b]) if with_histogram else None, fH, fW, self.D, dbound,
                                               _lib.ptr(self._raster_ws), self._raster_ws.numel(), stream)
                _lib.check(rc, "rasterise_depth")
        return (depth, counts) if with_histogram else depth

    def depth_distribution(self, counts=None, depth=None):
        """GT depth distribution (reference :636-686): from pre-accumulated `counts` [B,N,fH,fW,D] or from a depth
        image [BN,1,iH,iW].  Returns (gt_depth_distr, counts_3d) with bin 0 cleared."""
        iH, iW = self.image_size
        fH, fW = self.feature_size
        if counts is None:
            BN = depth.shape[0]
            counts = torch.empty(BN, fH, fW, self.D, device=depth.device, dtype=torch.float32)
            depth = depth.contiguous().float()
        else:
            BN = counts.shape[0] * counts.shape[1]
        distr = torch.empty_like(counts)
        with torch.cuda.device(counts.device):
            rc = _lib.load().bfhip_depth_histogram(_lib.ptr(depth) if depth is not None else None, BN, iH, iW, fH, fW, self.D,
                                                   _lib.host_f32(self.dbound), _lib.ptr(counts), _lib.ptr(distr),
                                                   _lib.stream_of(counts))
        _lib.check(rc, "depth_histogram")
        return distr, counts

    def prepare(self, n_cams, points, lidar2image, cam_intrinsic, camera2lidar, img_aug_matrix, lidar_aug_matrix,
                lidar_aug_matrix_inverse=None, geom_feats_precomputed=None):
        """Everything of forward() that does not need the image features: sparse depth images + GT depth histogram
        (reference :372-449,636-686), the BEV plan (:68-176) and, in DepthLSSTransform, the dtransform conv stack on the depth
        images (:581-591).  The detector runs it on the LiDAR side stream while the image backbone occupies the main one
        (bevfusion.BEVFusion.extract_feat); forward(..., prepared=...) then starts at the depthnet."""
        depth_img, counts = self.rasterise_depth(n_cams, points, lidar2image, img_aug_matrix, lidar_aug_matrix,
                                                 lidar_aug_matrix_inverse, with_histogram=True)
        if isinstance(geom_feats_precomputed, BevPlan):
            plan = geom_feats_precomputed
        else:
            plan = self.make_plan(**self._calibration(cam_intrinsic, camera2lidar, img_aug_matrix, lidar_aug_matrix))
        prep = dict(depth_img=depth_img, counts=counts, plan=plan)
        prep.update(self.prepare_depth(depth_img, counts))
        return prep

    def prepare_depth(self, depth_img, counts):
        return {}

    @staticmethod
    def prepared_tensors(prep):
        """Every tensor a `prepare()` result holds (for record_stream when it was produced on another stream)."""
        out = [v for v in prep.values() if torch.is_tensor(v)]
        plan = prep.get("plan")
        if plan is not None:
            out += [v for v in vars(plan).values() if torch.is_tensor(v)]
        return out

    def forward(self, img, points, lidar2image, cam_intrinsic, camera2lidar, img_aug_matrix, lidar_aug_matrix,
                metas=None, camera_intrinsics_inverse=None, img_aug_matrix_inverse=None,
                lidar_aug_matrix_inverse=None, geom_feats_precomputed=None, prepared=None):
        if prepared is None:
            prepared = self.prepare(img.shape[1], points, lidar2image, cam_intrinsic, camera2lidar, img_aug_matrix,
                                    lidar_aug_matrix, lidar_aug_matrix_inverse, geom_feats_precomputed)
        depth_img, counts, plan = prepared["depth_img"], prepared["counts"], prepared["plan"]
        depth, feat, est_depth_distr, gt_depth_distr, counts_3d = self.get_depth_and_feat(img, depth_img, counts, prepared)
        x = self.lift_splat_bev(depth, feat, plan)
        if self.training:
            # depth cross-entropy on cells that hold LiDAR returns (reference :540-547).  The reference computes it and never adds
            # it to the losses (BF/bevfusion.py:388-392), so nothing downstream waits for it: with an auxiliary stream its half-dozen
            # passes over [P, D] leave the main queue.  The stream must NOT be the LiDAR side stream: waiting for the main stream
            # here would hold the LiDAR branch, queued behind it, until the camera forward has finished (measured: +3 ms per
            # step); `aux_stream` is left unset by the detector
            aux = getattr(self, "aux_stream", None)
            if aux is not None and est_depth_distr.is_cuda:
                aux.wait_stream(torch.cuda.current_stream(est_depth_distr.device))
                for t in (est_depth_distr, gt_depth_distr, counts_3d):
                    t.record_stream(aux)
                with torch.cuda.stream(aux):
                    depth_loss = self._depth_loss(est_depth_distr, gt_depth_distr, counts_3d)
            else:
                depth_loss = self._depth_loss(est_depth_distr, gt_depth_distr, counts_3d)
        else:
            depth_loss = 0.0
        return x, depth_loss

    def _depth_loss(self, est_depth_distr, gt_depth_distr, counts_3d):
        mask_flat = counts_3d.sum(dim=-1).view(-1) > 0
        gt = gt_depth_distr.view(-1, self.D)
        est = est_depth_distr.reshape(-1, self.D)
        cross_ent = -torch.sum(gt * torch.log(est + 1e-8), dim=-1)
        return torch.sum(cross_ent * mask_flat.float()) / (mask_flat.sum() + 1e-8)


class _DepthLift(torch.autograd.Function):
    """y [BN, H, W, 8] = bias + d [BN, H, W, 1] * weight -- Conv2d(1, 8, 1) on the one-channel depth image (reference :592-594) in
    bf16, written channels-last.  Backward: csrc/raster.hip depth_lift_bwd_kernel (dw = sum dy * d, db = sum dy in one pass)."""

    @staticmethod
    def forward(ctx, d, weight, bias):
        y = torch.addcmul(bias.to(d.dtype), d, weight.reshape(1, 1, 1, -1).to(d.dtype))
        ctx.save_for_backward(d)
        ctx.dtypes = (weight.dtype, bias.dtype, weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (d,) = ctx.saved_tensors
        wdt, bdt, wshape = ctx.dtypes
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous() or dy.data_ptr() % 16:
            dy = dy.to(torch.bfloat16).contiguous()
        from . import _lib
        out = torch.empty(16, dtype=torch.float32, device=dy.device)
        ws = torch.empty(_lib.call_size("bfhip_depth_lift_bwd_workspace_bytes"), dtype=torch.uint8, device=dy.device)
        _lib.call("bfhip_depth_lift_bwd", dy.data_ptr(), d.data_ptr(), d.numel(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                  _lib.stream_of(dy))
        return None, out[8:].to(wdt).reshape(wshape), out[:8].to(bdt)


@MODELS.register_module()
class DepthLSSTransform(BaseDepthTransform):
    """(reference :555-733) dtransform on the sparse depth image + depthnet on cat(depth feats, image feats)."""

    def __init__(self, in_channels, out_channels, image_size, feature_size, xbound, ybound, zbound, dbound,
                 downsample: int = 1) -> None:
        super().__init__(in_channels, out_channels, image_size, feature_size, xbound, ybound, zbound, dbound)
        self.dtransform = nn.Sequential(
            nn.Conv2d(1, 8, 1), *bn_act(8),
            Conv2d(8, 32, 5, stride=4, padding=2), *bn_act(32),
            Conv2d(32, 64, 5, stride=2, padding=2), *bn_act(64))
        self.depthnet = nn.Sequential(
            Conv2d(in_channels + 64, in_channels, 3, padding=1), *bn_act(in_channels),
            Conv2d(in_channels, in_channels, 3, padding=1), *bn_act(in_channels),
            Conv2d(in_channels, self.D + self.C, 1))
        self.downsample = _make_downsample(out_channels, downsample)

    def run_dtransform(self, d):
        """dtransform with its first layer produced directly in channels-last memory.  Conv2d(1, 8, 1) on the one-channel
        depth image is a per-pixel outer product with 8 weights; as a conv its output comes back NCHW (a one-channel
        input carries no layout), which sends the following BatchNorm over [BN, 8, 256, 704] down the slow generic path
        (1.9 ms forward + 2.4 ms backward per batch of 4 frames)."""
        conv0 = self.dtransform[0]
        if not (d.is_cuda and d.dim() == 4 and d.shape[1] == 1 and conv0.kernel_size == (1, 1) and conv0.stride == (1, 1)):
            return self.dtransform(d)
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else d.dtype
        BN, _, H, W = d.shape
        if (dt == torch.bfloat16 and conv0.out_channels == 8 and conv0.bias is not None and torch.is_grad_enabled()
                and conv0.weight.requires_grad and not d.requires_grad):
            # same forward; the two parameter gradients in one pass over dy instead of a product tensor and two long reductions
            y = _DepthLift.apply(d.reshape(BN, H, W, 1).to(dt), conv0.weight, conv0.bias).permute(0, 3, 1, 2)
        else:
            w = conv0.weight.reshape(1, 1, 1, -1).to(dt)
            b = (conv0.bias if conv0.bias is not None else conv0.weight.new_zeros(w.shape[-1])).to(dt)
            y = torch.addcmul(b, d.reshape(BN, H, W, 1).to(dt), w).permute(0, 3, 1, 2)  # [BN, 8, H, W], channels-last strides
        for layer in list(self.dtransform)[1:]:
            y = layer(y)
        return y

    def gt_depth_distribution(self, d, B, N):
        """Histogram of the sparse depth image over (feature cell, depth bin) (reference :636-686); d [BN,1,iH,iW]."""
        distr, counts = self.depth_distribution(depth=d)
        fH, fW = self.feature_size
        return distr.view(B, N, fH, fW, self.D), counts.view(B, N, fH, fW, self.D)

    def prepare_depth(self, depth_img, counts):
        """GT depth distribution + dtransform features of the sparse depth images (no image features needed)."""
        B, N = depth_img.shape[:2]
        d = depth_img.reshape(B * N, *depth_img.shape[2:])
        if counts is not None:  # accumulated by the rasteriser in the same pass
            gt_depth_distr, counts_3d = self.depth_distribution(counts=counts)
        else:
            gt_depth_distr, counts_3d = self.gt_depth_distribution(d, B, N)
        with torch.autocast("cuda", dtype=self.conv_dtype or torch.bfloat16, enabled=self.conv_dtype is not None):
            dfeat = self.run_dtransform(d)
        return dict(gt_depth_distr=gt_depth_distr, counts_3d=counts_3d, dfeat=dfeat)

    def get_depth_and_feat(self, x, d, counts=None, prepared=None):
        B, N, C, fH, fW = x.shape
        BN = B * N
        x = x.reshape(BN, C, fH, fW)
        if prepared is None or "dfeat" not in prepared:
            prepared = self.prepare_depth(d, counts)
        gt_depth_distr, counts_3d, dfeat = prepared["gt_depth_distr"], prepared["counts_3d"], prepared["dfeat"]
        with torch.autocast("cuda", dtype=self.conv_dtype or torch.bfloat16, enabled=self.conv_dtype is not None):
            x = self.depthnet(torch.cat([dfeat, x], dim=1))
        if BF16_FEAT and x.dtype == torch.bfloat16 and x.is_cuda and self.C % 8 == 0:
            # the reference widens the whole [BN, D + C, fH, fW] tensor (x = x.float()).  Same values, less traffic: only the D
            # depth logits are widened (the softmax runs in fp32); the C feature channels stay as the bf16 convolution stored
            # them and the fused lift-splat gathers them in that form (lift_splat_bev)
            logits, feat_pm = _SplitDepthFeat.apply(x.permute(0, 2, 3, 1).reshape(BN * fH * fW, x.shape[1]), self.D, self.C)
            depth = logits.view(BN, fH, fW, self.D).softmax(dim=-1).permute(0, 3, 1, 2)   # [BN, D, fH, fW], channels-last memory
            feat = feat_pm.view(BN, fH, fW, self.C).permute(0, 3, 1, 2)
        else:
            x = x.float()
            depth = x[:, :self.D].softmax(dim=1)
            feat = x[:, self.D:self.D + self.C]
        est_depth_distr = depth.permute(0, 2, 3, 1).reshape(B, N, fH, fW, self.D)
        if self.training:
            depth_aux = gt_depth_distr.view(BN, fH, fW, self.D).permute(0, 3, 1, 2)
            depth = depth + (torch.maximum(depth_aux, depth) - depth).detach()  # straight-through (reference :702-706)
        return depth, feat, est_depth_distr, gt_depth_distr, counts_3d

    def get_cam_feats(self, x, d):
        """Materialised outer product + aux outputs (reference :624-727)."""
        B, N = x.shape[:2]
        depth, feat, est, gt, counts = self.get_depth_and_feat(x, d)
        fH, fW = depth.shape[-2:]
        out = depth.unsqueeze(1) * feat.unsqueeze(2)
        return out.reshape(B, N, self.C, self.D, fH, fW).permute(0, 1, 3, 4, 5, 2), est, gt, counts

    def forward(self, *args, **kwargs):
        x, depth_loss = super().forward(*args, **kwargs)
        with torch.autocast("cuda", dtype=self.conv_dtype or torch.bfloat16, enabled=self.conv_dtype is not None):
            x = self.downsample(x)
        return x, depth_loss
