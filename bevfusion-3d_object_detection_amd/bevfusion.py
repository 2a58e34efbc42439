"""Host mirror of the detector wiring, projects/BEVFusion/bevfusion/bevfusion.py (BEVFusion :23-399):
voxelize loop + mean reduce, camera branch, LiDAR branch, fusion, BEV backbone/neck, head.

Inputs follow the reference's `batch_inputs_dict` / metainfo contract (BF/bevfusion.py:300-322):
  points: list[Tensor[Ni, F]], imgs: Tensor[B, N, 3, H, W], and per-sample 4x4 matrices
  lidar2img, cam2img, cam2lidar, img_aug_matrix, lidar_aug_matrix.
"""
from typing import Dict, List, Optional

import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .ops import Voxelization
from .ops.voxel import voxel_layer
from .registry import MODELS
from . import dense_modules, depth_lss, sparse_encoder  # noqa: F401  (register the module names)


def voxel_mean(voxels, num_points):
    """feats.sum(dim=1) / sizes (BF/bevfusion.py:251-253) as one kernel (bfhip_voxel_mean)."""
    voxels = voxels.contiguous()
    M, P, Fdim = voxels.shape
    out = torch.empty((M, Fdim), dtype=torch.float32, device=voxels.device)
    with torch.cuda.device(voxels.device):
        rc = _lib.load().bfhip_voxel_mean(_lib.ptr(voxels), _lib.ptr(num_points.contiguous()), M, P, Fdim, _lib.ptr(out),
                                          _lib.stream_of(voxels))
    _lib.check(rc, "voxel_mean")
    return out


@MODELS.register_module()
class BEVFusion(nn.Module):

    def __init__(self, data_preprocessor: Optional[dict] = None, pts_voxel_encoder: Optional[dict] = None,
                 pts_middle_encoder: Optional[dict] = None, fusion_layer: Optional[dict] = None,
                 img_backbone: Optional[dict] = None, pts_backbone: Optional[dict] = None,
                 view_transform: Optional[dict] = None, img_neck: Optional[dict] = None,
                 pts_neck: Optional[dict] = None, bbox_head: Optional[dict] = None, **kwargs) -> None:
        super().__init__()
        voxelize_cfg = dict(data_preprocessor["voxelize_cfg"])
        self.voxelize_reduce = voxelize_cfg.pop("voxelize_reduce")
        if isinstance(voxelize_cfg.get("max_voxels"), list):
            voxelize_cfg["max_voxels"] = tuple(voxelize_cfg["max_voxels"])
        self.pts_voxel_layer = Voxelization(**voxelize_cfg)
        # pts_voxel_encoder (HardSimpleVFE) is built but never called by the reference (:52, reduce is inline)
        build = lambda cfg: MODELS.build(cfg) if cfg is not None else None
        self.img_backbone = build(img_backbone)
        self.img_neck = build(img_neck)
        self.view_transform = build(view_transform)
        self.pts_middle_encoder = build(pts_middle_encoder)
        self.fusion_layer = build(fusion_layer)
        self.pts_backbone = build(pts_backbone)
        self.pts_neck = build(pts_neck)
        self.bbox_head = build(bbox_head)
        self.lidar_side_stream = False  # opt-in: run the LiDAR branch on a second HIP stream (bench.py enables it)
        self._side_stream = None
        # with the side stream: also run the image-independent prologue of the view transform there (BFHIP_SIDE_PREPARE=0: A/B)
        self.side_prepare = os.environ.get("BFHIP_SIDE_PREPARE", "1") == "1"
        # static capacity mode of the LiDAR branch: buffers sized by (grow-only) bounds learnt from earlier frames, every row
        # count on the device, ZERO host reads per forward; the first forward runs the exact path to learn them.
        # Opt-in (BFHIP_STATIC_LIDAR=1, or a captured hipGraph, which needs it): on the side stream the two host reads of the
        # exact path cost nothing measurable, while capacity-sized buffers add 0.6 ms (slack 1.05) to 1.3 ms (slack 1.5) of
        # GPU work per batch-4 step (36.9 exact vs 37.5 / 37.7 / 38.2 ms, same box, bench.py `full`)
        self.static_lidar = os.environ.get("BFHIP_STATIC_LIDAR", "0") == "1"
        self._voxel_cap = None
        self._voxel_monitor = None
        self._voxel_status = None  # static capacity mode: device bool, this forward's voxels exceeded the row capacity

    # ------------------------------------------------------------------ LiDAR branch
    @torch.no_grad()
    def voxelize(self, points: List[torch.Tensor]):
        """(reference :227-255) per-sample hard voxelization, batch id prepended -> (b, x, y, z), mean reduce."""
        feats, coords, sizes = [], [], []
        layer = self.pts_voxel_layer
        if layer.max_num_points != -1 and (layer.max_voxels[0] if self.training else layer.max_voxels[1]) != -1:
            # hard voxelization: launch every sample first (sync-free C ABI), read all B counts with ONE host sync
            # (the reference syncs per sample: voxelization_cuda.cu:369-370)
            max_voxels = layer.max_voxels[0] if self.training else layer.max_voxels[1]
            bufs, counts = [], torch.empty(len(points), dtype=torch.int32, device=points[0].device)
            for k, res in enumerate(points):
                res = res.contiguous()
                v = res.new_zeros((max_voxels, layer.max_num_points, res.size(1)))
                c = res.new_zeros((max_voxels, 3), dtype=torch.int)
                n = res.new_zeros((max_voxels,), dtype=torch.int)
                voxel_layer.hard_voxelize_async(res, v, c, n, layer.voxel_size, layer.point_cloud_range,
                                                layer.max_num_points, max_voxels, counts[k:k + 1])
                bufs.append((v, c, n))
            ms = counts.tolist()
            self._note_voxel_total(sum(ms))
            rets = [(v[:m], c[:m], n[:m]) for (v, c, n), m in zip(bufs, ms)]
        else:
            rets = [layer(res) for res in points]
        for k, ret in enumerate(rets):
            if len(ret) == 3:
                f, c, n = ret
            else:
                assert len(ret) == 2
                f, c = ret
                n = None
            feats.append(f)
            coords.append(F.pad(c, (1, 0), mode="constant", value=k))
            if n is not None:
                sizes.append(n)
        feats = torch.cat(feats, dim=0)
        coords = torch.cat(coords, dim=0)
        if len(sizes) > 0:
            sizes = torch.cat(sizes, dim=0)
            if self.voxelize_reduce:
                feats = voxel_mean(feats, sizes)
        return feats, coords, sizes

    def _note_voxel_total(self, total):
        from .spconv import round_capacity
        self._voxel_cap = max(self._voxel_cap or 0, round_capacity(total))

    @torch.no_grad()
    def voxelize_static(self, points: List[torch.Tensor]):
        """voxelize() without host reads: one [B, max_voxels, ...] buffer set, per-sample voxelization with the counts left on
        the device, then ONE kernel that builds the mean features and (b, x, y, z) coordinates of all samples in a
        capacity-sized matrix whose active rows are the prefix (bfhip_voxel_compact_mean).  -> feats [cap, F],
        coords i32[cap, 4] (inactive rows: b = -1), n_valid i32[1] on the device."""
        from .spconv import CapacityMonitor
        layer = self.pts_voxel_layer
        B, dev = len(points), points[0].device
        max_voxels = layer.max_voxels[0] if self.training else layer.max_voxels[1]
        P, Fdim = layer.max_num_points, points[0].size(1)
        if self._voxel_monitor is None:
            self._voxel_monitor = CapacityMonitor()
        seen = self._voxel_monitor.poll()
        if seen is not None:
            if seen[1] > self._voxel_cap:
                import warnings
                warnings.warn("voxelize_static: %d voxels exceeded the row capacity %d; capacity grown" % (seen[1], self._voxel_cap))
            self._note_voxel_total(seen[1])
        cap = self._voxel_cap
        voxels = points[0].new_zeros((B, max_voxels, P, Fdim))
        coors = points[0].new_zeros((B, max_voxels, 3), dtype=torch.int)
        num = points[0].new_zeros((B, max_voxels), dtype=torch.int)
        counts = torch.empty(B, dtype=torch.int32, device=dev)
        for k, res in enumerate(points):
            voxel_layer.hard_voxelize_async(res.contiguous(), voxels[k], coors[k], num[k], layer.voxel_size, layer.point_cloud_range,
                                            P, max_voxels, counts[k:k + 1])
        feats = torch.empty((cap, Fdim), dtype=torch.float32, device=dev)
        coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        n_total = torch.empty(2, dtype=torch.int32, device=dev)
        _lib.call("bfhip_voxel_compact_mean", voxels.data_ptr(), coors.data_ptr(), num.data_ptr(), counts.data_ptr(), B, max_voxels,
                  P, Fdim, cap, feats.data_ptr(), coords.data_ptr(), n_total.data_ptr(), _lib.stream_of(feats))
        self._voxel_monitor.submit(n_total)
        self._voxel_status = n_total[1] > cap
        return feats, coords, n_total[0:1]

    def _static_lidar_ready(self):
        layer, enc = self.pts_voxel_layer, self.pts_middle_encoder
        hard = layer.max_num_points != -1 and (layer.max_voxels[0] if self.training else layer.max_voxels[1]) != -1
        return (self.static_lidar and hard and self.voxelize_reduce and self.training and self._voxel_cap is not None
                and getattr(enc, "static_caps", None) is not None)

    def extract_pts_feat(self, batch_inputs_dict) -> torch.Tensor:
        points = batch_inputs_dict["points"]
        if points[0].is_cuda and self._static_lidar_ready():
            with torch.autocast("cuda", enabled=False):
                feats, coords, n_valid = self.voxelize_static([p.float() for p in points])
            return self.pts_middle_encoder(feats, coords, len(points), n_valid=n_valid)
        self._voxel_status = None
        with torch.autocast("cuda", enabled=False):  # fp32 island = voxelization only, as the reference (:201-206)
            points = [p.float() for p in points]
            feats, coords, sizes = self.voxelize(points)
            batch_size = len(points)  # the reference reads coords[-1, 0] + 1 from the device (:206)
        # the encoder runs under the caller's autocast (the reference's spconv layers run in half precision under AMP);
        # here: bf16-input MFMA with fp32 accumulation, fp32 features / BatchNorm / index paths
        return self.pts_middle_encoder(feats, coords, batch_size)

    # ------------------------------------------------------------------ camera branch
    def extract_img_feat(self, x, points, lidar2image, camera_intrinsics, camera2lidar, img_aug_matrix,
                         lidar_aug_matrix, img_metas=None, geom_feats=None, prepared=None):
        B, N, C, H, W = x.size()
        x = self.img_backbone(x.reshape(B * N, C, H, W))
        x = self.img_neck(x)
        if not isinstance(x, torch.Tensor):
            x = x[0]
        BN, C, H, W = x.size()
        x = x.reshape(B, N, C, H, W)
        # the view transform's conv stacks run in its own autocast (conv_dtype); when that is on, the bf16 features go in
        # as they are instead of being widened to fp32 here and narrowed again in front of the first conv
        keep = getattr(self.view_transform, "conv_dtype", None) is not None and x.dtype == self.view_transform.conv_dtype
        with torch.autocast("cuda", enabled=False):  # fp32 island, as the reference (:177)
            if prepared is not None:
                prep, done = prepared
                main = torch.cuda.current_stream(x.device)
                main.wait_event(done)  # depth images, plan and dtransform features came from the side stream
                for t in self.view_transform.prepared_tensors(prep):
                    t.record_stream(main)
                return self.view_transform(x if keep else x.float(), points, lidar2image, camera_intrinsics, camera2lidar,
                                           img_aug_matrix, lidar_aug_matrix, img_metas, geom_feats_precomputed=geom_feats,
                                           prepared=prep)
            return self.view_transform(x if keep else x.float(), points, lidar2image, camera_intrinsics, camera2lidar, img_aug_matrix,
                                       lidar_aug_matrix, img_metas, geom_feats_precomputed=geom_feats)

    def extract_feat(self, batch_inputs_dict: Dict, batch_input_metas=None):
        imgs = batch_inputs_dict.get("imgs", None)
        points = batch_inputs_dict.get("points", None)
        features = []
        depth_loss = 0.0
        # Two orders.  Single stream: LiDAR branch first -- its host reads (voxel counts, one N_out per strided sparse conv)
        # happen while the stream is nearly empty and the long camera branch is queued behind them with no further sync
        # (the reference runs camera first, BF/bevfusion.py:305-361; the fused feature order [img, pts] is unchanged).
        # Two streams (lidar_side_stream): camera branch FIRST on the main stream, then the LiDAR branch on a side stream that
        # only depends on the state of the main stream at entry.  Its host reads wait for the side stream alone, its small
        # kernels overlap the camera forward, and -- because the autograd engine runs the most recently recorded nodes
        # first -- its backward is launched right after the fuser's and overlaps the long camera backward instead of
        # trailing it.
        pts_feature = None
        side = None
        has_cam = imgs is not None and self.view_transform is not None
        has_pts = self.pts_middle_encoder is not None and points is not None
        overlap = bool(has_pts and has_cam and self.lidar_side_stream and points[0].is_cuda)
        if overlap:
            if self._side_stream is None:
                # BFHIP_SIDE_PRIORITY: stream priority of the LiDAR side stream (lower number = higher priority; out-of-range values
                # map to the nearest valid one): the branch has slack (6 ms of work beside 25 ms on the main queue), so it can yield
                self._side_stream = torch.cuda.Stream(device=points[0].device, priority=int(os.environ.get("BFHIP_SIDE_PRIORITY", "0")))
            side, main = self._side_stream, torch.cuda.current_stream(points[0].device)
            entry = torch.cuda.Event()
            entry.record(main)
        elif has_pts:
            pts_feature = self.extract_pts_feat(batch_inputs_dict)
        if has_cam:
            mats = {}
            for key, meta_key in (("lidar2img", "lidar2img"), ("cam2img", "cam2img"), ("cam2lidar", "cam2lidar"),
                                  ("img_aug_matrix", "img_aug_matrix"), ("lidar_aug_matrix", "lidar_aug_matrix")):
                if key in batch_inputs_dict:
                    mats[key] = batch_inputs_dict[key]
                else:
                    default = np.eye(4) if "aug" in key else None
                    mats[key] = imgs.new_tensor(np.asarray([m.get(meta_key, default) for m in batch_input_metas]))
            prepared = None
            if overlap and self.side_prepare and hasattr(self.view_transform, "prepare"):
                # the part of the view transform that needs no image features -- sparse depth images, GT depth histogram, BEV plan,
                # the dtransform conv stack (and, through autograd, its backward) -- runs on the side stream while the image
                # backbone holds the main one; the LiDAR branch follows it there
                side.wait_stream(torch.cuda.current_stream(imgs.device))  # (not `entry`: the matrices may have been built since)
                with torch.cuda.stream(side), torch.autocast("cuda", enabled=False):
                    prep = self.view_transform.prepare(imgs.shape[1], [p.float() for p in points], mats["lidar2img"],
                                                       mats["cam2img"], mats["cam2lidar"], mats["img_aug_matrix"],
                                                       mats["lidar_aug_matrix"],
                                                       geom_feats_precomputed=batch_inputs_dict.get("geom_feats"))
                    done = torch.cuda.Event()
                    done.record(side)
                prepared = (prep, done)
            # the reference passes deepcopy(points) because its rasteriser mutates them (:326); ours does not
            img_feature, depth_loss = self.extract_img_feat(imgs, points, mats["lidar2img"], mats["cam2img"],
                                                            mats["cam2lidar"], mats["img_aug_matrix"],
                                                            mats["lidar_aug_matrix"], batch_input_metas,
                                                            geom_feats=batch_inputs_dict.get("geom_feats"), prepared=prepared)
            features.append(img_feature)
        if overlap:
            # the sparse LiDAR kernels are small for the chip (a few hundred workgroups); on their own HIP stream they
            # (and, through autograd, their backward) overlap the dense camera branch on the main stream
            side.wait_event(entry)
            with torch.cuda.stream(side):
                pts_feature = self.extract_pts_feat(batch_inputs_dict)
        if pts_feature is not None:
            if side is not None:
                main = torch.cuda.current_stream(pts_feature.device)
                main.wait_stream(side)
                pts_feature.record_stream(main)
            features.append(pts_feature)
        if self.fusion_layer is not None:
            x = self.fusion_layer(features)
        else:
            assert len(features) == 1, features
            x = features[0]
        x = self.pts_backbone(x)
        x = self.pts_neck(x)
        return x, depth_loss

    def forward(self, batch_inputs_dict, batch_input_metas=None, batch_data_samples=None):
        """With `batch_data_samples` (ground truth) this is `loss` (one entry point for DDP); otherwise the head
        outputs and the depth loss."""
        if batch_data_samples is not None:
            return self.loss(batch_inputs_dict, batch_data_samples)
        feats, depth_loss = self.extract_feat(batch_inputs_dict, batch_input_metas)
        outs = self.bbox_head(feats, batch_input_metas) if self.bbox_head is not None else feats
        return outs, depth_loss

    def loss(self, batch_inputs_dict, batch_data_samples, **kwargs):
        """BF/bevfusion.py:387-401: dict of the head's losses (the depth loss is computed but not added, :392)."""
        from .head_targets import PackedGT
        metas = None if isinstance(batch_data_samples, PackedGT) else [getattr(d, "metainfo", None) for d in batch_data_samples]
        feats, _ = self.extract_feat(batch_inputs_dict, metas)
        losses = dict(self.bbox_head.loss(feats, batch_data_samples))
        status = self.capacity_status()
        if status is not None:
            # static capacity mode: a frame that overflowed a row capacity was computed on truncated rows (detected on the
            # host only one forward later).  The step must not train on that BEV map: its losses become NaN on the device
            # (no host read) -- multiplied in, so that every gradient is NaN as well -- and the optimizers of this package
            # skip a step whose gradient norm is not finite (amp.skip_nonfinite_step)
            poison = torch.where(status, float("nan"), 1.0)
            for k in losses:
                if "loss" in k:
                    losses[k] = losses[k] * poison
        return losses

    def capacity_status(self):
        """Device bool (or None outside static capacity mode): the last forward of the LiDAR branch exceeded a row capacity."""
        enc = self.pts_middle_encoder
        flags = [f for f in (self._voxel_status, getattr(enc, "capacity_status", None)) if f is not None]
        if not flags:
            return None
        main = torch.cuda.current_stream(flags[0].device)
        for f in flags:
            f.record_stream(main)  # produced on the LiDAR side stream, consumed behind main.wait_stream(side)
        return flags[0] if len(flags) == 1 else flags[0] | flags[1]

    def predict(self, batch_inputs_dict, batch_data_samples=None, **kwargs):
        """BF/bevfusion.py:257-303 without the Det3DDataSample packaging: one dict of boxes/scores/labels per sample."""
        metas = [getattr(d, "metainfo", None) for d in (batch_data_samples or [])] or None
        feats, _ = self.extract_feat(batch_inputs_dict, metas)
        return self.bbox_head.predict(feats, metas)

    @staticmethod
    def parse_losses(losses):
        """BF/bevfusion.py:88-121: -> (loss, log_vars).  Every entry is reduced to its mean (lists of tensors: sum of the
        means), `loss` = sum of the entries whose key contains 'loss', `log_vars` additionally carries that total under
        'loss'.  With an initialised process group every logged scalar is all-reduced and divided by the world size, as the
        reference does for logging (:114-119; the returned `loss` itself stays local, gradients are averaged by the
        gradient exchange).  log_vars values are 0-d DEVICE tensors: the reference calls `.item()` on each (a host read per
        entry per step, :119) -- left to the caller's logger."""
        from collections import OrderedDict
        log_vars = OrderedDict()
        for name, value in losses.items():
            if torch.is_tensor(value):
                log_vars[name] = value.mean()
            elif isinstance(value, (list, tuple)):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError("%s is not a tensor or list of tensors" % name)
        loss = sum(v for k, v in log_vars.items() if "loss" in k)
        log_vars["loss"] = loss
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # ONE collective for all logged scalars instead of the reference's one per entry
            flat = torch.stack([v.detach().float() for v in log_vars.values()])
            dist.all_reduce(flat)
            flat = flat / dist.get_world_size()
            logged = OrderedDict((k, flat[i]) for i, k in enumerate(log_vars))
        else:
            logged = OrderedDict((k, v.detach()) for k, v in log_vars.items())
        return loss, logged


def nuscenes_config(camera=True, lidar=True):
    """Model dict of the reference's nuScenes configs
    (projects/BEVFusion/configs/nuscenes/bevfusion_lidar_voxel0075...py:44-131 and
    bevfusion_lidar-cam_voxel0075...py:9-57) with BASELINE.json's ResNet-50 image backbone."""
    cfg = dict(
        type="BEVFusion",
        data_preprocessor=dict(voxelize_cfg=dict(max_num_points=10, point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0],
                                                 voxel_size=[0.075, 0.075, 0.2], max_voxels=[120000, 160000],
                                                 voxelize_reduce=True)),
        pts_backbone=dict(type="SECOND", in_channels=256, out_channels=[128, 256], layer_nums=[5, 5],
                          layer_strides=[1, 2], norm_cfg=dict(type="BN", eps=0.001, momentum=0.01)),
        pts_neck=dict(type="SECONDFPN", in_channels=[128, 256], out_channels=[256, 256], upsample_strides=[1, 2],
                      norm_cfg=dict(type="BN", eps=0.001, momentum=0.01), use_conv_for_no_stride=True),
        bbox_head=dict(type="BEVFusionHead", num_proposals=200, auxiliary=True, in_channels=512, hidden_channel=128,
                       num_classes=10, nms_kernel_size=3, bn_momentum=0.1, num_decoder_layers=1,
                       decoder_layer=dict(self_attn_cfg=dict(embed_dims=128, num_heads=8, dropout=0.1),
                                          cross_attn_cfg=dict(embed_dims=128, num_heads=8, dropout=0.1),
                                          ffn_cfg=dict(embed_dims=128, feedforward_channels=256, num_fcs=2, ffn_drop=0.1),
                                          pos_encoding_cfg=dict(input_channel=2, num_pos_feats=128)),
                       common_heads=dict(center=[2, 2], height=[1, 2], dim=[3, 2], rot=[2, 2], vel=[2, 2]),
                       train_cfg=dict(dataset="nuScenes", point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0],
                                      grid_size=[1440, 1440, 41], voxel_size=[0.075, 0.075, 0.2], out_size_factor=8,
                                      gaussian_overlap=0.1, min_radius=2, pos_weight=-1,
                                      code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2],
                                      assigner=dict(type="HungarianAssigner3D",
                                                    iou_calculator=dict(type="BboxOverlaps3D", coordinate="lidar"),
                                                    cls_cost=dict(type="mmdet.FocalLossCost", gamma=2.0, alpha=0.25, weight=0.15),
                                                    reg_cost=dict(type="BBoxBEVL1Cost", weight=0.25),
                                                    iou_cost=dict(type="IoU3DCost", weight=0.25))),
                       test_cfg=dict(dataset="nuScenes", grid_size=[1440, 1440, 41], out_size_factor=8,
                                     voxel_size=[0.075, 0.075], pc_range=[-54.0, -54.0], nms_type=None),
                       bbox_coder=dict(type="TransFusionBBoxCoder", pc_range=[-54.0, -54.0],
                                       post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.0,
                                       out_size_factor=8, voxel_size=[0.075, 0.075], code_size=10),
                       loss_cls=dict(type="mmdet.FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean",
                                     loss_weight=1.0),
                       loss_heatmap=dict(type="mmdet.GaussianFocalLoss", reduction="mean", loss_weight=1.0),
                       loss_bbox=dict(type="mmdet.L1Loss", reduction="mean", loss_weight=0.25)),
    )
    if lidar:
        cfg["pts_middle_encoder"] = dict(
            type="BEVFusionSparseEncoder", in_channels=5, sparse_shape=[1440, 1440, 41], order=("conv", "norm", "act"),
            norm_cfg=dict(type="BN1d", eps=0.001, momentum=0.01),
            encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
            encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, (1, 1, 0)), (0, 0)), block_type="basicblock")
    if camera:
        cfg["img_backbone"] = dict(type="ResNet50")
        cfg["img_neck"] = dict(type="GeneralizedLSSFPN", in_channels=[512, 1024, 2048], out_channels=256, start_level=0,
                               num_outs=3, upsample_cfg=dict(mode="bilinear", align_corners=False))
        cfg["view_transform"] = dict(type="DepthLSSTransform", in_channels=256, out_channels=80, image_size=[256, 704],
                                     feature_size=[32, 88], xbound=[-54.0, 54.0, 0.3], ybound=[-54.0, 54.0, 0.3],
                                     zbound=[-10.0, 10.0, 20.0], dbound=[1.0, 60.0, 0.5], downsample=2)
    if camera and lidar:
        cfg["fusion_layer"] = dict(type="ConvFuser", in_channels=[80, 256], out_channels=256)
    elif camera:
        cfg["pts_backbone"]["in_channels"] = 80
    return cfg


def surrogate_loss(outs, depth_loss=0.0):
    """A ground-truth-free scalar over every head output (a Gaussian-focal style term on the dense heat-map plus L1 terms on
    every regression head), so that backward reaches every parameter the real loss reaches.  Used only by the smoke-sized
    plumbing tests in tests/test_model_gpu.py that have no GT boxes; bench.py and the parity tests use the real
    `BEVFusion.loss` -> `BEVFusionHead.loss` (BF/bevfusion_head.py:676-796, Hungarian targets on the device).  Not a
    training objective."""
    res = outs[0][0]
    hm = res["dense_heatmap"].float().sigmoid().clamp(1e-4, 1 - 1e-4)
    loss = -(torch.log(1 - hm) * hm.pow(2)).mean()
    for key in ("center", "height", "dim", "rot", "vel", "heatmap"):
        loss = loss + 0.25 * res[key].float().abs().mean()
    if torch.is_tensor(depth_loss):
        loss = loss + 0.0 * depth_loss  # the reference computes but does not add it (BF/bevfusion.py:388-392)
    return loss
