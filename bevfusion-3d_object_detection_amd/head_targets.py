"""Host mirror of the TransFusion head's box coder, Hungarian assigner, target builder and losses
(BF/utils.py:14-151,226-284; BF/bevfusion_head.py:450-796; mmdet3d/models/utils/gaussian.py) over csrc/head.hip.

The reference works sample by sample with CPU round trips (`cost.cpu()` + scipy, numpy gaussians, `.item()`); the
batched functions here (`assign_batch`, `build_targets`, `head_losses`) run the whole batch on the device without a
host read.  The per-sample classes keep the reference's names and call signatures on top of them.
"""
from typing import List, Optional, Sequence

import torch

from . import _lib
from .registry import MODELS


def _cfg_get(cfg, key, default=None):
    if cfg is None:
        return default
    return cfg.get(key, default) if hasattr(cfg, "get") else getattr(cfg, key, default)


def _f32c(t):
    return t.detach().float().contiguous()


class AssignResult:
    """mmdet AssignResult: gt_inds 0 = background, k = matched to GT k-1."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


# ----------------------------------------------------------------------------------------------- ground truth
def unpack_gt(inst):
    """(boxes [G, >=7] bottom-centre LiDAR boxes, labels [G]) from an InstanceData-like object
    (`.bboxes_3d.tensor`, `.labels_3d`; BF/bevfusion_head.py:532-533), a dict, or a (boxes, labels) pair."""
    if isinstance(inst, (tuple, list)):
        boxes, labels = inst
    elif isinstance(inst, dict):
        boxes, labels = inst["bboxes_3d"], inst["labels_3d"]
    else:
        boxes, labels = inst.bboxes_3d, inst.labels_3d
    boxes = getattr(boxes, "tensor", boxes)
    return torch.as_tensor(boxes), torch.as_tensor(labels)


def pack_gt(batch_gt, device):
    """Pad the per-sample ground truth to [B, G, W] / [B, G] / [B]; the sizes come from tensor shapes (host),
    so nothing is read back from the device."""
    gts = [unpack_gt(g) for g in batch_gt]
    B = len(gts)
    G = max(1, max(int(b.shape[0]) for b, _ in gts))
    W = max([int(b.shape[1]) for b, _ in gts if b.dim() == 2 and b.shape[0]] or [9])
    counts = [int(b.shape[0]) for b, _ in gts]
    on_host = all(not b.is_cuda for b, _ in gts)
    stage = torch.device("cpu") if on_host else device  # host GT: pad on the host, then ONE copy per array
    boxes = torch.zeros(B, G, W, dtype=torch.float32, device=stage)
    labels = torch.zeros(B, G, dtype=torch.int32, device=stage)
    for i, (b, lab) in enumerate(gts):
        if counts[i]:
            boxes[i, :counts[i], :b.shape[1]] = b.to(device=stage, dtype=torch.float32)
            labels[i, :counts[i]] = lab.to(device=stage, dtype=torch.int32)
    if on_host:
        boxes, labels = boxes.to(device, non_blocking=True), labels.to(device, non_blocking=True)
    n_gt = torch.tensor(counts, dtype=torch.int32).to(device, non_blocking=True)
    return boxes, labels, n_gt, counts


class PackedGT:
    """Ground truth of a batch already padded and resident on the device (what `pack_gt` builds), for callers that prepare
    it ahead of the step -- a data loader's prefetch thread, or a captured hipGraph whose replay must not contain host -> device
    copies.  `BEVFusionHead.loss` / `get_targets` accept it in place of the per-sample list."""

    def __init__(self, batch_gt, device):
        self.boxes, self.labels, self.n_gt, self.counts = pack_gt(batch_gt, device)

    def denom(self, num_proposals, num_layers):
        """max(#positives, 1) per sample, computed on the device from n_gt (no host -> device copy inside the step)."""
        return (torch.clamp(self.n_gt, max=num_proposals) * num_layers).clamp(min=1).float()


# ----------------------------------------------------------------------------------------------- box coder
@MODELS.register_module()
class TransFusionBBoxCoder:
    """BF/utils.py:14-124."""

    def __init__(self, pc_range, out_size_factor, voxel_size, post_center_range=None, score_threshold=None,
                 code_size=8):
        self.pc_range, self.out_size_factor, self.voxel_size = pc_range, out_size_factor, voxel_size
        self.post_center_range, self.score_threshold, self.code_size = post_center_range, score_threshold, code_size

    def encode(self, dst_boxes):
        """:33-46 (torch ops; the training path encodes inside bfhip_assign_targets)."""
        t = torch.zeros([dst_boxes.shape[0], self.code_size], device=dst_boxes.device)
        t[:, 0] = (dst_boxes[:, 0] - self.pc_range[0]) / (self.out_size_factor * self.voxel_size[0])
        t[:, 1] = (dst_boxes[:, 1] - self.pc_range[1]) / (self.out_size_factor * self.voxel_size[1])
        t[:, 3:6] = dst_boxes[:, 3:6].log()
        t[:, 2] = dst_boxes[:, 2] + dst_boxes[:, 5] * 0.5
        t[:, 6] = torch.sin(dst_boxes[:, 6])
        t[:, 7] = torch.cos(dst_boxes[:, 6])
        if self.code_size == 10:
            t[:, 8:10] = dst_boxes[:, 7:]
        return t

    def decode_boxes(self, rot, dim, center, height, vel=None, p_off=0, num=None):
        """Decoded boxes f32[B, P, 7|9] of the proposals [p_off, p_off + num) (one launch, inputs untouched)."""
        B, ld = center.shape[0], center.shape[-1]
        P = ld - p_off if num is None else num
        boxes = torch.empty(B, P, 9 if vel is not None else 7, dtype=torch.float32, device=center.device)
        cfg = _lib.host_f32([self.out_size_factor, self.voxel_size[0], self.voxel_size[1], self.pc_range[0],
                             self.pc_range[1]])
        c, h, d, r = _f32c(center), _f32c(height), _f32c(dim), _f32c(rot)
        v = _f32c(vel) if vel is not None else None
        _lib.call("bfhip_decode_boxes", _lib.ptr(c), _lib.ptr(h), _lib.ptr(d), _lib.ptr(r), _lib.ptr(v), B, P, ld, p_off,
                  cfg, _lib.ptr(boxes), _lib.stream_of(boxes))
        return boxes

    def decode(self, heatmap, rot, dim, center, height, vel, filter=False):
        """:48-124.  Returns a list (one dict per sample) of bboxes / scores / labels."""
        scores, labels = heatmap.max(1)
        boxes = self.decode_boxes(rot, dim, center, height, vel)
        preds = [dict(bboxes=boxes[i], scores=scores[i], labels=labels[i]) for i in range(heatmap.shape[0])]
        if not filter:
            return preds
        if self.post_center_range is None:
            raise NotImplementedError("Need to reorganize output as a batch, only support post_center_range is not None for now!")
        rng = torch.as_tensor(self.post_center_range, device=heatmap.device, dtype=boxes.dtype)
        mask = (boxes[..., :3] >= rng[:3]).all(2) & (boxes[..., :3] <= rng[3:]).all(2)
        if self.score_threshold is not None:
            thresh = scores > self.score_threshold
        out = []
        for i in range(heatmap.shape[0]):
            cmask = mask[i]
            if self.score_threshold:
                cmask = cmask & thresh[i]
            out.append(dict(bboxes=boxes[i, cmask], scores=scores[i, cmask], labels=labels[i, cmask]))
        return out


# ----------------------------------------------------------------------------------------------- assignment
def _assigner_weights(assigner_cfg):
    cls = _cfg_get(assigner_cfg, "cls_cost", {}) or {}
    reg = _cfg_get(assigner_cfg, "reg_cost", {}) or {}
    iou = _cfg_get(assigner_cfg, "iou_cost", {}) or {}
    return dict(cls_w=float(_cfg_get(cls, "weight", 1.0)), alpha=float(_cfg_get(cls, "alpha", 0.25)),
                gamma=float(_cfg_get(cls, "gamma", 2.0)), eps=float(_cfg_get(cls, "eps", 1e-12)),
                reg_w=float(_cfg_get(reg, "weight", 1.0)), iou_w=float(_cfg_get(iou, "weight", 1.0)))


def assign_batch(boxes, cls_logits, gt_boxes, gt_labels, n_gt, point_cloud_range, weights, p_off=0):
    """Costs + Hungarian matching for the whole batch.
    boxes f32[B,P,W] decoded, cls_logits f32[B,C,ld], gt_* padded.  -> assigned i32[B,P], iou, cost f32[B,P,G], status."""
    B, P, W = boxes.shape
    G, Wg = gt_boxes.shape[1], gt_boxes.shape[2]
    C, ld = cls_logits.shape[1], cls_logits.shape[2]
    dev = boxes.device
    cost = torch.empty(B, P, G, dtype=torch.float32, device=dev)
    iou = torch.empty_like(cost)
    assigned = torch.empty(B, P, dtype=torch.int32, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    pc = point_cloud_range
    cfg = _lib.host_f32([weights["cls_w"], weights["alpha"], weights["gamma"], weights["eps"], weights["reg_w"],
                         weights["iou_w"], pc[0], pc[1], pc[3], pc[4]])
    logits = _f32c(cls_logits)
    s = _lib.stream_of(boxes)
    _lib.call("bfhip_assign_cost", _lib.ptr(boxes), W, _lib.ptr(logits), C, ld, p_off, _lib.ptr(gt_boxes), Wg,
              _lib.ptr(gt_labels), _lib.ptr(n_gt), B, P, G, cfg, _lib.ptr(cost), _lib.ptr(iou), s)
    _lib.call("bfhip_hungarian", _lib.ptr(cost), _lib.ptr(n_gt), B, P, G, _lib.ptr(assigned), _lib.ptr(status), s)
    return assigned, iou, cost, status


def hungarian(cost, n_gt):
    """Minimum-cost assignment of a padded cost batch f32[B,P,G] (device) -> assigned i32[B,P], status i32[B]."""
    B, P, G = cost.shape
    cost = _f32c(cost)
    assigned = torch.empty(B, P, dtype=torch.int32, device=cost.device)
    status = torch.empty(B, dtype=torch.int32, device=cost.device)
    _lib.call("bfhip_hungarian", _lib.ptr(cost), _lib.ptr(n_gt), B, P, G, _lib.ptr(assigned), _lib.ptr(status),
              _lib.stream_of(cost))
    return assigned, status


@MODELS.register_module()
class HungarianAssigner3D:
    """BF/utils.py:226-284, one sample per call like the reference (costs: mmdet FocalLossCost, BBoxBEVL1Cost,
    IoU3DCost with BboxOverlaps3D(coordinate='lidar'))."""

    def __init__(self, cls_cost=dict(type="ClassificationCost", weight=1.0), reg_cost=dict(type="BBoxBEVL1Cost", weight=1.0),
                 iou_cost=dict(type="IoU3DCost", weight=1.0), iou_calculator=dict(type="BboxOverlaps3D")):
        assert _cfg_get(cls_cost, "type", "").endswith("FocalLossCost"), "only the focal classification cost is implemented"
        self.weights = _assigner_weights(dict(cls_cost=cls_cost, reg_cost=reg_cost, iou_cost=iou_cost))

    def assign(self, bboxes, gt_bboxes, gt_labels, cls_pred, train_cfg):
        """bboxes [P, >=7], gt_bboxes [G, >=7], gt_labels [G], cls_pred [1, C, P] logits."""
        P, G = bboxes.size(0), gt_bboxes.size(0)
        gt_inds = bboxes.new_full((P,), -1, dtype=torch.long)
        labels = bboxes.new_full((P,), -1, dtype=torch.long)
        if G == 0 or P == 0:
            if G == 0:
                gt_inds[:] = 0
            return AssignResult(G, gt_inds, None, labels=labels)
        dev = bboxes.device
        gtb = gt_bboxes.to(dev, torch.float32).contiguous()[None]
        gtl = gt_labels.to(dev, torch.int32).contiguous()[None]
        n_gt = torch.tensor([G], dtype=torch.int32).to(dev)
        assigned, iou, _, _ = assign_batch(_f32c(bboxes)[None], cls_pred, gtb, gtl, n_gt,
                                           _cfg_get(train_cfg, "point_cloud_range"), self.weights)
        gt_inds = assigned[0].long()
        pos = gt_inds > 0
        sel = (gt_inds - 1).clamp(min=0)
        labels = torch.where(pos, gt_labels.to(dev).long()[sel], labels)
        max_overlaps = torch.where(pos, iou[0].gather(1, sel[:, None])[:, 0], torch.zeros_like(iou[0, :, 0]))
        return AssignResult(G, gt_inds, max_overlaps, labels=labels)


@MODELS.register_module()
class HeuristicAssigner3D:
    """BF/utils.py:154-223: every GT box takes its nearest prediction in BEV (same class only when query labels are given);
    a prediction claimed by several GT boxes keeps the closest one (the lower GT index on ties).  Vectorised on the device:
    the reference walks the GT boxes in a Python loop with a device read per step."""

    def __init__(self, dist_thre=100, iou_calculator=dict(type="BboxOverlaps3D")):
        self.dist_thre = dist_thre

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None, query_labels=None):
        P, G = bboxes.size(0), gt_bboxes.size(0)
        dev = bboxes.device
        inds = torch.zeros(P, dtype=torch.long, device=dev)
        labels = torch.full((P,), -1, dtype=torch.long, device=dev)
        overlaps = torch.zeros(P, dtype=torch.float32, device=dev)
        if G == 0 or P == 0:
            return AssignResult(G, inds, overlaps, labels=labels)
        gt_bboxes = gt_bboxes.to(dev, torch.float32)
        dist = torch.norm(bboxes[:, 0:2][None, :, :] - gt_bboxes[:, 0:2][:, None, :], dim=-1)  # [G, P]
        if query_labels is not None:
            dist = dist + (query_labels[None] != gt_labels[:, None]).to(dist.dtype) * self.dist_thre
        near_val, near = dist.min(1)                                   # per GT: its nearest prediction
        ok = near_val <= self.dist_thre
        big = torch.full((P,), 10000.0, dtype=dist.dtype, device=dev)
        best = big.scatter_reduce(0, near[ok], near_val[ok], reduce="amin", include_self=True)
        wins = ok & (near_val == best[near]) & (near_val < 10000.0)    # candidates at the winning distance
        g_idx = torch.arange(G, device=dev)
        first = torch.full((P,), G, dtype=torch.long, device=dev).scatter_reduce(0, near[wins], g_idx[wins], reduce="amin",
                                                                                 include_self=True)
        matched = first < G
        inds = torch.where(matched, first + 1, inds)
        if gt_labels is not None:
            labels = torch.where(matched, gt_labels.to(dev).long()[first.clamp(max=G - 1)], labels)
        # IoU of the matched pairs (BF/utils.py:218-221)
        gtl = (gt_labels if gt_labels is not None else torch.zeros(G, device=dev)).to(dev, torch.int32).contiguous()[None]
        n_gt = torch.tensor([G], dtype=torch.int32).to(dev)
        zeros = torch.zeros(1, 1, P, dtype=torch.float32, device=dev)
        w = dict(cls_w=0.0, alpha=0.25, gamma=2.0, eps=1e-12, reg_w=0.0, iou_w=1.0)
        _, iou, _, _ = assign_batch(_f32c(bboxes)[None], zeros, gt_bboxes.contiguous()[None], gtl.clamp(min=0, max=0), n_gt,
                                    [-1e4, -1e4, -1e4, 1e4, 1e4, 1e4], w)
        overlaps = torch.where(matched, iou[0].gather(1, first.clamp(max=G - 1)[:, None])[:, 0], overlaps)
        return AssignResult(G, inds, overlaps, labels=labels)


def circle_nms(dets, thresh, post_max_size=83):
    """mmdet3d/models/layers/box3d_nms.py:186-228 on the device: dets [N, 3] = (x, y, score) -> kept indices (int64, highest
    score first).  One host read (the number of kept boxes), as the reference returns a Python list."""
    n = int(dets.shape[0])
    d = _f32c(dets)
    keep = torch.empty(max(min(n, post_max_size), 1), dtype=torch.int32, device=d.device)
    count = torch.empty(1, dtype=torch.int32, device=d.device)
    _lib.call("bfhip_circle_nms", _lib.ptr(d) if n else None, n, float(thresh), int(post_max_size), _lib.ptr(keep), _lib.ptr(count),
              _lib.stream_of(d))
    return keep[:int(count.item())].long()


def rotate_nms(boxes_xywhr, scores, thresh, pre_max_size=None, post_max_size=None):
    """mmcv.ops.nms_rotated behind nms_bev, on the device: boxes [N, 5] = (x, y, w, h, angle) -> kept indices (int64,
    highest score first).  One host read (the number of kept boxes)."""
    n = int(boxes_xywhr.shape[0])
    b, sc = _f32c(boxes_xywhr), _f32c(scores)
    pre = n if pre_max_size is None else int(pre_max_size)
    post = n if post_max_size is None else int(post_max_size)
    keep = torch.empty(max(min(n, pre, post), 1), dtype=torch.int32, device=b.device)
    count = torch.empty(1, dtype=torch.int32, device=b.device)
    wsb = _lib.call_size("bfhip_rotate_nms_workspace_bytes", n, pre)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=b.device)
    _lib.call("bfhip_rotate_nms", _lib.ptr(b) if n else None, _lib.ptr(sc) if n else None, n, float(thresh), pre, post,
              _lib.ptr(keep), _lib.ptr(count), _lib.ptr(ws), wsb, _lib.stream_of(b))
    return keep[:int(count.item())].long()


def xywhr2xyxyr(boxes_xywhr):
    """mmdet3d/structures/bbox_3d/utils.py:128-147."""
    half_w, half_h = boxes_xywhr[..., 2] / 2, boxes_xywhr[..., 3] / 2
    return torch.stack([boxes_xywhr[..., 0] - half_w, boxes_xywhr[..., 1] - half_h, boxes_xywhr[..., 0] + half_w,
                        boxes_xywhr[..., 1] + half_h, boxes_xywhr[..., 4]], dim=-1)


def nms_bev(boxes, scores, thresh, pre_max_size=None, post_max_size=None):
    """mmdet3d/models/layers/box3d_nms.py:234-275 with its argument convention: boxes [N, 5] = (x1, y1, x2, y2, ry).
    The xyxyr -> xywhr conversion is the reference's fp32 arithmetic; sort, pre/post limits and suppression run in
    bfhip_rotate_nms."""
    assert boxes.shape[1] == 5, "Input boxes shape should be [N, 5]"
    xywhr = torch.stack(((boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2, boxes[:, 2] - boxes[:, 0],
                         boxes[:, 3] - boxes[:, 1], boxes[:, 4]), dim=-1)
    return rotate_nms(xywhr, scores, thresh, pre_max_size, post_max_size)


# ----------------------------------------------------------------------------------------------- targets
def build_targets(assigned, iou, gt_boxes, gt_labels, num_classes, code_size, pc_range, out_size_factor, voxel_size,
                  pos_weight=-1):
    """BF/bevfusion_head.py:604-633 for the batch -> labels i32[B,P], label_weights, bbox_targets, bbox_weights, ious."""
    B, P = assigned.shape
    G, Wg = gt_boxes.shape[1], gt_boxes.shape[2]
    dev = assigned.device
    labels = torch.empty(B, P, dtype=torch.int32, device=dev)
    label_weights = torch.empty(B, P, dtype=torch.float32, device=dev)
    bbox_targets = torch.empty(B, P, code_size, dtype=torch.float32, device=dev)
    bbox_weights = torch.empty_like(bbox_targets)
    ious = torch.empty(B, P, dtype=torch.float32, device=dev)
    cfg = _lib.host_f32([pc_range[0], pc_range[1], out_size_factor * voxel_size[0], out_size_factor * voxel_size[1],
                         pos_weight])
    _lib.call("bfhip_assign_targets", _lib.ptr(assigned), _lib.ptr(iou), _lib.ptr(gt_boxes), Wg, _lib.ptr(gt_labels), B, P,
              G, num_classes, code_size, cfg, _lib.ptr(labels), _lib.ptr(label_weights), _lib.ptr(bbox_targets),
              _lib.ptr(bbox_weights), _lib.ptr(ious), _lib.stream_of(assigned))
    return labels, label_weights, bbox_targets, bbox_weights, ious


def draw_heatmap(gt_boxes, gt_labels, n_gt, num_classes, grid_size, pc_range, voxel_size, out_size_factor,
                 gaussian_overlap=0.1, min_radius=2):
    """Dense heat-map targets f32[B, num_classes, Y', X'] (BF/bevfusion_head.py:636-662)."""
    B, G, Wg = gt_boxes.shape
    fx, fy = int(grid_size[0]) // int(out_size_factor), int(grid_size[1]) // int(out_size_factor)
    heat = torch.empty(B, num_classes, fy, fx, dtype=torch.float32, device=gt_boxes.device)
    cfg = _lib.host_f32([pc_range[0], pc_range[1], voxel_size[0], voxel_size[1], out_size_factor])
    _lib.call("bfhip_draw_heatmap", _lib.ptr(gt_boxes), Wg, _lib.ptr(gt_labels), _lib.ptr(n_gt), B, G, num_classes, fy, fx,
              cfg, float(gaussian_overlap), int(min_radius), _lib.ptr(heat), _lib.stream_of(heat))
    return heat


# ----------------------------------------------------------------------------------------------- losses
class _GaussianFocal(torch.autograd.Function):
    """(sum of GaussianFocalLoss over clip_sigmoid(logits), number of target == 1) in one pass; d/dlogits saved."""

    @staticmethod
    def forward(ctx, logits, target, clip_eps):
        x, t = logits.float().contiguous(), target.float().contiguous()
        n = x.numel()
        out = torch.empty(2, dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        ws_bytes = _lib.call_size("bfhip_gaussian_focal_loss_workspace_bytes", n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        _lib.call("bfhip_gaussian_focal_loss", _lib.ptr(x), _lib.ptr(t), n, float(clip_eps), _lib.ptr(out), _lib.ptr(grad),
                  _lib.ptr(ws), ws_bytes, _lib.stream_of(x))
        ctx.save_for_backward(grad)
        ctx.in_dtype = logits.dtype
        total, npos = out[0].clone(), out[1].clone()
        ctx.mark_non_differentiable(npos)
        return total, npos

    @staticmethod
    def backward(ctx, g_sum, g_cnt):
        (grad,) = ctx.saved_tensors
        return (grad * g_sum).to(ctx.in_dtype), None, None


def gaussian_focal_loss_with_logits(logits, target, clip_eps=1e-4, loss_weight=1.0):
    """loss_heatmap of BF/bevfusion_head.py:714-720: GaussianFocalLoss(clip_sigmoid(x), target, avg_factor =
    max(#(target == 1), 1)), reduction 'mean'.  The averaging factor stays on the device."""
    total, npos = _GaussianFocal.apply(logits, target, clip_eps)
    return total / npos.clamp(min=1.0) * loss_weight


class _QueryLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_logits, box_pred, labels, label_weights, bbox_targets, bbox_weights, code_weights, p_off, P,
                gamma, alpha):
        x, bp = cls_logits.float().contiguous(), box_pred.float().contiguous()
        B, C, ld = x.shape
        K = bp.shape[1]
        out = torch.empty(2, dtype=torch.float32, device=x.device)
        g_cls = torch.zeros_like(x) if ld != P else torch.empty_like(x)
        g_box = torch.zeros_like(bp) if ld != P else torch.empty_like(bp)
        _lib.call("bfhip_query_losses", _lib.ptr(x), _lib.ptr(labels), _lib.ptr(label_weights), _lib.ptr(bp),
                  _lib.ptr(bbox_targets), _lib.ptr(bbox_weights), _lib.ptr(code_weights), B, C, P, K, ld, p_off,
                  float(gamma), float(alpha), _lib.ptr(g_cls), _lib.ptr(g_box), _lib.ptr(out), _lib.stream_of(x))
        ctx.save_for_backward(g_cls, g_box)
        ctx.dtypes = (cls_logits.dtype, box_pred.dtype)
        return out[0].clone(), out[1].clone()

    @staticmethod
    def backward(ctx, g0, g1):
        g_cls, g_box = ctx.saved_tensors
        return ((g_cls * g0).to(ctx.dtypes[0]), (g_box * g1).to(ctx.dtypes[1])) + (None,) * 9


def query_losses(cls_logits, box_pred, labels, label_weights, bbox_targets, bbox_weights, code_weights, p_off, P,
                 gamma=2.0, alpha=0.25):
    """(weighted sigmoid-focal sum, weighted L1 sum) of one decoder layer's proposals (BF/bevfusion_head.py:729-788)."""
    return _QueryLosses.apply(cls_logits, box_pred, labels.contiguous(), label_weights.contiguous(),
                              bbox_targets.contiguous(), bbox_weights.contiguous(), code_weights, p_off, P, gamma, alpha)
