"""CPU restatement of the TransFusion head's target assignment and losses.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  numpy (+ scipy, which is what the reference itself
calls for the Hungarian step, BF/utils.py:6,270).  Each function cites the reference lines it follows
(BF = projects/BEVFusion/bevfusion, M3D = mmdet3d).

Pinning:
  * gaussian_radius / gaussian_2d / draw_heatmap_gaussian and TransFusionBBoxCoder.encode/decode,
    BBoxBEVL1Cost: checked against the reference's own code (tests/golden/head_ref.npz, made by
    tests/golden/make_golden.py from M3D/models/utils/gaussian.py and BF/utils.py).
  * Hungarian assignment: scipy.optimize.linear_sum_assignment IS the reference implementation.
  * rotated BEV IoU (`mmcv.ops.box_iou_rotated`), `mmdet.FocalLossCost`, `mmdet.FocalLoss`,
    `mmdet.GaussianFocalLoss`, `mmdet.L1Loss`: the arithmetic lives in mmcv / mmdet, which are not in
    /root/reference and not installed -> restated from their published formulas; PARITY UNPINNED for these
    (the IoU is cross-checked against exact cases and a Monte-Carlo estimate in tests/test_head_oracle.py;
    FocalLossCost also exists in-tree as projects/PETR/petr/match_cost.py:210-260, same formula).
"""
import numpy as np

try:
    from scipy.optimize import linear_sum_assignment
except ImportError:  # pragma: no cover
    linear_sum_assignment = None


# ------------------------------------------------------------------------------ rotated IoU
def _corners(b):
    """b = (xc, yc, w, h, angle) -> 4 corners, counter-clockwise."""
    xc, yc, w, h, a = [float(v) for v in b]
    c, s = np.cos(a), np.sin(a)
    ux, uy = c * w / 2, s * w / 2
    vx, vy = -s * h / 2, c * h / 2
    return np.array([[xc - ux - vx, yc - uy - vy], [xc + ux - vx, yc + uy - vy],
                     [xc + ux + vx, yc + uy + vy], [xc - ux + vx, yc - uy + vy]], dtype=np.float64)


def _clip(subject, a, b):
    """Sutherland-Hodgman: keep the part of polygon `subject` left of the directed edge a->b."""
    out = []
    n = len(subject)
    for i in range(n):
        p, q = subject[i], subject[(i + 1) % n]
        sp = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        sq = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
        if sp >= 0:
            out.append(p)
        if (sp >= 0) != (sq >= 0):
            t = sp / (sp - sq)
            out.append(p + t * (q - p))
    return out


def rotated_intersection_area(b1, b2):
    poly = list(_corners(b1))
    clipper = _corners(b2)
    for i in range(4):
        if len(poly) < 3:
            return 0.0
        poly = _clip(poly, clipper[i], clipper[(i + 1) % 4])
    if len(poly) < 3:
        return 0.0
    p = np.array(poly)
    x, y = p[:, 0], p[:, 1]
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def box_iou_rotated(b1, b2):
    """IoU of rotated rectangles (x, y, w, h, angle) -> [N, M]   (mmcv.ops.box_iou_rotated, mode='iou')."""
    b1, b2 = np.asarray(b1, np.float64), np.asarray(b2, np.float64)
    out = np.zeros((len(b1), len(b2)))
    for i in range(len(b1)):
        for j in range(len(b2)):
            inter = rotated_intersection_area(b1[i], b2[j])
            a1, a2 = b1[i, 2] * b1[i, 3], b2[j, 2] * b2[j, 3]
            out[i, j] = inter / (a1 + a2 - inter) if inter > 0 else 0.0
    return out


def bbox_overlaps_3d_lidar(boxes1, boxes2):
    """M3D/structures/bbox_3d/base_box3d.py:529-590 with LiDAR boxes (x, y, z_bottom, dx, dy, dz, yaw):
    height overlap x BEV overlap, IoU over volumes (M3D/structures/ops/iou3d_calculator.py:148-176)."""
    b1, b2 = np.asarray(boxes1, np.float64), np.asarray(boxes2, np.float64)
    rows, cols = len(b1), len(b2)
    if rows * cols == 0:
        return np.zeros((rows, cols))
    top1, bot1 = (b1[:, 2] + b1[:, 5])[:, None], b1[:, 2][:, None]
    top2, bot2 = (b2[:, 2] + b2[:, 5])[None], b2[:, 2][None]
    overlaps_h = np.clip(np.minimum(top1, top2) - np.maximum(bot1, bot2), 0, None)        # :517-526
    bev1, bev2 = b1[:, [0, 1, 3, 4, 6]].copy(), b2[:, [0, 1, 3, 4, 6]].copy()
    bev1[:, 2:4] = np.clip(bev1[:, 2:4], 1e-4, None)                                        # :566-567
    bev2[:, 2:4] = np.clip(bev2[:, 2:4], 1e-4, None)
    iou2d = box_iou_rotated(bev1, bev2)
    areas1 = (bev1[:, 2] * bev1[:, 3])[:, None]
    areas2 = (bev2[:, 2] * bev2[:, 3])[None]
    overlaps_bev = iou2d * (areas1 + areas2) / (1 + iou2d)                                  # :575
    overlaps_3d = overlaps_bev * overlaps_h
    vol1 = (b1[:, 3] * b1[:, 4] * b1[:, 5])[:, None]
    vol2 = (b2[:, 3] * b2[:, 4] * b2[:, 5])[None]
    return overlaps_3d / np.clip(vol1 + vol2 - overlaps_3d, 1e-8, None)                    # :585-586


# ------------------------------------------------------------------------------ matching costs
def focal_loss_cost(cls_logits, gt_labels, alpha=0.25, gamma=2.0, weight=0.15, eps=1e-12):
    """mmdet FocalLossCost (same formula in-tree: projects/PETR/petr/match_cost.py:244-260).
    cls_logits [P, C] -> [P, G]."""
    p = 1.0 / (1.0 + np.exp(-np.asarray(cls_logits, np.float64)))
    neg = -np.log(1 - p + eps) * (1 - alpha) * p ** gamma
    pos = -np.log(p + eps) * alpha * (1 - p) ** gamma
    return (pos[:, gt_labels] - neg[:, gt_labels]) * weight


def bev_l1_cost(bboxes, gt_bboxes, point_cloud_range, weight=0.25):
    """BF/utils.py:133-140."""
    pc = np.asarray(point_cloud_range, np.float64)
    start, rng = pc[0:2], pc[3:5] - pc[0:2]
    a = (np.asarray(bboxes, np.float64)[:, :2] - start) / rng
    b = (np.asarray(gt_bboxes, np.float64)[:, :2] - start) / rng
    return np.abs(a[:, None, :] - b[None, :, :]).sum(-1) * weight


def hungarian_assign(bboxes, gt_bboxes, gt_labels, cls_logits, point_cloud_range,
                     cls_w=0.15, alpha=0.25, gamma=2.0, reg_w=0.25, iou_w=0.25, cost_override=None):
    """BF/utils.py:241-284.  bboxes [P, >=7] decoded predictions, cls_logits [C, P] (the head's `heatmap` output of
    one sample).  Returns (assigned_gt_inds [P] (0 = background, g+1 = matched), max_overlaps [P], labels [P],
    cost [P, G], iou [P, G])."""
    P, G = len(bboxes), len(gt_bboxes)
    assigned = np.full(P, -1, np.int64)
    labels = np.full(P, -1, np.int64)
    if G == 0 or P == 0:                                                                    # :247-252
        if G == 0:
            assigned[:] = 0
        return assigned, np.zeros(P), labels, np.zeros((P, G)), np.zeros((P, G))
    gt_labels = np.asarray(gt_labels, np.int64)
    cls_cost = focal_loss_cost(np.asarray(cls_logits).T, gt_labels, alpha, gamma, cls_w)    # :257-258
    reg_cost = bev_l1_cost(bboxes, gt_bboxes, point_cloud_range, reg_w)                      # :259
    iou = bbox_overlaps_3d_lidar(np.asarray(bboxes)[:, :7], np.asarray(gt_bboxes)[:, :7])    # :260
    cost = cls_cost + reg_cost + (-iou * iou_w)                                              # :261-264
    if cost_override is not None:
        cost = np.asarray(cost_override, np.float64)
    rows, cols = linear_sum_assignment(cost)                                                 # :270
    assigned[:] = 0
    assigned[rows] = cols + 1                                                                # :276-279
    labels[rows] = gt_labels[cols]
    max_overlaps = np.zeros(P)
    max_overlaps[rows] = iou[rows, cols]                                                     # :281-282
    return assigned, max_overlaps, labels, cost, iou


# ------------------------------------------------------------------------------ box coder
def bbox_encode(dst_boxes, pc_range, out_size_factor, voxel_size, code_size=10):
    """BF/utils.py:33-46 (fp32 as the reference computes it)."""
    b = np.asarray(dst_boxes, np.float32)
    t = np.zeros((len(b), code_size), np.float32)
    t[:, 0] = (b[:, 0] - np.float32(pc_range[0])) / np.float32(out_size_factor * voxel_size[0])
    t[:, 1] = (b[:, 1] - np.float32(pc_range[1])) / np.float32(out_size_factor * voxel_size[1])
    t[:, 3:6] = np.log(b[:, 3:6])
    t[:, 2] = b[:, 2] + b[:, 5] * np.float32(0.5)
    t[:, 6] = np.sin(b[:, 6])
    t[:, 7] = np.cos(b[:, 6])
    if code_size == 10:
        t[:, 8:10] = b[:, 7:9]
    return t


def bbox_decode(center, height, dim, rot, vel, pc_range, out_size_factor, voxel_size):
    """BF/utils.py:72-85 for ONE sample: inputs [ch, P] -> boxes [P, 7 | 9] (x, y, z_bottom, dx, dy, dz, yaw, vx, vy)."""
    center, height, dim, rot = [np.asarray(a, np.float32) for a in (center, height, dim, rot)]
    x = center[0] * np.float32(out_size_factor) * np.float32(voxel_size[0]) + np.float32(pc_range[0])
    y = center[1] * np.float32(out_size_factor) * np.float32(voxel_size[1]) + np.float32(pc_range[1])
    d = np.exp(dim)
    z = height[0] - d[2] * np.float32(0.5)
    yaw = np.arctan2(rot[0], rot[1])
    cols = [x, y, z, d[0], d[1], d[2], yaw]
    if vel is not None:
        vel = np.asarray(vel, np.float32)
        cols += [vel[0], vel[1]]
    return np.stack(cols, 1).astype(np.float32)


# ------------------------------------------------------------------------------ heat-map targets
def gaussian_radius(height, width, min_overlap):
    """M3D/models/utils/gaussian.py:62-92 in fp32 (the reference evaluates it on 0-dim fp32 tensors)."""
    f = np.float32
    height, width, mo = f(height), f(width), min_overlap
    b1 = height + width
    c1 = width * height * f(1 - mo) / f(1 + mo)
    sq1 = np.sqrt(b1 * b1 - f(4 * 1) * c1)
    r1 = (b1 + sq1) / f(2)
    b2 = f(2) * (height + width)
    c2 = f(1 - mo) * width * height
    sq2 = np.sqrt(b2 * b2 - f(4 * 4) * c2)
    r2 = (b2 + sq2) / f(2)
    a3 = 4 * mo
    b3 = f(-2 * mo) * (height + width)
    c3 = f(mo - 1) * width * height
    sq3 = np.sqrt(b3 * b3 - f(4 * a3) * c3)
    r3 = (b3 + sq3) / f(2)
    return min(r1, r2, r3)


def gaussian_2d(diameter, sigma):
    """gaussian.py:9-25."""
    m = (diameter - 1.0) / 2.0
    y, x = np.ogrid[-m:m + 1, -m:m + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_heatmap_gaussian(heatmap, center, radius, k=1):
    """gaussian.py:28-59; heatmap [H, W] float32 modified in place; center = (x, y)."""
    diameter = 2 * radius + 1
    gaussian = gaussian_2d(diameter, sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    masked_heatmap = heatmap[y - top:y + bottom, x - left:x + right]
    masked_gaussian = gaussian[radius - top:radius + bottom, radius - left:radius + right].astype(np.float32)
    if min(masked_gaussian.shape) > 0 and min(masked_heatmap.shape) > 0:
        np.maximum(masked_heatmap, masked_gaussian * k, out=masked_heatmap)
    return heatmap


def heatmap_targets(gt_boxes, gt_labels, num_classes, grid_size, pc_range, voxel_size, out_size_factor,
                    gaussian_overlap=0.1, min_radius=2):
    """BF/bevfusion_head.py:636-662.  gt_boxes [G, >=5] (x, y, z, dx, dy, ...).  Returns [num_classes, Y', X'] where
    the reference's `center_int[[1, 0]]` fix puts box (cx, cy) at heatmap[cls][cx_cell][cy_cell]."""
    f = np.float32
    fx, fy = grid_size[0] // out_size_factor, grid_size[1] // out_size_factor
    heat = np.zeros((num_classes, fy, fx), np.float32)
    for idx in range(len(gt_boxes)):
        width = f(gt_boxes[idx][3]) / f(voxel_size[0]) / f(out_size_factor)
        length = f(gt_boxes[idx][4]) / f(voxel_size[1]) / f(out_size_factor)
        if width > 0 and length > 0:
            radius = gaussian_radius(length, width, gaussian_overlap)
            radius = max(min_radius, int(radius))
            coor_x = (f(gt_boxes[idx][0]) - f(pc_range[0])) / f(voxel_size[0]) / f(out_size_factor)
            coor_y = (f(gt_boxes[idx][1]) - f(pc_range[1])) / f(voxel_size[1]) / f(out_size_factor)
            cx, cy = int(coor_x), int(coor_y)          # .to(torch.int32): truncation
            draw_heatmap_gaussian(heat[int(gt_labels[idx])], (cy, cx), radius)
    return heat


# ------------------------------------------------------------------------------ targets of one sample
def get_targets_single(gt_boxes, gt_labels, boxes_pred, cls_logits, cfg, cost_override=None):
    """BF/bevfusion_head.py:514-674 for one sample and one decoder layer.
    gt_boxes [G, 9] bottom-centre LiDAR boxes (+ velocity), boxes_pred [P, 9] decoded, cls_logits [C, P]."""
    P = len(boxes_pred)
    a = cfg["assigner"]
    assigned, max_overlaps, _, cost, iou = hungarian_assign(
        boxes_pred, gt_boxes, gt_labels, cls_logits, cfg["point_cloud_range"], a["cls_w"], a["alpha"], a["gamma"],
        a["reg_w"], a["iou_w"], cost_override)
    code = cfg["code_size"]
    bbox_targets = np.zeros((P, code), np.float32)
    bbox_weights = np.zeros((P, code), np.float32)
    ious = np.clip(max_overlaps, 0.0, 1.0)                                                   # :607-608
    labels = np.full(P, cfg["num_classes"], np.int64)                                        # :609-613
    label_weights = np.zeros(P, np.int64)
    pos = np.nonzero(assigned > 0)[0]
    neg = np.nonzero(assigned == 0)[0]
    if len(pos):
        gsel = assigned[pos] - 1
        bbox_targets[pos] = bbox_encode(np.asarray(gt_boxes)[gsel], cfg["point_cloud_range"], cfg["out_size_factor"],
                                        cfg["voxel_size"], code)                            # :618-621
        bbox_weights[pos] = 1.0
        labels[pos] = np.asarray(gt_labels)[gsel]
        label_weights[pos] = 1 if cfg.get("pos_weight", -1) <= 0 else cfg["pos_weight"]     # :627-630
    label_weights[neg] = 1
    gt = np.asarray(gt_boxes, np.float32)
    heat = heatmap_targets(gt, gt_labels, cfg["num_classes"], cfg["grid_size"], cfg["point_cloud_range"],
                           cfg["voxel_size"], cfg["out_size_factor"], cfg["gaussian_overlap"], cfg["min_radius"])
    mean_iou = ious[pos].sum() / max(len(pos), 1)
    return dict(labels=labels, label_weights=label_weights, bbox_targets=bbox_targets, bbox_weights=bbox_weights,
                ious=ious, num_pos=len(pos), matched_iou=float(mean_iou), heatmap=heat, assigned=assigned,
                cost=cost, iou=iou)


# ------------------------------------------------------------------------------ losses (fp64)
def clip_sigmoid(x, eps=1e-4):
    """BF/bevfusion_head.py:20-23."""
    return np.clip(1.0 / (1.0 + np.exp(-np.asarray(x, np.float64))), eps, 1 - eps)


def gaussian_focal_loss(pred, target, alpha=2.0, gamma=4.0, avg_factor=1.0):
    """mmdet GaussianFocalLoss (reduction='mean' with avg_factor -> sum / avg_factor); pred already in (0, 1)."""
    eps = 1e-12
    pred, target = np.asarray(pred, np.float64), np.asarray(target, np.float64)
    pos_w = (target == 1).astype(np.float64)
    neg_w = (1 - target) ** gamma
    pos = -np.log(pred + eps) * (1 - pred) ** alpha * pos_w
    neg = -np.log(1 - pred + eps) * pred ** alpha * neg_w
    return (pos + neg).sum() / avg_factor


def sigmoid_focal_loss(logits, labels, weights, gamma=2.0, alpha=0.25, avg_factor=1.0):
    """mmdet FocalLoss(use_sigmoid=True): logits [R, C], labels [R] (C = background), weights [R]."""
    x = np.asarray(logits, np.float64)
    p = 1.0 / (1.0 + np.exp(-x))
    t = np.zeros_like(x)
    lab = np.asarray(labels)
    fg = lab < x.shape[1]
    t[np.nonzero(fg)[0], lab[fg]] = 1.0
    pt = (1 - p) * t + p * (1 - t)
    fw = (alpha * t + (1 - alpha) * (1 - t)) * pt ** gamma
    bce = np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x)))
    return (bce * fw * np.asarray(weights, np.float64)[:, None]).sum() / avg_factor


def l1_loss(pred, target, weight, avg_factor=1.0):
    """mmdet L1Loss(reduction='mean') with avg_factor."""
    return (np.abs(np.asarray(pred, np.float64) - np.asarray(target, np.float64)) * np.asarray(weight, np.float64)).sum() / avg_factor


# ------------------------------------------------------------------------------ inference post-processing
def circle_nms(dets, thresh, post_max_size=83):
    """mmdet3d/models/layers/box3d_nms.py:186-228 (numpy in the reference too; the squared distance is compared with
    `thresh` as given).  Ties in the score order resolve to the lower index first."""
    dets = np.asarray(dets, np.float32)
    x1, y1, scores = dets[:, 0], dets[:, 1], dets[:, 2]
    order = np.lexsort((np.arange(len(scores)), -scores)).astype(np.int32)
    n = len(dets)
    suppressed = np.zeros(n, np.int32)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(int(i))
        for _j in range(_i + 1, n):
            j = order[_j]
            if suppressed[j]:
                continue
            dist = (x1[i] - x1[j]) ** 2 + (y1[i] - y1[j]) ** 2
            if dist <= np.float32(thresh):
                suppressed[j] = 1
    return keep[:post_max_size] if post_max_size < len(keep) else keep


def nms_rotated(boxes_xywhr, scores, thresh, pre_max_size=None, post_max_size=None):
    """nms_bev (mmdet3d/models/layers/box3d_nms.py:234-275) around mmcv.ops.nms_rotated (third-party; its published
    algorithm: descending score order, box j dropped when an earlier kept box has IoU(i, j) > thresh, IoU on the exact
    intersection polygon of the rotated rectangles, 0 when either area < 1e-14).  Equal scores: lower index first.
    Returns (kept indices into the input, IoU matrix of the sorted boxes)."""
    b = np.asarray(boxes_xywhr, np.float64).reshape(-1, 5)
    sc = np.asarray(scores, np.float32)
    order = np.lexsort((np.arange(len(sc)), -sc))
    if pre_max_size is not None:
        order = order[:pre_max_size]
    bs = b[order]
    m = len(bs)
    iou = np.zeros((m, m))
    for i in range(m):
        for j in range(i + 1, m):
            a1, a2 = bs[i, 2] * bs[i, 3], bs[j, 2] * bs[j, 3]
            if a1 < 1e-14 or a2 < 1e-14:
                continue
            inter = rotated_intersection_area(bs[i], bs[j])
            iou[i, j] = inter / (a1 + a2 - inter)
    removed = np.zeros(m, bool)
    keep = []
    for i in range(m):
        if removed[i]:
            continue
        keep.append(int(order[i]))
        removed |= iou[i] > thresh
    if post_max_size is not None:
        keep = keep[:post_max_size]
    return keep, iou


def heuristic_assign(bboxes, gt_bboxes, gt_labels=None, query_labels=None, dist_thre=100.0):
    """BF/utils.py:161-223 (HeuristicAssigner3D.assign) without the IoU of the matched pairs.
    Returns assigned_gt_inds [P] (0 = background, g+1), assigned labels [P] (-1 = none)."""
    b, g = np.asarray(bboxes, np.float32), np.asarray(gt_bboxes, np.float32)
    P, G = len(b), len(g)
    dist = np.linalg.norm(b[None, :, :2] - g[:, None, :2], axis=-1).astype(np.float32)  # [G, P]
    if query_labels is not None:
        dist = dist + (np.asarray(query_labels)[None] != np.asarray(gt_labels)[:, None]) * np.float32(dist_thre)
    inds = np.zeros(P, np.int64)
    vals = np.full(P, 10000.0, np.float32)
    labs = np.full(P, -1, np.int64)
    for i in range(G):
        p = int(dist[i].argmin())
        if dist[i, p] <= dist_thre and dist[i, p] < vals[p]:
            vals[p] = dist[i, p]
            inds[p] = i + 1
            labs[p] = int(gt_labels[i]) if gt_labels is not None else -1
    return inds, labs
