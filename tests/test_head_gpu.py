"""GPU parity of csrc/head.hip (TransFusion head targets + losses) against oracle/head_oracle.py,
scipy's linear_sum_assignment (the reference's own Hungarian step) and the reference outputs in tests/golden/head_ref.npz.
Tolerances: assignment / labels / heat-map bit-exact; fp32 transcendental chains 1e-5; IoU 2e-4 abs (fp32 polygon
clipping vs the fp64 oracle)."""
import os

import numpy as np
import pytest
import torch

import bevfusion_amd  # noqa: F401
from bevfusion_amd import head_targets as ht
from bevfusion_amd import synthetic
from oracle import head_oracle as ho

pytestmark = pytest.mark.gpu
N = synthetic.NUSC
PC, VS, OSF = N["point_cloud_range"], N["voxel_size"], 8
GOLD = os.path.join(os.path.dirname(__file__), "golden", "head_ref.npz")
W = dict(cls_w=0.15, alpha=0.25, gamma=2.0, eps=1e-12, reg_w=0.25, iou_w=0.25)
CFG = dict(point_cloud_range=PC, voxel_size=VS, out_size_factor=OSF, grid_size=[1440, 1440, 41], num_classes=10,
           code_size=10, gaussian_overlap=0.1, min_radius=2, pos_weight=-1,
           assigner=dict(cls_w=0.15, alpha=0.25, gamma=2.0, reg_w=0.25, iou_w=0.25))


def coder():
    return ht.TransFusionBBoxCoder(pc_range=PC[:2], out_size_factor=OSF, voxel_size=VS[:2],
                                   post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.0,
                                   code_size=10)


def scene(seed, n_gt, P=200):
    """GT boxes and P decoded predictions: noisy copies of the GT first (so IoUs are non-trivial), random boxes after."""
    boxes, labels = synthetic.gt_boxes(seed=seed, n=n_gt)
    rs = np.random.RandomState(seed + 7)
    rnd, _ = synthetic.gt_boxes(seed=seed + 1, n=P)
    pred = rnd.copy()
    k = min(n_gt, P)
    perm = rs.permutation(P)[:k]
    pred[perm] = boxes[:k] + rs.normal(0, 1, (k, 9)).astype(np.float32) * np.array([0.4, 0.4, 0.1, 0.2, 0.1, 0.1, 0.15, 0.1, 0.1], np.float32)
    pred[:, 3:6] = np.abs(pred[:, 3:6]) + 0.05
    logits = rs.normal(-2.5, 1.5, (10, P)).astype(np.float32)
    return boxes, labels, pred.astype(np.float32), logits


def test_decode_matches_reference(dev):
    g = np.load(GOLD)
    t = {k: torch.from_numpy(g[k]).to(dev) for k in ("dec_heat", "dec_rot", "dec_dim", "dec_center", "dec_height", "dec_vel")}
    keep = {k: v.clone() for k, v in t.items()}
    out = coder().decode(t["dec_heat"], t["dec_rot"], t["dec_dim"], t["dec_center"], t["dec_height"], t["dec_vel"])
    for b in range(2):
        np.testing.assert_allclose(out[b]["bboxes"].cpu().numpy(), g["dec_boxes"][b], rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(out[b]["labels"].cpu().numpy(), g["dec_labels"][b])
        np.testing.assert_array_equal(out[b]["scores"].cpu().numpy(), g["dec_scores"][b])
    assert all(torch.equal(keep[k], t[k]) for k in t)  # "carefully ! don't change the network outputs" (:536)
    # 7-column variant and a proposal window
    b7 = coder().decode_boxes(t["dec_rot"], t["dec_dim"], t["dec_center"], t["dec_height"], None, p_off=10, num=20)
    np.testing.assert_allclose(b7.cpu().numpy(), g["dec_boxes"][:, 10:30, :7], rtol=1e-5, atol=1e-5)


def test_encode_matches_reference(dev):
    g = np.load(GOLD)
    boxes, _ = synthetic.gt_boxes(seed=3000, n=40)
    enc = coder().encode(torch.from_numpy(boxes).to(dev)).cpu().numpy()
    np.testing.assert_allclose(enc, g["encode"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n_gts", [(40, 0, 17), (1, 230, 60)])
def test_cost_iou_and_assignment(dev, n_gts):
    B, P = len(n_gts), 200
    sc = [scene(4000 + 10 * i, n) for i, n in enumerate(n_gts)]
    gt_boxes, gt_labels, n_gt, counts = ht.pack_gt([(torch.from_numpy(s[0]), torch.from_numpy(s[1])) for s in sc], dev)
    assert counts == list(n_gts)
    pred = torch.from_numpy(np.stack([s[2] for s in sc])).to(dev)
    logits = torch.from_numpy(np.stack([s[3] for s in sc])).to(dev)
    assigned, iou, cost, status = ht.assign_batch(pred, logits, gt_boxes, gt_labels, n_gt, PC, W)
    assert (status.cpu().numpy() == 0).all()
    for b, (boxes, labels, p, lg) in enumerate(sc):
        G = len(boxes)
        a_dev = assigned[b].cpu().numpy()
        if G == 0:
            assert (a_dev == 0).all()
            continue
        a_ref, mo, lab, cost_ref, iou_ref = ho.hungarian_assign(p, boxes, labels, lg, PC, **{k: W[k] for k in ("cls_w", "alpha", "gamma", "reg_w", "iou_w")})
        np.testing.assert_allclose(iou[b, :, :G].cpu().numpy(), iou_ref, atol=2e-4, rtol=0)
        np.testing.assert_allclose(cost[b, :, :G].cpu().numpy(), cost_ref, atol=2e-4, rtol=1e-4)
        assert (iou_ref > 0.3).sum() >= min(G, P) // 4  # the case really exercises overlapping boxes
        # the Hungarian step itself: scipy on the very same (device) cost matrix -> identical matching
        from scipy.optimize import linear_sum_assignment
        c = cost[b, :, :G].cpu().numpy().astype(np.float64)
        rows, cols = linear_sum_assignment(c)
        expect = np.zeros(P, np.int64)
        expect[rows] = cols + 1
        np.testing.assert_array_equal(a_dev, expect)
        assert (a_dev > 0).sum() == min(G, P)
    assert (cost[1, :, n_gts[1]:] == 0).all() and (iou[1, :, n_gts[1]:] == 0).all()  # padding


@pytest.mark.parametrize("P,G", [(200, 37), (200, 200), (64, 300), (5, 1), (1, 9)])
def test_hungarian_random_costs_match_scipy(dev, P, G):
    from scipy.optimize import linear_sum_assignment
    rs = np.random.RandomState(P * 1000 + G)
    B = 4
    cost = rs.uniform(-1, 1, (B, P, G)).astype(np.float32)
    cost[1] = rs.exponential(1.0, (P, G))                      # skewed
    cost[2] = np.round(rs.uniform(0, 6, (P, G)))               # integer costs: many ties
    n = np.array([G, max(1, G // 2), G, 0], np.int32)
    assigned, status = ht.hungarian(torch.from_numpy(cost).to(dev), torch.from_numpy(n).to(dev))
    a = assigned.cpu().numpy()
    assert (status.cpu().numpy() == 0).all()
    for b in range(B):
        g = int(n[b])
        if g == 0:
            assert (a[b] == 0).all()
            continue
        c = cost[b, :, :g].astype(np.float64)
        rows, cols = linear_sum_assignment(c)
        m = a[b] > 0
        assert m.sum() == min(P, g) and len(set(a[b][m])) == m.sum()          # a matching of full size
        total = c[np.nonzero(m)[0], a[b][m] - 1].sum()
        assert total == pytest.approx(c[rows, cols].sum(), rel=1e-12, abs=1e-9)  # optimal
        if b != 2:                                                               # generic costs: unique optimum
            expect = np.zeros(P, np.int64)
            expect[rows] = cols + 1
            np.testing.assert_array_equal(a[b], expect)


def test_hungarian_flags_non_finite_costs(dev):
    cost = torch.zeros(2, 8, 4, device=dev)
    cost[0, :, 1] = float("nan")
    assigned, status = ht.hungarian(cost, torch.tensor([4, 4], dtype=torch.int32, device=dev))
    assert status.cpu().tolist() == [1, 0]
    assert (assigned[0] == 0).all() and (assigned[1] > 0).sum() == 4
    # a single invalid entry is enough (scipy: "matrix contains invalid numeric entries"), +inf alone is a valid (forbidden) pair
    cost = torch.rand(3, 8, 4, device=dev)
    cost[0, 5, 2] = float("nan")
    cost[1, 1, 0] = float("-inf")
    cost[2, 3, 3] = float("inf")
    assigned, status = ht.hungarian(cost, torch.tensor([4, 4, 4], dtype=torch.int32, device=dev))
    assert status.cpu().tolist() == [1, 1, 0]
    assert (assigned[2] > 0).sum() == 4 and assigned[2, 3] != 4


def test_targets_match_oracle(dev):
    n_gts = (33, 0, 210)
    P = 200
    sc = [scene(5000 + 10 * i, n) for i, n in enumerate(n_gts)]
    gt_boxes, gt_labels, n_gt, _ = ht.pack_gt([(torch.from_numpy(s[0]).to(dev), torch.from_numpy(s[1]).to(dev)) for s in sc], dev)
    pred = torch.from_numpy(np.stack([s[2] for s in sc])).to(dev)
    logits = torch.from_numpy(np.stack([s[3] for s in sc])).to(dev)
    assigned, iou, cost, _ = ht.assign_batch(pred, logits, gt_boxes, gt_labels, n_gt, PC, W)
    labels, lw, bt, bw, ious = ht.build_targets(assigned, iou, gt_boxes, gt_labels, 10, 10, PC, OSF, VS, -1)
    heat = ht.draw_heatmap(gt_boxes, gt_labels, n_gt, 10, [1440, 1440, 41], PC, VS, OSF, 0.1, 2)
    for b, (boxes, lab, p, lg) in enumerate(sc):
        G = len(boxes)
        override = cost[b, :, :G].cpu().numpy() if G else None  # same matching problem as the device solved
        ref = ho.get_targets_single(boxes, lab, p, lg, CFG, cost_override=override)
        np.testing.assert_array_equal(assigned[b].cpu().numpy(), ref["assigned"])
        np.testing.assert_array_equal(labels[b].cpu().numpy(), ref["labels"])
        np.testing.assert_array_equal(lw[b].cpu().numpy(), ref["label_weights"].astype(np.float32))
        np.testing.assert_allclose(bt[b].cpu().numpy(), ref["bbox_targets"], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(bw[b].cpu().numpy(), ref["bbox_weights"])
        np.testing.assert_allclose(ious[b].cpu().numpy(), ref["ious"], atol=2e-4)
        np.testing.assert_array_equal(heat[b].cpu().numpy(), ref["heatmap"])


def test_heatmap_matches_reference_bitwise(dev):
    g = np.load(GOLD)
    boxes, labels = synthetic.gt_boxes(seed=3000, n=40)
    gt_boxes, gt_labels, n_gt, _ = ht.pack_gt([(torch.from_numpy(boxes), torch.from_numpy(labels))] * 2, dev)
    heat = ht.draw_heatmap(gt_boxes, gt_labels, n_gt, 10, [1440, 1440, 41], PC, VS, OSF, 0.1, 2).cpu().numpy()
    np.testing.assert_array_equal(heat[0], g["heatmap"])
    np.testing.assert_array_equal(heat[1], g["heatmap"])
    # boxes whose centre is at the border are clipped, not wrapped
    edge = boxes[:3].copy()
    edge[:, 0] = [-53.9, 53.9, 0.0]
    edge[:, 1] = [0.0, 53.9, -53.95]
    ref = ho.heatmap_targets(edge, labels[:3], 10, [1440, 1440, 41], PC, VS, OSF, 0.1, 2)
    gb, gl, n, _ = ht.pack_gt([(torch.from_numpy(edge), torch.from_numpy(labels[:3]))], dev)
    np.testing.assert_array_equal(ht.draw_heatmap(gb, gl, n, 10, [1440, 1440, 41], PC, VS, OSF, 0.1, 2)[0].cpu().numpy(), ref)


def test_gaussian_focal_loss_and_grad(dev):
    g = np.load(GOLD)
    target = torch.from_numpy(np.stack([g["heatmap"]] * 2)).to(dev)
    torch.manual_seed(0)
    logits = (torch.randn(2, 10, 180, 180, device=dev) * 3 - 2).requires_grad_(True)
    with torch.no_grad():
        logits[0, 0, 0, :4] = torch.tensor([-20.0, 20.0, -9.3, 9.3], device=dev)  # inside the clip_sigmoid plateau
    loss = ht.gaussian_focal_loss_with_logits(logits, target)
    loss.backward()
    x = logits.detach().double().clone().requires_grad_(True)
    p = x.sigmoid().clamp(1e-4, 1 - 1e-4)
    t = target.double()
    pos = t.eq(1)
    ref = (-(p + 1e-12).log() * (1 - p).pow(2) * pos + -(1 - p + 1e-12).log() * p.pow(2) * (1 - t).pow(4) * (~pos)).sum() / max(float(pos.sum()), 1)
    ref.backward()
    assert float(loss) == pytest.approx(float(ref), rel=2e-5)
    assert float(ho.gaussian_focal_loss(ho.clip_sigmoid(logits.detach().cpu().numpy()), target.cpu().numpy(),
                                        avg_factor=max(float(pos.sum()), 1))) == pytest.approx(float(ref), rel=1e-9)
    err = (logits.grad.double() - x.grad).abs().max() / x.grad.abs().max()
    assert float(err) < 1e-5
    assert float(logits.grad[0, 0, 0, 0]) == 0.0 and float(logits.grad[0, 0, 0, 1]) == 0.0  # clamped -> no gradient


def test_query_losses_and_grad(dev):
    B, C, P, K, L = 3, 10, 200, 10, 2
    torch.manual_seed(1)
    logits = (torch.randn(B, C, L * P, device=dev) * 2 - 1).requires_grad_(True)
    pred = torch.randn(B, K, L * P, device=dev, requires_grad=True)
    labels = torch.randint(0, C + 1, (B, P), device=dev, dtype=torch.int32)
    lw = (torch.rand(B, P, device=dev) > 0.1).float()
    bt = torch.randn(B, P, K, device=dev)
    bw = (labels < C).float()[:, :, None].expand(B, P, K).contiguous()
    cw = torch.tensor([1.0] * 8 + [0.2, 0.2], device=dev)
    for layer in range(L):
        logits.grad = pred.grad = None
        cls_sum, box_sum = ht.query_losses(logits, pred, labels, lw, bt, bw, cw, layer * P, P, 2.0, 0.25)
        (cls_sum / 7 + 0.25 * box_sum / 7).backward()
        sl = slice(layer * P, (layer + 1) * P)
        x = logits.detach().double().clone().requires_grad_(True)
        q = pred.detach().double().clone().requires_grad_(True)
        xs = x[..., sl].permute(0, 2, 1)
        t = torch.nn.functional.one_hot(labels.long(), C + 1)[..., :C].double()
        ps = xs.sigmoid()
        pt = (1 - ps) * t + ps * (1 - t)
        fw = (0.25 * t + 0.75 * (1 - t)) * pt.pow(2)
        ref_cls = (torch.nn.functional.binary_cross_entropy_with_logits(xs, t, reduction="none") * fw * lw.double()[..., None]).sum()
        ref_box = ((q[..., sl].permute(0, 2, 1) - bt.double()).abs() * bw.double() * cw.double()).sum()
        (ref_cls / 7 + 0.25 * ref_box / 7).backward()
        assert float(cls_sum) == pytest.approx(float(ref_cls), rel=2e-5)
        assert float(box_sum) == pytest.approx(float(ref_box), rel=2e-5)
        assert float(ho.sigmoid_focal_loss(xs.detach().reshape(-1, C).cpu().numpy(), labels.reshape(-1).cpu().numpy(),
                                           lw.reshape(-1).cpu().numpy())) == pytest.approx(float(ref_cls), rel=1e-9)
        assert float((logits.grad.double() - x.grad).abs().max() / x.grad.abs().max()) < 1e-5
        assert float((pred.grad.double() - q.grad).abs().max()) < 1e-6
        other = slice((1 - layer) * P, (2 - layer) * P)
        assert float(logits.grad[..., other].abs().max()) == 0.0  # the other decoder layer is untouched


def test_assigner_class_keeps_the_reference_call(dev):
    boxes, labels, pred, logits = scene(6000, 25)
    a = ht.HungarianAssigner3D(cls_cost=dict(type="mmdet.FocalLossCost", gamma=2.0, alpha=0.25, weight=0.15),
                               reg_cost=dict(type="BBoxBEVL1Cost", weight=0.25), iou_cost=dict(type="IoU3DCost", weight=0.25),
                               iou_calculator=dict(type="BboxOverlaps3D", coordinate="lidar"))
    res = a.assign(torch.from_numpy(pred).to(dev), torch.from_numpy(boxes).to(dev), torch.from_numpy(labels).to(dev),
                   torch.from_numpy(logits).to(dev)[None], dict(point_cloud_range=PC))
    ref, mo, lab, _, _ = ho.hungarian_assign(pred, boxes, labels, logits, PC)
    np.testing.assert_array_equal(res.gt_inds.cpu().numpy(), ref)
    np.testing.assert_array_equal(res.labels.cpu().numpy(), lab)
    np.testing.assert_allclose(res.max_overlaps.cpu().numpy(), mo, atol=2e-4)
    empty = a.assign(torch.from_numpy(pred).to(dev), torch.zeros(0, 9, device=dev), torch.zeros(0, dtype=torch.long, device=dev),
                     torch.from_numpy(logits).to(dev)[None], dict(point_cloud_range=PC))
    assert (empty.gt_inds == 0).all() and empty.max_overlaps is None


def test_circle_nms_matches_oracle(dev):
    rs = np.random.RandomState(3)
    for n, thresh in ((0, 0.175), (1, 0.175), (57, 0.175), (200, 0.5), (700, 2.0)):
        centres = rs.uniform(-5, 5, (max(n // 6, 1), 2))
        xy = (centres[rs.randint(0, len(centres), n)] + rs.normal(0, 0.3, (n, 2))).astype(np.float32)
        dets = np.concatenate([xy, rs.uniform(0, 1, (n, 1)).astype(np.float32)], 1) if n else np.zeros((0, 3), np.float32)
        if n > 10:
            dets[5, 2] = dets[9, 2]  # a score tie
        want = ho.circle_nms(dets, thresh, post_max_size=83)
        got = ht.circle_nms(torch.from_numpy(dets).to(dev), thresh, post_max_size=83).cpu().tolist()
        assert got == want, (n, thresh)
    if True:  # suppression really happens and the cap applies
        assert len(ho.circle_nms(dets, 2.0, 83)) < 700 and len(ht.circle_nms(torch.from_numpy(dets).to(dev), 1e-9, 83)) == 83


def _nms_boxes(rs, n):
    centres = rs.uniform(-20, 20, (max(n // 5, 1), 2))
    xy = centres[rs.randint(0, len(centres), n)] + rs.normal(0, 0.8, (n, 2))
    wh = rs.uniform(0.5, 4.5, (n, 2))
    ang = rs.uniform(-np.pi, np.pi, (n, 1))
    return np.concatenate([xy, wh, ang], 1).astype(np.float32), rs.uniform(0, 1, n).astype(np.float32)


def test_rotate_nms_matches_oracle(dev):
    """bfhip_rotate_nms (fp32 polygon intersection) against the fp64 restatement of nms_bev / mmcv nms_rotated: the kept
    index lists are identical whenever no pair's IoU lies within 1e-4 of the threshold (asserted for these seeds)."""
    rs = np.random.RandomState(11)
    for n, thresh, pre, post in ((0, 0.2, 1000, 83), (1, 0.2, 1000, 83), (64, 0.2, 1000, 83), (65, 0.1, 1000, 83),
                                 (200, 0.2, 1000, 83), (300, 0.05, 130, 40), (500, 0.3, None, None)):
        boxes, scores = _nms_boxes(rs, n) if n else (np.zeros((0, 5), np.float32), np.zeros(0, np.float32))
        if n > 20:
            scores[7] = scores[13]      # a score tie -> lower index first
            boxes[3, 2] = 0.0           # a degenerate box: IoU 0 with everything, always kept
            boxes[17] = boxes[16]       # identical boxes: IoU 1
        want, iou = ho.nms_rotated(boxes, scores, thresh, pre, post)
        assert n == 0 or float(np.abs(iou[iou > 0] - thresh).min(initial=1.0)) > 1e-4
        got = ht.rotate_nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), thresh, pre, post).cpu().tolist()
        assert got == want, (n, thresh)
        if n >= 200:
            assert len(want) < min(n, pre or n) or post is not None   # suppression really happens
    # the reference's calling convention: corner boxes in, the same kept set
    xyxyr = ht.xywhr2xyxyr(torch.from_numpy(boxes).to(dev))
    got2 = ht.nms_bev(xyxyr, torch.from_numpy(scores).to(dev), 0.3).cpu().tolist()
    xywhr = torch.stack(((xyxyr[:, 0] + xyxyr[:, 2]) / 2, (xyxyr[:, 1] + xyxyr[:, 3]) / 2, xyxyr[:, 2] - xyxyr[:, 0],
                         xyxyr[:, 3] - xyxyr[:, 1], xyxyr[:, 4]), -1).cpu().numpy()
    assert got2 == ho.nms_rotated(xywhr, scores, 0.3)[0]


def test_rotate_nms_large_is_consistent(dev):
    """4096 boxes (the ABI maximum): the kept set is an independent set of the IoU > thresh graph and every dropped box is
    covered by a kept box with a higher score (checked with a sampled fp64 IoU on the CPU)."""
    rs = np.random.RandomState(5)
    boxes, scores = _nms_boxes(rs, 4096)
    keep = ht.rotate_nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 0.2).cpu().numpy()
    assert 100 < len(keep) < 4096 and len(set(keep.tolist())) == len(keep)
    assert (np.diff(scores[keep]) <= 0).all()
    sub = keep[:60]
    iou = ho.box_iou_rotated(boxes[sub], boxes[sub])
    np.fill_diagonal(iou, 0)
    assert iou.max() <= 0.2 + 1e-4
    dropped = np.setdiff1d(np.arange(4096), keep)[:40]
    for j in dropped:
        higher = keep[scores[keep] >= scores[j]]
        near = higher[np.abs(boxes[higher, :2] - boxes[j, :2]).max(1) < 8]
        assert ho.box_iou_rotated(boxes[j:j + 1], boxes[near]).max() > 0.2 - 1e-4


def test_heuristic_assigner_matches_oracle(dev):
    boxes, labels, pred, logits = scene(7000, 30)
    qlab = np.random.RandomState(0).randint(0, 10, len(pred))
    qlab[:30] = labels  # so that some same-class matches exist
    a = ht.HeuristicAssigner3D(dist_thre=100)
    for use_labels in (False, True):
        res = a.assign(torch.from_numpy(pred).to(dev), torch.from_numpy(boxes).to(dev), None, torch.from_numpy(labels).to(dev),
                       torch.from_numpy(qlab).to(dev) if use_labels else None)
        inds, labs = ho.heuristic_assign(pred, boxes, labels, qlab if use_labels else None, 100.0)
        np.testing.assert_array_equal(res.gt_inds.cpu().numpy(), inds)
        np.testing.assert_array_equal(res.labels.cpu().numpy(), labs)
        m = inds > 0
        iou = ho.bbox_overlaps_3d_lidar(pred[m][:, :7], boxes[inds[m] - 1][:, :7]).diagonal()
        np.testing.assert_allclose(res.max_overlaps.cpu().numpy()[m], iou, atol=2e-4)
        assert (res.max_overlaps.cpu().numpy()[~m] == 0).all()


def test_predict_with_circle_nms(dev):
    from bevfusion_amd.bevfusion import nuscenes_config
    from bevfusion_amd.registry import MODELS
    cfg = nuscenes_config(camera=False)["bbox_head"]
    cfg["test_cfg"] = dict(cfg["test_cfg"], nms_type="circle")
    torch.manual_seed(0)
    head = MODELS.build(cfg).to(dev).eval()
    with torch.no_grad():
        feats = torch.randn(2, 512, 180, 180, device=dev)
        plain = MODELS.build(nuscenes_config(camera=False)["bbox_head"]).to(dev).eval()
        plain.load_state_dict(head.state_dict())
        a, b = head.predict(feats), plain.predict(feats)
    for ra, rb in zip(a, b):
        assert ra["bboxes_3d"].shape[0] <= rb["bboxes_3d"].shape[0]
        keep_other = rb["labels_3d"] < 8
        assert int((ra["labels_3d"] < 8).sum()) == int(keep_other.sum())  # classes 0-7 are not suppressed
    # any other nms_type: rotated-IoU NMS with the task radius as the threshold
    cfg["test_cfg"] = dict(cfg["test_cfg"], nms_type="rotate", pre_max_size=1000, post_max_size=83)
    rot = MODELS.build(cfg).to(dev).eval()
    rot.load_state_dict(head.state_dict())
    with torch.no_grad():
        c = rot.predict(feats)
    for rc, rb in zip(c, b):
        assert int((rc["labels_3d"] < 8).sum()) == int((rb["labels_3d"] < 8).sum())
        for cls in (8, 9):
            sel = rb["labels_3d"] == cls
            bev = rb["bboxes_3d"][sel][:, [0, 1, 3, 4, 6]].cpu().numpy()
            want, _ = ho.nms_rotated(bev, rb["scores_3d"][sel].cpu().numpy(), 0.175, 1000, 83)
            assert int((rc["labels_3d"] == cls).sum()) == len(want)
