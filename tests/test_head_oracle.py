"""CPU: the head-target oracle (oracle/head_oracle.py) against the reference's own outputs (tests/golden/head_ref.npz,
made from M3D/models/utils/gaussian.py and BF/utils.py by tests/golden/make_golden.py) and against exact cases."""
import hashlib
import os

import numpy as np
import pytest
import torch

import bevfusion_amd  # noqa: F401
from bevfusion_amd import synthetic
from oracle import head_oracle as ho

GOLD = os.path.join(os.path.dirname(__file__), "golden", "head_ref.npz")
N = synthetic.NUSC
PC, VS, OSF = N["point_cloud_range"], N["voxel_size"], 8


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def gt(gold):
    boxes, labels = synthetic.gt_boxes(seed=3000, n=40)
    assert hashlib.sha256(boxes.tobytes()).hexdigest() == str(gold["gt_in_sha"])
    assert hashlib.sha256(labels.tobytes()).hexdigest() == str(gold["labels_in_sha"])
    return boxes, labels


def test_encode_matches_reference(gold, gt):
    enc = ho.bbox_encode(gt[0], PC, OSF, VS, 10)
    np.testing.assert_allclose(enc, gold["encode"], rtol=1e-6, atol=1e-6)


def test_decode_matches_reference(gold):
    for b in range(2):
        dec = ho.bbox_decode(gold["dec_center"][b], gold["dec_height"][b], gold["dec_dim"][b], gold["dec_rot"][b],
                             gold["dec_vel"][b], PC, OSF, VS)
        np.testing.assert_allclose(dec, gold["dec_boxes"][b], rtol=1e-6, atol=1e-5)
    assert (gold["dec_heat"].argmax(1) == gold["dec_labels"]).all()


def test_bev_l1_cost_matches_reference(gold, gt):
    c = ho.bev_l1_cost(gold["dec_boxes"][0], gt[0], PC, 0.25)
    np.testing.assert_allclose(c, gold["l1_cost"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gold["iou_cost_of_half"], -0.125)


def test_gaussian_radius_matches_reference(gold):
    r = np.array([ho.gaussian_radius(h, w, 0.1) for h, w in gold["radius_hw"]], np.float32)
    np.testing.assert_array_equal(r, gold["radius"])


def test_heatmap_matches_reference_bitwise(gold, gt):
    heat = ho.heatmap_targets(gt[0], gt[1], 10, [1440, 1440, 41], PC, VS, OSF, 0.1, 2)
    np.testing.assert_array_equal(heat, gold["heatmap"])
    assert (heat == 1).sum() >= 30  # one peak per box unless two boxes share a cell


def test_rotated_iou_exact_cases():
    sq = [0, 0, 2, 2, 0.0]
    assert ho.box_iou_rotated([sq], [sq])[0, 0] == pytest.approx(1.0)
    assert ho.box_iou_rotated([sq], [[1, 0, 2, 2, 0.0]])[0, 0] == pytest.approx(2 / 6)
    assert ho.box_iou_rotated([sq], [[0, 0, 2, 2, np.pi / 2]])[0, 0] == pytest.approx(1.0)
    inter = 8 * (np.sqrt(2) - 1)  # square of side 2 with itself turned 45 degrees: regular octagon
    assert ho.box_iou_rotated([sq], [[0, 0, 2, 2, np.pi / 4]])[0, 0] == pytest.approx(inter / (8 - inter))
    assert ho.box_iou_rotated([sq], [[5, 5, 2, 2, 0.3]])[0, 0] == 0.0
    # long thin box crossing a square
    assert ho.rotated_intersection_area([0, 0, 2, 2, 0], [0, 0, 10, 0.5, np.pi / 2]) == pytest.approx(1.0)


def test_rotated_iou_monte_carlo():
    rs = np.random.RandomState(0)
    pts = rs.uniform(-4, 4, (400000, 2))

    def inside(b, p):
        c, s = np.cos(b[4]), np.sin(b[4])
        d = p - np.array(b[:2])
        u, v = d[:, 0] * c + d[:, 1] * s, -d[:, 0] * s + d[:, 1] * c
        return (np.abs(u) <= b[2] / 2) & (np.abs(v) <= b[3] / 2)

    for _ in range(6):
        b1 = [rs.uniform(-1, 1), rs.uniform(-1, 1), rs.uniform(1, 4), rs.uniform(0.5, 3), rs.uniform(-3.2, 3.2)]
        b2 = [rs.uniform(-1, 1), rs.uniform(-1, 1), rs.uniform(1, 4), rs.uniform(0.5, 3), rs.uniform(-3.2, 3.2)]
        mc = (inside(b1, pts) & inside(b2, pts)).mean() * 64
        assert ho.rotated_intersection_area(b1, b2) == pytest.approx(mc, abs=0.05)


def test_iou3d_identity_and_height():
    b = np.array([[1.0, 2.0, -1.0, 4.0, 2.0, 1.5, 0.7]])
    assert ho.bbox_overlaps_3d_lidar(b, b)[0, 0] == pytest.approx(1.0)
    up = b.copy()
    up[0, 2] += 0.75  # half the height overlaps
    assert ho.bbox_overlaps_3d_lidar(b, up)[0, 0] == pytest.approx(0.5 / 1.5)


def test_hungarian_is_scipy_and_semantics():
    boxes, labels = synthetic.gt_boxes(seed=3001, n=7)
    rs = np.random.RandomState(1)
    pred = np.concatenate([boxes, boxes + rs.normal(0, 0.3, boxes.shape).astype(np.float32)])[:, :9]
    pred[:, 3:6] = np.abs(pred[:, 3:6]) + 0.1
    logits = rs.normal(-2, 1, (10, len(pred))).astype(np.float32)
    assigned, mo, lab, cost, iou = ho.hungarian_assign(pred, boxes, labels, logits, PC)
    assert sorted(assigned[assigned > 0]) == list(range(1, 8))  # every GT matched exactly once
    assert (lab[assigned > 0] == labels[assigned[assigned > 0] - 1]).all()
    assert (mo[assigned == 0] == 0).all()
    a0, *_ = ho.hungarian_assign(pred, boxes[:0], labels[:0], logits, PC)
    assert (a0 == 0).all()                                       # no GT: everything background (BF/utils.py:249-252)


def test_losses_against_torch_builtins():
    rs = np.random.RandomState(2)
    x = rs.normal(0, 2, (50, 10))
    lab = rs.randint(0, 11, 50)
    w = rs.randint(0, 2, 50).astype(np.float64)
    # focal loss = BCE-with-logits x focal weight (mmdet py_sigmoid_focal_loss)
    xt = torch.tensor(x)
    t = torch.nn.functional.one_hot(torch.tensor(lab), 11)[:, :10].double()
    p = xt.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    fw = (0.25 * t + 0.75 * (1 - t)) * pt.pow(2.0)
    ref = (torch.nn.functional.binary_cross_entropy_with_logits(xt, t, reduction="none") * fw * torch.tensor(w)[:, None]).sum() / 7
    assert ho.sigmoid_focal_loss(x, lab, w, 2.0, 0.25, 7) == pytest.approx(float(ref), rel=1e-10)
    pred = rs.uniform(0.01, 0.99, (4, 8))
    tgt = rs.uniform(0, 1, (4, 8))
    tgt[0, 0] = 1.0
    pos = -np.log(pred[0, 0] + 1e-12) * (1 - pred[0, 0]) ** 2
    tg = tgt.copy()
    full = ho.gaussian_focal_loss(pred, tg, avg_factor=1)
    tg2 = tgt.copy()
    tg2[0, 0] = 0.5
    neg_at = -np.log(1 - pred[0, 0] + 1e-12) * pred[0, 0] ** 2 * 0.5 ** 4
    assert full - ho.gaussian_focal_loss(pred, tg2, avg_factor=1) == pytest.approx(pos - neg_at, rel=1e-9)
    assert ho.l1_loss([[1, 2]], [[0, 4]], [[1, 0.5]], 2) == pytest.approx(1.0)


def test_nms_rotated_known_answers():
    """Hand-computed: axis-aligned overlap 3/5, a 90-degree copy (IoU 1), a far box; pre/post limits; tie order."""
    b = np.array([[0, 0, 2, 2, 0], [0.5, 0, 2, 2, 0], [5, 5, 1, 1, 0.3], [0, 0, 2, 2, np.pi / 2]], np.float32)
    s = np.array([0.9, 0.8, 0.7, 0.6], np.float32)
    keep, iou = ho.nms_rotated(b, s, 0.5)
    assert keep == [0, 2]
    np.testing.assert_allclose(iou[0], [0, 0.6, 0, 1.0], atol=1e-6)
    assert ho.nms_rotated(b, s, 0.65)[0] == [0, 1, 2]          # 0.6 is no longer above the threshold; box 3 falls to box 0
    assert ho.nms_rotated(b, s, 0.5, pre_max_size=1)[0] == [0]
    assert ho.nms_rotated(b, s, 0.5, post_max_size=1)[0] == [0]
    assert ho.nms_rotated(b[[1, 0]], np.array([0.5, 0.5], np.float32), 0.5)[0] == [0]   # tie: lower index first
    # a 45-degree unit square over an axis-aligned one, same centre: the intersection is a regular octagon of area 2(sqrt2-1)
    o = ho.box_iou_rotated(np.array([[0, 0, 1, 1, 0]]), np.array([[0, 0, 1, 1, np.pi / 4]]))[0, 0]
    inter = 2 * (np.sqrt(2) - 1)
    assert abs(o - inter / (2 - inter)) < 1e-9
    assert ho.nms_rotated(np.zeros((0, 5)), np.zeros(0), 0.5)[0] == []
