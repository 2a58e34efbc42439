#!/usr/bin/env python3
"""Times the TransFusion target/loss path (csrc/head.hip) at batch 4 and, beside it, the reference's formulation of
the Hungarian step (cost.cpu() + scipy, BF/utils.py:266-272) on the same cost matrices.  GPU box only."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bevfusion_amd  # noqa: E402,F401
from bevfusion_amd import head_targets as ht, synthetic  # noqa: E402
from bevfusion_amd.bevfusion import nuscenes_config  # noqa: E402
from bevfusion_amd.registry import MODELS  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, P = 4, 200
    head = MODELS.build(nuscenes_config(camera=False)["bbox_head"]).to(dev).train()
    gts = [tuple(torch.from_numpy(a) for a in synthetic.gt_boxes(seed=3000 + i)) for i in range(B)]
    torch.manual_seed(0)
    feats = torch.randn(B, 512, 180, 180, device=dev)
    preds = head(feats)
    p0 = preds[0][0]

    def timed(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    res = {"n_gt": [int(g[0].shape[0]) for g in gts]}
    res["get_targets_ms"] = timed(lambda: head.get_targets(gts, p0))
    res["loss_by_feat_fwd_ms"] = timed(lambda: head.loss_by_feat(preds, gts))

    def fwd_bwd():
        for k in ("heatmap", "center", "height", "dim", "rot", "vel", "dense_heatmap"):
            if p0[k].grad is not None:
                p0[k].grad = None
        losses = head.loss_by_feat(preds, gts)
        sum(v for k, v in losses.items() if "loss" in k).backward(retain_graph=True)
    res["loss_fwd_bwd_ms"] = timed(fwd_bwd, 20)
    gt_boxes, gt_labels, n_gt, _ = ht.pack_gt(gts, dev)
    boxes = head.bbox_coder.decode_boxes(p0["rot"], p0["dim"], p0["center"], p0["height"], p0["vel"])
    assigned, iou, cost, _ = ht.assign_batch(boxes, p0["heatmap"], gt_boxes, gt_labels, n_gt, head.train_cfg["point_cloud_range"], head.assign_weights)
    res["cost_plus_hungarian_ms"] = timed(lambda: ht.assign_batch(boxes, p0["heatmap"], gt_boxes, gt_labels, n_gt,
                                                                  head.train_cfg["point_cloud_range"], head.assign_weights))
    res["hungarian_only_ms"] = timed(lambda: ht.hungarian(cost, n_gt))
    res["heatmap_ms"] = timed(lambda: ht.draw_heatmap(gt_boxes, gt_labels, n_gt, 10, [1440, 1440, 41], head.train_cfg["point_cloud_range"],
                                                      head.train_cfg["voxel_size"], 8, 0.1, 2))
    from scipy.optimize import linear_sum_assignment
    t0 = time.perf_counter()
    for _ in range(20):
        for b in range(B):
            c = cost[b, :, :res["n_gt"][b]].detach().cpu()      # the reference's round trip, per sample
            linear_sum_assignment(c)
    res["reference_style_cpu_scipy_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    # worst case sizes
    big = torch.rand(B, 200, 200, device=dev)
    nb = torch.full((B,), 200, dtype=torch.int32, device=dev)
    res["hungarian_200x200_ms"] = timed(lambda: ht.hungarian(big, nb))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
