#!/usr/bin/env python3
"""lift_splat_fwd alone at batch-4 nuScenes sizes, camera-major vs rank order (library event timer).  Under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc TCC_HIT_sum TCC_MISS_sum` the per-dispatch counters of the two orders can be compared
(the two variants are launched in separate, labelled phases: first 10 dispatches rank order, next 10 camera-major)."""
import json
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bevfusion_amd  # noqa: E402,F401
from bevfusion_amd import _lib, synthetic  # noqa: E402
from bevfusion_amd.depth_lss import LSSTransform, lift_splat  # noqa: E402

dev = torch.device("cuda:0")
B, N = int(os.environ.get("LS_BATCH", "4")), synthetic.NUSC
vt = LSSTransform(in_channels=256, out_channels=80, image_size=N["image_size"], feature_size=N["feature_size"],
                  xbound=N["xbound"], ybound=N["ybound"], zbound=N["zbound"], dbound=N["dbound"]).to(dev)
rig = synthetic.camera_rig(batch=B, seed=1, train_aug=True)
t = {k: torch.from_numpy(v).to(dev) for k, v in rig.items()}
cal = vt._calibration(t["camera_intrinsics"], t["camera2lidar"], t["img_aug_matrix"], t["lidar_aug_matrix"])
plan = vt.make_plan(**cal)
P, D, C = B * 6 * 32 * 88, vt.D, 80
depth = torch.softmax(torch.randn(P, D, device=dev), 1)
feat = torch.randn(P, C, device=dev)
order = plan.interval_order
res = {}
for name, o in (("rank_order", None), ("camera_major", order)):
    plan.interval_order = o
    for _ in range(3):
        lift_splat(depth, feat, plan)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    _lib.profile_read("lift_splat_fwd", reset=True)
    for _ in range(10):
        lift_splat(depth, feat, plan)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    ms, cnt = _lib.profile_read("lift_splat_fwd", reset=True)
    res[name] = round(ms / cnt, 4)
print(json.dumps(res))
