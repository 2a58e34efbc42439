#!/usr/bin/env python3
"""lift_splat fwd + bwd alone at batch-4 nuScenes sizes in the four storage variants (feat fp32 | bf16) x (BEV out fp32 | bf16):
library event timer per launch.  Under rocprofv3 --pmc the kernels can be told apart by name (lift_splat_fwd_kernel /
lift_splat_fwd16_kernel, <false> / <true> = fp32 / bf16 output)."""
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
plan = vt.make_plan(**vt._calibration(t["camera_intrinsics"], t["camera2lidar"], t["img_aug_matrix"], t["lidar_aug_matrix"]))
P, D, C = B * 6 * 32 * 88, vt.D, 80
depth = torch.softmax(torch.randn(P, D, device=dev), 1).requires_grad_(True)
res = {"n_kept_and_intervals": plan.counts.tolist()}
only = os.environ.get("LS_ONLY")
for fdt in (torch.float32, torch.bfloat16):
    for odt in (torch.float32, torch.bfloat16):
        name = "feat_%s_out_%s" % ("bf16" if fdt == torch.bfloat16 else "f32", "bf16" if odt == torch.bfloat16 else "f32")
        if only and only != name:
            continue
        feat = torch.randn(P, C, device=dev).to(fdt).requires_grad_(True)
        og = torch.randn(B, 1, 360, 360, C, device=dev).to(odt)

        def run():
            depth.grad = feat.grad = None
            lift_splat(depth, feat, plan, odt).backward(og)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for op in ("lift_splat_fwd", "lift_splat_bwd"):
            _lib.profile_read(op, reset=True)
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        res[name] = {op: round(_lib.profile_read(op, reset=True)[0] / 10, 4) for op in ("lift_splat_fwd", "lift_splat_bwd")}
print(json.dumps(res))
