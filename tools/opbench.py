#!/usr/bin/env python3
"""Per-op micro-benchmark at BASELINE sizes (batch of B frames), HIP-event timed by the library.
Usage: python tools/opbench.py [--batch 4] [--iters 20]          (GPU box)"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bevfusion_amd  # noqa: E402,F401
from bevfusion_amd import _lib, synthetic  # noqa: E402
from bevfusion_amd.depth_lss import LSSTransform, lift_splat  # noqa: E402
from bevfusion_amd.ops import Voxelization, bev_pool_ext  # noqa: E402
from bevfusion_amd.sparse_encoder import BEVFusionSparseEncoder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, N = args.batch, synthetic.NUSC
    res = {}
    # ---- camera plan + lift-splat + op-boundary bev_pool
    vt = LSSTransform(in_channels=256, out_channels=80, image_size=N["image_size"], feature_size=N["feature_size"],
                      xbound=N["xbound"], ybound=N["ybound"], zbound=N["zbound"], dbound=N["dbound"]).to(dev)
    rig = synthetic.camera_rig(batch=B, seed=1, train_aug=True)
    t = {k: torch.from_numpy(v).to(dev) for k, v in rig.items()}
    cal = vt._calibration(t["camera_intrinsics"], t["camera2lidar"], t["img_aug_matrix"], t["lidar_aug_matrix"])
    plan = vt.make_plan(**cal, with_reference_outputs=True)
    nk, m = [int(v) for v in plan.counts.cpu()]
    P, D, C = B * 6 * 32 * 88, vt.D, 80
    depth = torch.softmax(torch.randn(P, D, device=dev), 1).requires_grad_(True)
    feat = torch.randn(P, C, device=dev).requires_grad_(True)
    pd = plan.sorted_pd[:nk].long() & 0xFFFFFFFF
    x = (depth.detach()[pd >> 8, pd & 255].unsqueeze(1) * feat.detach()[pd >> 8]).contiguous()
    geom, starts, lengths = plan.geom_sorted[:nk].contiguous(), plan.starts[:m].contiguous(), plan.lengths[:m].contiguous()
    og = torch.randn(B, 1, 360, 360, C, device=dev)
    pts = [torch.from_numpy(synthetic.lidar_sweep(40000, seed=1000 + i)).to(dev) for i in range(B)]
    vox = Voxelization(N["voxel_size"], N["point_cloud_range"], 10, (120000, 160000)).to(dev)
    enc = BEVFusionSparseEncoder(in_channels=5, sparse_shape=[1440, 1440, 41], norm_cfg=dict(type="BN1d", eps=0.001, momentum=0.01),
                                 encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                                 encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, (1, 1, 0)), (0, 0)), block_type="basicblock").to(dev)
    from bevfusion_amd.bevfusion import voxel_mean
    vs = [vox(p) for p in pts]
    vfeats = voxel_mean(torch.cat([v[0] for v in vs]), torch.cat([v[2] for v in vs]))
    coords = torch.cat([torch.nn.functional.pad(v[1], (1, 0), value=i) for i, v in enumerate(vs)])

    def run(n):
        for _ in range(n):
            vt.make_plan(**cal)
            out = lift_splat(depth, feat, plan)
            out.backward(og)
            bev_pool_ext.bev_pool_forward(x, geom, lengths, starts, B, 1, 360, 360)
            bev_pool_ext.bev_pool_backward(og, geom, lengths, starts, B, 1, 360, 360, _cover_all=True)
            for p in pts:
                vox(p)
            y = enc(vfeats, coords, B)
            y.mean().backward()

    run(3)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for op in _lib.OPS:
        _lib.profile_read(op, reset=True)
    run(args.iters)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    work = {
        "bev_pool_fwd": nk * C * 4 + m * 24 + B * 360 * 360 * C * 4,
        "bev_pool_bwd": m * C * 4 + m * 24 + nk * C * 4,
        "lift_splat_fwd": P * D * 4 + P * C * 4 + nk * 4 + B * 360 * 360 * C * 4,
        "lift_splat_bwd": 2 * P * D * 4 + 2 * P * C * 4 + P * D * 4 + m * C * 4,
        "bev_aux": B * 6 * D * 2816 * 20,
    }
    for op in _lib.OPS:
        ms, cnt = _lib.profile_read(op, reset=True)
        if cnt:
            e = {"avg_ms": round(ms / cnt, 5), "launches_per_iter": cnt / args.iters, "ms_per_iter": round(ms / args.iters, 4)}
            if op in work:
                e["alg_GB_s"] = round(work[op] / (ms / cnt * 1e-3) / 1e9, 1)
            res[op] = e
    res["_sizes"] = dict(batch=B, frustum_kept=nk, intervals=m, voxels=int(coords.shape[0]))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
