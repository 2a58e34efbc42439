#!/usr/bin/env python3
"""Which part of the training step survives hipGraph capture?  Each part runs in its own process:
    python tools/graph_probe.py            -> runs every part, one subprocess each
    python tools/graph_probe.py PART       -> captures + replays PART"""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
PARTS = ["lidar_fwd", "lidar_fwdbwd", "camera_fwd", "camera_fwdbwd", "head_fwdbwd", "full_fwd", "full_fwdbwd", "optimizer", "full_step"]


def run(part):
    import torch
    import bench
    os.environ["BENCH_GRAPH"] = "1" if part in ("optimizer", "full_step") else "0"
    dev = torch.device("cuda:0")
    wl = bench.FullModel(dev, int(os.environ.get("PROBE_BATCH", "2")), 40000)
    wl.use_graph = False
    wl.model.static_lidar = True
    if os.environ.get("PROBE_WORK"):
        wl.collect_work()
    wl.model.lidar_side_stream = os.environ.get("PROBE_SIDE", "0") == "1"
    for _ in range(3):
        wl._eager_step()
    torch.cuda.synchronize()
    from bevfusion_amd.head_targets import PackedGT
    gts = PackedGT(wl.gts, dev)
    m = wl.model
    wl.opt.zero_grad()

    def body():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if part.startswith("lidar"):
                out = m.extract_pts_feat(wl.inputs).float().square().mean()
            elif part.startswith("camera"):
                i = wl.inputs
                out = m.extract_img_feat(i["imgs"], i["points"], i["lidar2img"], i["cam2img"], i["cam2lidar"], i["img_aug_matrix"],
                                         i["lidar_aug_matrix"])[0].float().square().mean()
            elif part.startswith("head"):
                x = torch.randn(2, 512, 180, 180, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                out = m.parse_losses(m.bbox_head.loss([x], gts))[0]
            elif part == "optimizer":
                wl._update()
                return None
            else:
                out = m.parse_losses(m(wl.inputs, None, gts))[0]
        if part.endswith("bwd") or part == "full_step":
            out.backward()
        if part == "full_step":
            wl._update()
        return out

    if part == "optimizer":
        wl._forward_backward(gts)
    g = torch.cuda.CUDAGraph()
    print(part, "capturing", flush=True)
    with torch.cuda.graph(g):
        if os.environ.get("PROBE_COUNTER"):
            from bevfusion_amd import attention
            attention.step_counter(dev).add_(1)
        out = body()
    print(part, "captured", flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(part, "replayed ok", None if out is None else float(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for p in PARTS:
            r = subprocess.run([sys.executable, "-X", "faulthandler", __file__, p], capture_output=True, text=True, timeout=600)
            tail = [ln for ln in (r.stdout + r.stderr).splitlines() if p in ln or "Error" in ln or "error" in ln][-4:]
            print("==", p, "rc", r.returncode, "|", " | ".join(t[:160] for t in tail), flush=True)
