#!/usr/bin/env python3
"""bench.py -- hot-path throughput on MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic nuScenes-shaped frames that
are already resident in HBM.  Frames are independent units: every rank processes
its own batch, there is no data-path collective, scaling is weak.  Rank 0 prints ONE JSON line.

The workload registry below names what one step contains; `config.workload` in the JSON line
says which one ran.  The roofline object is for the dominant kernel (bev_pool forward at the op
boundary) and is measured with HIP events recorded by the library on the kernel's own stream
(bfhip_profile_*).  cpu_baseline times the CPU oracle on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="frames per GPU per step")
    ap.add_argument("--workload", default="hotpath_v1")
    ap.add_argument("--points", type=int, default=40000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2, help="bounded CPU sample (frames)")
    return ap.parse_args()


class HotPathV1:
    """Per frame: deterministic hard voxelization of one 40k-point sweep (BF/bevfusion.py:227-255)
    + bev_pool forward and backward at the op boundary on the eval-aug nuScenes frustum
    (n_kept ~ 1.83 M rows x C=80 -> 360x360, BF/depth_lss.py:179-204).  The batch is processed
    the way the reference does: voxelization per sample, bev_pool on the whole batch (the batch id
    is part of the rank)."""

    name = "hotpath_v1: hard_voxelize(40k pts, nuScenes grid) + bev_pool fwd+bwd (op boundary, C=80, 360x360)"

    def __init__(self, device, batch, points, seed_base=0):
        import bevfusion_amd  # noqa: F401
        from bevfusion_amd import synthetic
        from bevfusion_amd.ops import Voxelization
        self.dev = device
        self.B = batch
        N = synthetic.NUSC
        self.N = N
        self.vox = Voxelization(N["voxel_size"], N["point_cloud_range"], N["max_num_points"], N["max_voxels"]).to(device)
        self.points_np = [synthetic.lidar_sweep(points, seed=1000 + seed_base + i) for i in range(batch)]
        self.points = [torch.from_numpy(p).to(device) for p in self.points_np]
        # camera geometry: eval augmentation, identical rig per sample -> ranks via the oracle-free
        # host glue (torch on device), done once: inputs are "already resident in HBM"
        self._build_intervals(synthetic)
        g = torch.Generator(device="cpu").manual_seed(2000 + seed_base)
        self.x = torch.randn(self.nk, N["C"], generator=g).to(device)
        self.out_grad = torch.randn(batch, 1, 360, 360, N["C"], generator=g).to(device)

    def _build_intervals(self, synthetic):
        from bevfusion_amd.ops.bev_pool.bev_pool import intervals_from_ranks
        dev = self.dev
        B = self.B
        rig = synthetic.camera_rig(batch=B)
        t = {k: torch.from_numpy(v).to(dev) for k, v in rig.items()}
        frustum = synthetic.create_frustum().to(dev)
        dx, bx, nx = [a.to(dev) for a in synthetic.gen_dx_bx()]
        Ncam = 6
        intr_inv = torch.inverse(t["camera_intrinsics"][..., :3, :3])
        post_rots_inv = torch.inverse(t["img_aug_matrix"][..., :3, :3])
        post_trans = t["img_aug_matrix"][..., :3, 3]
        pts = frustum - post_trans.view(B, Ncam, 1, 1, 1, 3)
        pts = post_rots_inv.view(B, Ncam, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(-1))
        pts = torch.cat((pts[..., :2, :] * pts[..., 2:3, :], pts[..., 2:3, :]), 5)
        comb = t["camera2lidar"][..., :3, :3].matmul(intr_inv)
        pts = comb.view(B, Ncam, 1, 1, 1, 3, 3).matmul(pts).squeeze(-1) + t["camera2lidar"][..., :3, 3].view(B, Ncam, 1, 1, 1, 3)
        cells = ((pts - (bx - dx / 2.0)) / dx).long().view(-1, 3)
        nprime = cells.shape[0]
        bidx = torch.arange(nprime, device=dev) // (nprime // B)
        kept = ((cells >= 0) & (cells < nx)).all(1)
        cells, bidx = cells[kept], bidx[kept]
        W, D = int(nx[1]), int(nx[2])
        ranks = cells[:, 0] * (W * D * B) + cells[:, 1] * (D * B) + cells[:, 2] * B + bidx
        order = ranks.argsort()
        self.ranks = ranks[order]
        self.geom = torch.cat((cells, bidx[:, None]), 1)[order].int().contiguous()
        self.starts, self.lengths = intervals_from_ranks(self.ranks)
        self.nk = int(self.ranks.shape[0])
        self.m = int(self.starts.shape[0])

    def step(self):
        from bevfusion_amd.ops import bev_pool_ext
        for p in self.points:
            self.vox(p)
        out = bev_pool_ext.bev_pool_forward(self.x, self.geom, self.lengths, self.starts, self.B, 1, 360, 360)
        xg = bev_pool_ext.bev_pool_backward(self.out_grad, self.geom, self.lengths, self.starts, self.B, 1, 360, 360,
                                            _cover_all=True)
        return out, xg

    # algorithmic bytes of ONE bev_pool_fwd launch: Nk*C*4 + m*24 + B*D*H*W*C*4
    def dominant_bytes(self):
        C = self.N["C"]
        return self.nk * C * 4 + self.m * 24 + self.B * 360 * 360 * C * 4

    dominant_op = "bev_pool_fwd"

    def cpu_baseline(self, frames):
        """The C oracle (single thread) on `frames` frames of the same workload."""
        import oracle
        N = self.N
        C = N["C"]
        per = self.nk // self.B
        # per-frame slice of the batch intervals: rebuild single-frame geometry on the host
        geom = self.geom.cpu().numpy()
        ranks = self.ranks.cpu().numpy()
        sel = geom[:, 3] == 0
        g1, r1 = geom[sel], ranks[sel]
        starts, lengths = oracle.intervals_from_ranks(r1)
        x1 = np.random.default_rng(0).standard_normal((g1.shape[0], C)).astype(np.float32)
        og = np.random.default_rng(1).standard_normal((1, 1, 360, 360, C)).astype(np.float32)
        t0 = time.perf_counter()
        for f in range(frames):
            oracle.hard_voxelize(self.points_np[f % self.B], N["voxel_size"], N["point_cloud_range"], 10, 120000)
            oracle.bev_pool_fwd(x1, g1, starts, lengths, 1, 1, 360, 360)
            oracle.bev_pool_bwd(og, g1, starts, lengths, g1.shape[0])
        dt = time.perf_counter() - t0
        return frames / dt, "%d frame(s): oracle hard_voxelize + bev_pool fwd+bwd, %d rows/frame" % (frames, per)


WORKLOADS = {"hotpath_v1": HotPathV1}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, "launch with --nproc-per-node == --gpus (WORLD_SIZE=%d, --gpus=%d)" % (world, args.gpus)

    from bevfusion_amd import _lib
    wl = WORKLOADS[args.workload](dev, args.batch, args.points, seed_base=100 * rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        wl.step()
    barrier()
    _lib.profile_enable(True)
    _lib.profile_read(wl.dominant_op, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    barrier()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    k_ms, k_cnt = _lib.profile_read(wl.dominant_op, reset=True)
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    frames = args.batch * world * args.steps
    value = frames / dt

    if rank == 0:
        avg_ms = k_ms / max(k_cnt, 1)
        achieved = wl.dominant_bytes() / (avg_ms * 1e-3) / 1e9 if k_cnt else None
        line = {
            "metric": "nuScenes frames/sec (hot path fwd+bwd)", "value": round(value, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.name, "frames_per_gpu_per_step": args.batch, "points_per_frame": args.points,
                       "frustum_rows_kept": wl.nk, "bev_intervals": wl.m, "parallelism": "independent frames per rank"},
            "roofline": {"bound": "hbm", "kernel": "bev_pool_fwd_v4", "achieved": round(achieved, 1) if achieved else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "traffic": None, "algorithmic_bytes_per_launch": wl.dominant_bytes(),
                         "avg_launch_ms": round(avg_ms, 5), "launches": k_cnt},
        }
        if world == 1 and not args.no_cpu_baseline:
            v, sample = wl.cpu_baseline(args.cpu_frames)
            line["cpu_baseline"] = {"value": round(v, 4), "unit": "frames/s", "cores": 1, "kind": "port", "sample": sample}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
