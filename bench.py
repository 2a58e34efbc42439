#!/usr/bin/env python3
"""bench.py -- hot-path throughput on MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic nuScenes-shaped frames that
are already resident in HBM.  Frames are independent units: every rank processes
its own batch, there is no data-path collective, scaling is weak.  Rank 0 prints ONE JSON line.

The workload registry below names what one step contains; `config.workload` in the JSON line
says which one ran.  `roofline` is for the dominant hand-written op of the workload (the one with
the largest time per step: the fused BatchNorm backward in `full` / `camera_only`, a sparse-conv op
in `lidar_*`, bev_pool_fwd in `hotpath_v1`); `roofline_ops` lists every hand-written op of the
hot path with its own fraction.  Times are HIP events recorded by the library on the op's own
stream (bfhip_profile_*) inside the timed region: the sparse / lift-splat / voxel ops on every step,
the dense ops (conv2d_*, bn2d_*: ~280 more event pairs per step) on every 10th step.  `cpu_baseline` (rank 0, N = 1) times the reference's CPU formulation
of the path on all host cores: C oracle voxelization, restated QuickCumsum bev_pool, the 21-layer
sparse encoder as gather -> mm -> index_add_, and the config-0 forward of one frame
(oracle/cpu_pipeline.py); it is a reported baseline, never the thing measured.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import importlib.util as _ilu  # noqa: E402

# the hosts of this pool only support dmabuf IPC: RCCL between the ranks of a node needs this before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# MIOpen reads MIOPEN_USER_DB_PATH when it initialises: point it at the shipped tuning db before torch loads
_spec = _ilu.spec_from_file_location("_bfhip_tuning", os.path.join(ROOT, "bevfusion-3d_object_detection_amd", "tuning", "__init__.py"))
_tuning = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(_tuning)
MIOPEN_DB = _tuning.use_shipped_miopen_db()

import numpy as np  # noqa: E402
import torch  # noqa: E402

MIOPEN_FIND = os.environ.get("BENCH_MIOPEN_FIND", "0") == "1"
# gradient exchange between ranks: "flat" = one all-reduce per dtype after the backward (default), "ddp" = torch DDP buckets
GRAD_SYNC = os.environ.get("BENCH_GRAD_SYNC", "flat")
# library profiler level: 1 = sparse / lift-splat / voxel ops, 2 = also the dense ops (conv2d_*, bn2d_*: ~280 more event pairs
# per training step)
PROFILE_LEVEL = int(os.environ.get("BENCH_PROFILE_LEVEL", "2"))
DENSE_EVERY = max(1, int(os.environ.get("BENCH_PROFILE_DENSE_EVERY", "10")))
DENSE_OPS = ("conv2d_fwd", "conv2d_dgrad", "conv2d_wgrad", "bn2d_fwd", "bn2d_bwd", "conv2d_pw_fwd", "conv2d_pw_dgrad")
# `full` runs on three HIP queues (camera / BEV chain; LiDAR branch; weight gradients), so an event pair around an op in the
# timed region also measures the wait for CUs the other queues hold.  Roofline fractions are computed from
# `kernel_ms_per_step`: the same ops timed over ISOLATED_STEPS extra steps after the timed region with the side streams off
# (one queue, nothing beside them) -- the number a rocprofv3 kernel trace reproduces.
HOST_SLEEP_US = int(os.environ.get("BENCH_HOST_SLEEP_US", "0"))
# The backward pass runs on the calling thread (BENCH_AUTOGRAD_MT=1: on the autograd engine's device thread, torch's default): with one
# process per GPU there is nothing for a per-device thread to overlap, and the hand-over costs host time -- issue time of the `full`
# step 24.4-24.6 -> 20.1-21.0 ms on a moderately busy host (tools/host_bound.py, alternating runs); same kernels, same streams.
AUTOGRAD_ON_CALLER = os.environ.get("BENCH_AUTOGRAD_MT", "0") == "0"
PREWARM_STEPS = int(os.environ.get("BENCH_PREWARM_STEPS", "6"))
SETTLE_MAX_S = float(os.environ.get("BENCH_SETTLE_MAX_S", "15"))


def _issue_stats(ts):
    """median / max of the host-side intervals between consecutive steps of the timed region (informational: `ms_per_step` is
    the synchronised wall time over all K steps; a median well below it means a few steps were held up -- a busy host)."""
    d = sorted((b - a) * 1e3 for a, b in zip(ts[:-1], ts[1:]))
    if not d:
        return None
    return {"median": round(d[len(d) // 2], 3), "max": round(d[-1], 3), "min": round(d[0], 3)}
LIDAR_OPS = ("hard_voxelize", "spconv_fwd", "spconv_bwd", "spconv_wgrad", "spconv_wgrad_main", "rulebook")
NO_WORK = os.environ.get("BENCH_NO_WORK") == "1"  # counter passes under rocprofv3: warm-up + timed steps only, nothing else
ISOLATED_STEPS = 0 if NO_WORK else int(os.environ.get("BENCH_ISOLATED_STEPS", "5"))
CPU_REPEATS = max(1, int(os.environ.get("BENCH_CPU_REPEATS", "3")))
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured achievable)
# dense matrix-core peaks (MI355X_MICROARCH.md): bf16 v_mfma_f32_32x32x16_bf16 / fp32-input v_mfma_f32_16x16x4_f32
MFMA_PEAK_BF16 = (2500.0, "bf16 MFMA dense peak ~2500 TFLOP/s (v_mfma_f32_*_bf16)")
MFMA_PEAK_F32 = (157.3, "fp32-input MFMA peak 157.3 TFLOP/s (v_mfma_f32_16x16x4_f32)")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--batch", type=int, default=4, help="frames per GPU per step")
    ap.add_argument("--workload", default="full", choices=["full", "lidar_only", "lidar_branch", "camera_only", "hotpath_v1", "dist_selftest"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: --workload dist_selftest (CPU), or a one-GPU rehearsal of the multi-rank path with BENCH_ONE_GPU=1")
    ap.add_argument("--points", type=int, default=40000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=1, help="bounded CPU sample (frames)")
    ap.add_argument("--vt-fp32", action="store_true", help="view-transform conv stacks in fp32 (the reference's fp32 island, "
                    "BF/bevfusion.py:177) instead of bf16")
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



class _ModelWorkload:
    """Shared driver for the model-level workloads: fwd + head targets/losses + bwd + grad-clip + AdamW step
    (optimizer settings of the reference config: AdamW lr 2e-4, wd 0.01, clip_grad max_norm 35,
    bevfusion_lidar_voxel0075...py:369-372)."""

    camera = True
    lidar = True
    amp = True
    channels_last = True

    def __init__(self, device, batch, points, seed_base=0, ddp=False, local_rank=0):
        import bevfusion_amd  # noqa: F401
        from bevfusion_amd import synthetic
        from bevfusion_amd.bevfusion import BEVFusion, nuscenes_config
        from bevfusion_amd.registry import MODELS
        self.dev, self.B = device, batch
        self.N = synthetic.NUSC
        torch.manual_seed(0)  # identical weights on every rank
        self.model = MODELS.build(nuscenes_config(camera=self.camera, lidar=self.lidar)).to(device).train()
        if self.channels_last:
            for name in ("img_backbone", "img_neck", "view_transform", "fusion_layer", "pts_backbone", "pts_neck", "bbox_head"):
                sub = getattr(self.model, name, None)  # 2-D conv stacks only (sparse conv weights are 5-D)
                if sub is not None:
                    sub.to(memory_format=torch.channels_last)
        if self.channels_last and getattr(self.model, "pts_middle_encoder", None) is not None:
            self.model.pts_middle_encoder.bev_channels_last = True  # BEV map handed to the NHWC convs without a relayout
        self.vt_bf16 = self.amp and os.environ.get("BENCH_VT_FP32", "0") != "1"
        if self.vt_bf16 and getattr(self.model, "view_transform", None) is not None:
            self.model.view_transform.conv_dtype = torch.bfloat16  # dense convs bf16, index paths + pooling fp32
        self.model.lidar_side_stream = os.environ.get("BENCH_SIDE_STREAM", "1") == "1"
        torch.backends.cudnn.benchmark = MIOPEN_FIND               # MIOpen exhaustive find (minutes of warm-up on a fresh box)
        self.n_params = sum(p.numel() for p in self.model.parameters())
        self.step_model = self.model
        self.master_weights = self.amp and os.environ.get("BENCH_MASTER_WEIGHTS", "1") == "1"
        self.use_graph = os.environ.get("BENCH_GRAPH", "0") == "1"
        if self.use_graph:
            self.model.static_lidar = True  # a captured step cannot contain the exact path's host reads
        if self.master_weights:
            # conv / linear weights held in bf16 (what the kernels consume), fp32 masters in the optimizer: same arithmetic
            # as autocast without ~320 per-step cast launches; DDP then reduces bf16 gradients for these layers
            from bevfusion_amd.amp import MasterWeightAdamW
            exclude = ("pts_middle_encoder", "heatmap_head") + (() if self.vt_bf16 else ("view_transform",))  # fp32 island: fp32 weights
            self.opt = MasterWeightAdamW(self.model, lr=2e-4, weight_decay=0.01, max_grad_norm=35.0, exclude=exclude,
                                         capturable=self.use_graph)  # before DDP: dtypes fixed
        else:
            self.opt = torch.optim.AdamW(self.model.parameters(), lr=2e-4, weight_decay=0.01, fused=True,
                                         capturable=self.use_graph)
        self.grad_sync = None
        self._graph, self._calls, self._graph_has_update = None, 0, False
        if ddp and GRAD_SYNC == "ddp":
            from torch.nn.parallel import DistributedDataParallel as DDP
            # BatchNorm statistics stay local (no SyncBN in the reference configs): buffers are not broadcast
            self.step_model = DDP(self.model, device_ids=[local_rank], gradient_as_bucket_view=True, broadcast_buffers=False)
        elif ddp and GRAD_SYNC != "none":  # "none": process group only (measures what the communicator itself costs)
            # one flat all-reduce per dtype after the backward (bevfusion_amd/grad_sync.py): the DDP wrapper costs 3 ms of
            # host time per step on this host-bound model; BatchNorm statistics stay local as under DDP above
            from bevfusion_amd.grad_sync import FlatGradAllReduce, broadcast_parameters
            broadcast_parameters(self.model)
            self.grad_sync = FlatGradAllReduce(self.model.parameters())
        self.parse_losses = BEVFusion.parse_losses
        from bevfusion_amd import conv2d as _c2
        # opt-in: over 6 + 6 alternating runs on one box the third queue changes the mean step time by +0.1 ms (31.35 vs
        # 31.21 ms) and widens its spread (30.5 ... 32.5 vs 31.0 ... 31.4 ms): the weight gradients fill the chip by themselves
        _c2.WGRAD_SIDE_STREAM = os.environ.get("BENCH_WGRAD_SIDE_STREAM", "0") == "1"
        if ddp and GRAD_SYNC == "ddp":
            # torch DDP reduces a bucket from the AccumulateGrad hooks of its parameters; the grouped weight gradients store .grad
            # themselves at the end of the pass and never reach those hooks
            _c2.WGRAD_GROUPED = False
        self.wgrad_grouped = bool(_c2.WGRAD_GROUPED and not _c2.WGRAD_SIDE_STREAM and not self.use_graph)
        self.backward_on_caller = bool(AUTOGRAD_ON_CALLER and not (ddp and GRAD_SYNC == "ddp"))
        torch.autograd.set_multithreading_enabled(not self.backward_on_caller)
        self._wgrad_join = _c2.wgrad_join if _c2.WGRAD_SIDE_STREAM else None
        self._params = [p for p in self.model.parameters() if p.requires_grad]
        # ground truth as a dataloader hands it over: per-frame host tensors (boxes [G, 9], labels [G]), G ~ U(15, 60)
        self.gts = [tuple(torch.from_numpy(a) for a in synthetic.gt_boxes(seed=3000 + seed_base + i)) for i in range(batch)]
        self.inputs = {}
        if self.lidar or self.camera:
            self.points_np = [synthetic.lidar_sweep(points, seed=1000 + seed_base + i) for i in range(batch)]
            self.inputs["points"] = [torch.from_numpy(p).to(device) for p in self.points_np]
        if self.camera:
            rig = synthetic.camera_rig(batch=batch, seed=seed_base + 1, train_aug=True)
            g = torch.Generator().manual_seed(2000 + seed_base)
            self.inputs["imgs"] = torch.randn(batch, 6, 3, 256, 704, generator=g).to(device)
            for src, dst in (("lidar2image", "lidar2img"), ("camera_intrinsics", "cam2img"), ("camera2lidar", "cam2lidar"),
                             ("img_aug_matrix", "img_aug_matrix"), ("lidar_aug_matrix", "lidar_aug_matrix")):
                self.inputs[dst] = torch.from_numpy(rig[src]).to(device)
        self.nk = self.m = None
        self._layer_stats = None

    def _forward_backward(self, gts):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.amp):
            # BEVFusion.loss: extract_feat + TransFusion head + Hungarian targets + focal / L1 / gaussian-focal losses
            losses = self.step_model(self.inputs, None, gts)
            loss, self.log_vars = self.parse_losses(losses)  # (loss, log_vars) as BF/bevfusion.py:88-121
        loss.backward()
        if self._wgrad_join is not None:
            self._wgrad_join()  # weight gradients launched on their own stream (conv2d.WGRAD_SIDE_STREAM): join before any use
        return loss

    def _update(self):
        if self.grad_sync is not None:
            self.grad_sync.reduce()
        if not self.master_weights:
            from bevfusion_amd.amp import skip_nonfinite_step
            skip_nonfinite_step(self.opt, torch.nn.utils.clip_grad_norm_(self._params, 35.0, foreach=True))
        self.opt.step()

    def _eager_step(self):
        if HOST_SLEEP_US:  # experiment: is the step bound by its host side?  (a host-bound step grows by the sleep, a GPU-bound one does not)
            time.sleep(HOST_SLEEP_US * 1e-6)
        self.opt.zero_grad() if self.master_weights else self.opt.zero_grad(set_to_none=True)
        loss = self._forward_backward(self.gts)
        self._update()
        return loss.detach()  # a caller holding the loss must not keep the step's autograd nodes alive

    def step(self):
        """Eager by default.  BENCH_GRAPH=1: after three eager steps (row capacities of the LiDAR branch learnt, workspaces
        and MIOpen solutions in place) the whole step -- forward of both branches on their two HIP streams, head targets and
        losses, backward, gradient clipping, AdamW -- is captured ONCE into a hipGraph and replayed: the step has no host
        reads (static capacity mode, device-resident ground truth, device-side dropout counter), so nothing in it needs
        the host.  With a gradient exchange between ranks the graph ends after the backward."""
        if not self.use_graph:
            return self._eager_step()
        self._calls += 1
        if self._graph is None:
            if self._calls <= 3:
                return self._eager_step()
            self._capture()
        self._graph.replay()
        if not self._graph_has_update:
            self._update()
        return self._static_loss

    def _capture(self):
        from bevfusion_amd import attention
        from bevfusion_amd.head_targets import PackedGT
        self._gts_dev = PackedGT(self.gts, self.dev)
        self._graph_has_update = self.grad_sync is None and self.step_model is self.model
        torch.cuda.synchronize()
        self.opt.zero_grad() if self.master_weights else self.opt.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            attention.step_counter(self.dev).add_(1)  # a replay repeats its host-side dropout seed: the counter does not
            loss = self._forward_backward(self._gts_dev)
            if self._graph_has_update:
                self._update()
        self._graph, self._static_loss = g, loss.detach()

    # ---- algorithmic work of the hand-written ops of ONE step (for the roofline object)
    def collect_work(self):
        """One instrumented forward (host syncs allowed here, outside the timed region)."""
        from bevfusion_amd import spconv as sp
        work = {}
        if self.lidar:
            layers = []
            orig = sp._SparseConvFunction.forward

            def spy(ctx, features, weight, data, n_in):
                P = int(data.n_pairs.sum().item())
                layers.append((P, weight.shape[-1], weight.shape[0], n_in, data.pair_fwd.shape[1]))
                return orig(ctx, features, weight, data, n_in)

            sp._SparseConvFunction.forward = staticmethod(spy)
            try:
                with torch.no_grad():
                    self.model.extract_pts_feat(self.inputs)
            finally:
                sp._SparseConvFunction.forward = staticmethod(orig)
            flops = sum(2.0 * P * ci * co for P, ci, co, _, _ in layers)
            # algorithmic bytes: features in and out once (bf16 storage under bf16 autocast), 8 B per rulebook pair,
            # the fp32 weights once
            fb = 2 if self.amp else 4
            byts = sum((ni * ci + no * co) * fb + P * 8 + 27 * ci * co * 4 for P, ci, co, ni, no in layers)
            # under bf16 autocast forward, data gradient AND weight gradient run on the bf16 MFMA (the 5-channel first layer
            # excepted); in fp32 all three use the fp32-input MFMA.  The binding roof is the lower of the two ceilings
            # min(MFMA peak, intensity x HBM peak): with bf16 MFMAs these ops sit on the HBM side of the ridge.
            peak, note = MFMA_PEAK_BF16 if self.amp else MFMA_PEAK_F32
            bound = "mfma" if flops / byts * HBM_PEAK_GBS * 1e9 > peak * 1e12 else "hbm"
            sp_work = dict(bound=bound, flops=flops, bytes=byts, unit_peak=peak, peak_note=note)
            work["spconv_fwd"] = dict(sp_work, scope="21 gather-GEMM launches (all sparse conv layers of the step)")
            work["spconv_bwd"] = dict(sp_work, scope="dgrad: 20 gather-GEMM launches")
            work["spconv_wgrad"] = dict(sp_work, scope="whole op x 21 layers: main kernel + partial-slab reduce (+ offset counts on the fp32 path)")
            work["spconv_wgrad_main"] = dict(sp_work, scope="dominant kernel only (rocprofv3: spconv_wgrad*_kernel)")
            # rulebooks actually built per step (SubM ones are shared per stage): N_in*16 + hash table 2*N*8 + P*8
            uniq = {(P, ni, no) for P, _, _, ni, no in layers}
            work["rulebook"] = dict(bound="hbm", bytes=sum(ni * 16 + 2 * max(ni, no) * 8 + P * 8 for P, ni, no in uniq),
                                    scope="%d rulebook builds per step (hash / bitmap, pairs, row masks, row sort)" % len(uniq))
            self._layer_stats = layers
        if self.camera:
            vt = self.model.view_transform
            cal = vt._calibration(self.inputs["cam2img"], self.inputs["cam2lidar"], self.inputs["img_aug_matrix"],
                                  self.inputs["lidar_aug_matrix"])
            plan = vt.make_plan(**cal)
            self.nk, self.m = [int(v) for v in plan.counts.cpu()]
            C, D = vt.C, vt.D
            P = self.B * 6 * 32 * 88
            cells = self.B * 360 * 360
            # element sizes as STORED: with bf16 conv stacks in the view transform the feature rows, the BEV map and its
            # gradient are bf16 (depth_lss.BF16_FEAT / BF16_BEV_OUT); depth, d_depth and the index arrays are 4-byte
            from bevfusion_amd import depth_lss as dl
            low = getattr(vt, "conv_dtype", None) == torch.bfloat16
            fs = 2 if (low and dl.BF16_FEAT and C % 8 == 0) else 4   # feat / d_feat
            os_ = 2 if (low and dl.BF16_BEV_OUT) else 4               # BEV map / its gradient
            work["lift_splat_fwd"] = dict(bound="hbm", bytes=P * D * 4 + P * C * fs + self.nk * 4 + cells * C * os_,
                                          scope="fused outer product + gathers + bev_pool forward, one launch "
                                                "(feat %d B/elem, BEV out %d B/elem)" % (fs, os_))
            work["lift_splat_bwd"] = dict(bound="hbm", bytes=P * D * 4 + P * C * fs + P * D * 4 + self.m * C * os_ + P * D * 4 + P * C * fs,
                                          scope="fused backward, one launch (depth + feat + cell map read, touched out_grad rows, "
                                                "d_depth + d_feat written)")
        if PROFILE_LEVEL >= 2:
            work.update(self._dense_work())
        return work

    def _dense_work(self):
        """Algorithmic flops / bytes of the hand-written dense ops (csrc/conv2d.hip, csrc/bn2d.hip) of one step, from one
        instrumented forward: every call's shape is recorded at the Python entry of the op."""
        from bevfusion_amd import bn2d as b2, conv2d as c2
        convs, bns, hybrid = [], [], []
        orig_c, orig_b, orig_h = c2._Conv2dFunction.forward, b2._apply, c2._LibConvHipWgradFunction.forward

        def shape(x, weight, stride, pad, dil):
            N, Cin, H, W = x.shape
            Cout, _, KH, KW = weight.shape
            OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
            OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
            pw = KH == 1 and KW == 1 and stride == 1 and pad == 0   # served by conv_pw_kernel (own profiler ids)
            return (N * OH * OW, Cin * KH * KW, Cout, bool(x.requires_grad), N * H * W * Cin), pw

        # every entry: (M, K, Cout, data gradient wanted, input elements, forward on HIP, data gradient on HIP, pointwise)
        def spy_c(ctx, x, weight, bias, stride, pad, dil, emit_stats, dgrad_lib=False, *more):
            sh, pw = shape(x, weight, stride, pad, dil)
            convs.append(sh + (True, not dgrad_lib, pw))
            return orig_c(ctx, x, weight, bias, stride, pad, dil, emit_stats, dgrad_lib, *more)

        orig_s = c2._Conv2dSplitFunction.forward

        def spy_s(ctx, x, weight, bias, stride, pad, dil, emit_stats):
            # fp32 convolution as three bf16 products: one launch per direction with 3x the channels (forward, data gradient)
            # or 3x the batch (weight gradient) -- 3x the flops of the bf16 layer in each
            (M, K, Cout, rg, xin), pw = shape(x, weight, stride, pad, dil)
            convs.append((M, 3 * K, Cout, rg, 3 * xin, True, True, pw))
            return orig_s(ctx, x, weight, bias, stride, pad, dil, emit_stats)

        def spy_b(x, residual, *a, **k):
            if x.is_cuda and x.dim() == 4:
                has_partial = (len(a) > 7 and a[7] is not None) or k.get("partial") is not None
                bns.append((x.numel() * x.element_size(), residual is not None, has_partial))
            return orig_b(x, residual, *a, **k)

        def spy_h(ctx, x, weight, stride, pad, dil, dgrad_hip=False, *more):  # library forward, HIP weight gradient (ResNet-50 trunk)
            sh, pw = shape(x, weight, stride, pad, dil)
            convs.append(sh + (False, bool(dgrad_hip), pw))
            hybrid.append(1)
            return orig_h(ctx, x, weight, stride, pad, dil, dgrad_hip, *more)

        c2._Conv2dFunction.forward, b2._apply = staticmethod(spy_c), spy_b
        c2._LibConvHipWgradFunction.forward = staticmethod(spy_h)
        conv_ext, c2.CONV_EXT = c2.CONV_EXT, False  # the spies sit on the Python Functions: this one pass goes through them
        c2._Conv2dSplitFunction.forward = staticmethod(spy_s)
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.amp):
                self.step_model(self.inputs, None, self.gts)
        finally:
            c2._Conv2dFunction.forward, b2._apply = staticmethod(orig_c), orig_b
            c2._LibConvHipWgradFunction.forward = staticmethod(orig_h)
            c2.CONV_EXT = conv_ext
            c2._Conv2dSplitFunction.forward = staticmethod(orig_s)
        work = {}
        peak, note = MFMA_PEAK_BF16

        def conv_entry(sel, scope):
            rows = [c for c in convs if sel(c)]
            fl = sum(2.0 * c[0] * c[1] * c[2] for c in rows)
            by = sum((c[4] + c[0] * c[2] + c[1] * c[2]) * 2 for c in rows)
            bound = "mfma" if fl / max(by, 1) * HBM_PEAK_GBS * 1e9 > peak * 1e12 else "hbm"
            return dict(bound=bound, flops=fl, bytes=by, unit_peak=peak, peak_note=note, scope="%d launches per step: %s" % (len(rows), scope))

        if convs:
            work["conv2d_fwd"] = conv_entry(lambda c: c[5] and not c[7], "implicit-GEMM forward (conv_igemm_kernel) of the BEV, view-transform, "
                                            "LSS-FPN and ResNet-50 3x3 convolutions")
            work["conv2d_dgrad"] = conv_entry(lambda c: c[3] and c[6] and not c[7], "implicit GEMM in transposed-gather mode (conv_igemm_kernel)")
            work["conv2d_pw_fwd"] = conv_entry(lambda c: c[5] and c[7], "pointwise kernel (conv_pw_kernel): 1x1 stride-1 forward, a GEMM over the "
                                               "pixel matrix with K = 64 ... 3072")
            work["conv2d_pw_dgrad"] = conv_entry(lambda c: c[3] and c[6] and c[7], "pointwise kernel over dy (+ the fused residual-gradient addend)")
            for k in ("conv2d_pw_fwd", "conv2d_pw_dgrad"):
                if not work[k]["flops"]:
                    del work[k]
            work["conv2d_wgrad"] = conv_entry(lambda c: True, "pixel-major LDS tiles, transposing LDS reads, split pixel range + fixed-order slab sum "
                                              "(incl. the %d ResNet-50 layers whose forward stays on the library)" % len(hybrid))
        if bns:
            # forward: statistics pass (unless the producing conv accumulated them) + apply (read, write) [+ residual read];
            # backward: reduce (dy, x) + apply (dy, x -> dx) [+ y read and d_residual write for residual layers]
            work["bn2d_fwd"] = dict(bound="hbm", bytes=sum(b * ((2 if p else 3) + (1 if r else 0)) for b, r, p in bns),
                                    scope="%d BatchNorm(+residual)(+ReLU) layers per step: statistics, finalize, apply" % len(bns))
            # residual layers: + d_residual written, + the ReLU decisions read twice: as one bit per element (1/16 of a pass) where
            # the forward stored the bit mask (statistics from a HIP conv: bn2d.RELU_BITS), else from the saved output
            from bevfusion_amd import bn2d as _b2
            work["bn2d_bwd"] = dict(bound="hbm", bytes=sum(b * (5 + ((1.125 if (p and _b2.RELU_BITS) else 3) if r else 0)) for b, r, p in bns),
                                    scope="%d layers per step: reduce, finalize, apply" % len(bns))
        return work

    def cpu_baseline(self, frames):
        """SURVEY 8(d) CPU baseline on `frames` frame(s) of this workload, all host cores: (i) hard voxelization = the C
        oracle (single thread, as voxelization_cpu.cpp); (ii) bev_pool = restated QuickCumsum in fp32 (the reference's only
        CPU-capable formulation), forward + backward; (iii) the whole 21-layer sparse encoder as gather -> mm -> index_add_,
        forward + backward; (iv) the config-0 forward of the frame(s) (dense layers = torch.nn on the CPU, ops = oracle)
        = BASELINE.json configs[0].  Returns (frames/s of (iv), description, parts in ms)."""
        import copy
        from oracle import cpu_pipeline as cp
        N = self.N
        nf = max(1, min(frames, self.B))
        threads = torch.get_num_threads()
        side, self.model._side_stream = self.model._side_stream, None  # HIP stream handles do not deep-copy
        try:
            model = copy.deepcopy(self.model).float().cpu().train()
        finally:
            self.model._side_stream = side
        pts = self.points_np[:nf]
        parts = {}
        mats = imgs = None
        if self.camera:
            mats = {k: self.inputs[k][:nf].float().cpu().numpy() for k in ("lidar2img", "cam2img", "cam2lidar", "img_aug_matrix",
                                                                             "lidar_aug_matrix")}
            imgs = self.inputs["imgs"][:nf].float().cpu()
        # config-0 forward: CPU_REPEATS timed repeats (default 3), the MEDIAN is the baseline (a single shot on a shared host
        # spread 2x between boxes in round 2); the per-stage timings are medians over the same repeats
        runs, tms = [], []
        for _ in range(CPU_REPEATS):
            tm = {}
            t0 = time.perf_counter()
            with torch.no_grad():
                cp.model_forward(model, pts, imgs, mats, N, timings=tm)
            runs.append(time.perf_counter() - t0)
            tms.append(tm)
        full = float(np.median(runs))
        parts["config0_forward_ms"] = round(full * 1e3 / nf, 1)
        parts["config0_forward_ms_all_repeats"] = [round(r * 1e3 / nf, 1) for r in runs]
        parts.update({"config0_" + k + "_ms": round(float(np.median([t[k] for t in tms])) * 1e3 / nf, 1) for k in tms[0]})
        if self.camera:
            vt = model.view_transform
            gf, kept, ranks, order = cp.bev_geometry(vt, mats, nf)
            x = torch.randn(gf.shape[0], vt.C, requires_grad=True)
            t0 = time.perf_counter()
            out = cp.bev_pool_quickcumsum(x, torch.from_numpy(gf), torch.from_numpy(ranks), nf, vt._nx_host[2], vt._nx_host[0],
                                          vt._nx_host[1])
            t1 = time.perf_counter()
            out.backward(torch.ones_like(out))
            parts["bev_pool_quickcumsum_fwd_ms"] = round((t1 - t0) * 1e3 / nf, 1)
            parts["bev_pool_quickcumsum_bwd_ms"] = round((time.perf_counter() - t1) * 1e3 / nf, 1)
            parts["bev_pool_rows_per_frame"] = int(gf.shape[0] // nf)
        if self.lidar:
            t0 = time.perf_counter()
            vf, coords = cp.voxelize_mean(pts, N)
            parts["voxelize_oracle_1thread_ms"] = round((time.perf_counter() - t0) * 1e3 / nf, 1)
            t0 = time.perf_counter()
            bev = cp.sparse_encoder_forward(model.pts_middle_encoder, vf, coords, nf)
            t1 = time.perf_counter()
            bev.square().mean().backward()
            parts["sparse_encoder_fwd_ms"] = round((t1 - t0) * 1e3 / nf, 1)
            parts["sparse_encoder_bwd_ms"] = round((time.perf_counter() - t1) * 1e3 / nf, 1)
        cpu = "?"
        try:
            cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
        except (OSError, IndexError):
            pass
        what = ("median of %d repeats of %d frame(s), FORWARD ONLY (BASELINE configs[0]): oracle voxelize/rasterise/geometry + "
                "torch.nn dense layers + QuickCumsum bev_pool (fp32) + gather-mm-index_add sparse encoder; per-op parts "
                "(QuickCumsum fwd/bwd, sparse encoder fwd/bwd, 1-thread voxelization) single shot; os.cpu_count()=%s, "
                "torch threads=%d, CPU=%s" % (CPU_REPEATS, nf, os.cpu_count(), threads, cpu))
        return nf / full, what, parts, threads


class LidarOnly(_ModelWorkload):
    """BASELINE configs[1]: hard voxelization + sparse encoder (+ SECOND/FPN/head) fwd+bwd, fp32."""
    camera, lidar, amp = False, True, False
    name = "lidar_only: hard voxelize + BEVFusionSparseEncoder + SECOND/SECONDFPN + TransFusion head fwd+bwd+AdamW, fp32"


class LidarBranch(LidarOnly):
    """BASELINE configs[1] read literally: ONLY hard voxelization + the 4-stage sparse encoder, forward + backward, fp32 --
    no BEV backbone, head or optimizer around it (the gradient of a sum of squares of the BEV map drives the backward)."""
    name = "lidar_branch: hard voxelize + voxel mean + BEVFusionSparseEncoder (21 sparse convs, fused BN1d) fwd+bwd, fp32"

    def step(self):
        for p in self._enc_params:
            p.grad = None
        bev = self.model.extract_pts_feat(self.inputs)
        loss = bev.float().square().mean()
        loss.backward()
        return loss

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._enc_params = list(self.model.pts_middle_encoder.parameters())


class CameraOnly(_ModelWorkload):
    """BASELINE configs[2]: ResNet-50 + LSS depth outer product + bev_pool to 180x180 (+ BEV backbone/head), bf16."""
    camera, lidar, amp = True, False, True
    name = "camera_only: ResNet-50 + LSSFPN + DepthLSSTransform (fused lift-splat) + SECOND/SECONDFPN + head fwd+bwd+AdamW, bf16"


class FullModel(_ModelWorkload):
    """BASELINE configs[3]/[4]: full BEVFusion fwd+bwd, batch 4 per GPU, bf16 with fp32 index paths; DDP over RCCL for N>1."""
    name = ("full: BEVFusion camera+LiDAR (ResNet-50, LSS 6x256x704, hard voxelize 40k pts, sparse encoder, ConvFuser, "
            "SECOND/SECONDFPN, TransFusion head + Hungarian targets + focal/L1/gaussian-focal losses) fwd + bwd + clip + AdamW, bf16 autocast, fp32 index paths")

class DistSelfTest:
    """CPU + gloo rehearsal of the multi-rank plumbing (tests/test_distributed_cpu.py): the dense BEV tail of the
    model (ConvFuser -> SECOND -> SECONDFPN -> head convs, reduced size) under DDP, one batch per rank, no data-path
    collective.  The hand-written HIP ops have no CPU path and are not part of this workload."""

    name = "dist_selftest: reduced ConvFuser+SECOND+SECONDFPN under DDP on CPU (gloo), plumbing rehearsal only"
    amp = False

    def __init__(self, device, batch, points, seed_base=0, ddp=False, local_rank=0):
        import bevfusion_amd  # noqa: F401
        from bevfusion_amd import dense_modules as dm
        torch.manual_seed(0)
        self.B = batch
        self.nk = self.m = None
        self.model = torch.nn.Sequential()
        self.fuser = dm.ConvFuser([8, 16], 16)
        self.backbone = dm.SECOND(16, [16, 32], [1, 1], [1, 2])
        self.neck = dm.SECONDFPN([16, 32], [16, 16], [1, 2], use_conv_for_no_stride=True)
        self.net = torch.nn.ModuleList([self.fuser, self.backbone, self.neck])
        self.n_params = sum(p.numel() for p in self.net.parameters())

        class Tail(torch.nn.Module):
            def __init__(s, fuser, backbone, neck):
                super().__init__()
                s.fuser, s.backbone, s.neck = fuser, backbone, neck

            def forward(s, a, b):
                return s.neck(s.backbone(s.fuser([a, b])))[0]

        self.tail = Tail(self.fuser, self.backbone, self.neck)
        self.step_model = self.tail
        self.grad_sync = None
        if ddp and GRAD_SYNC == "ddp":
            from torch.nn.parallel import DistributedDataParallel as DDP
            self.step_model = DDP(self.tail)
        elif ddp:
            from bevfusion_amd.grad_sync import FlatGradAllReduce, broadcast_parameters
            broadcast_parameters(self.tail)
            self.grad_sync = FlatGradAllReduce(self.tail.parameters())
        self.opt = torch.optim.AdamW(self.tail.parameters(), lr=2e-4)
        g = torch.Generator().manual_seed(seed_base)
        self.a = torch.randn(batch, 8, 24, 24, generator=g)
        self.b = torch.randn(batch, 16, 24, 24, generator=g)

    def step(self):
        self.opt.zero_grad(set_to_none=True)
        loss = self.step_model(self.a, self.b).abs().mean()
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.reduce()
        self.opt.step()
        return loss

    def collect_work(self):
        return {}

    def grad_fingerprint(self):
        return float(sum(p.grad.double().abs().sum() for p in self.tail.parameters() if p.grad is not None))


WORKLOADS = {"dist_selftest": DistSelfTest, "full": FullModel, "lidar_only": LidarOnly, "lidar_branch": LidarBranch, "camera_only": CameraOnly, "hotpath_v1": HotPathV1}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_mode = args.workload == "dist_selftest"
    # stdout carries exactly ONE line (the JSON): native libraries write banners to fd 1 (RCCL prints its version block
    # there on communicator creation), so fd 1 points at stderr for the whole run and the JSON goes to the saved fd
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if not cpu_mode and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # BENCH_ONE_GPU=1 (with --backend gloo): every rank uses cuda:0 and gradients are reduced through gloo -- a rehearsal of the
    # multi-rank code path (rank-dependent inputs, DDP buckets incl. the bf16 ones, max-over-ranks timing) on a one-GPU box
    one_gpu = os.environ.get("BENCH_ONE_GPU", "0") == "1" and args.backend == "gloo"
    if one_gpu:
        local_rank = 0
    if cpu_mode:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    force_ddp = os.environ.get("BENCH_FORCE_DDP", "0") == "1"  # rehearse the DDP wrapper with a 1-rank group
    if world > 1 or force_ddp:
        import torch.distributed as dist
        if force_ddp and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29531"), RANK="0",
                              WORLD_SIZE="1", LOCAL_RANK="0")
        if cpu_mode or one_gpu:
            dist.init_process_group("gloo")
        else:
            # no device_id: binding the group to the device at init ("eager" communicator) costs this host-bound step 4 ms
            # (38.5 plain, 38.8 with the lazily created communicator + flat all-reduce, 42.6 eager, one MI355X, world size 1)
            dist.init_process_group("nccl")
    assert world == args.gpus, "launch with --nproc-per-node == --gpus (WORLD_SIZE=%d, --gpus=%d)" % (world, args.gpus)

    if args.vt_fp32:
        os.environ["BENCH_VT_FP32"] = "1"
        from bevfusion_amd import conv2d as _c2v
        _c2v.FP32_SPLIT = False  # reference numerics throughout: the head's fp32 island on the library's exact fp32 convolution too
    from bevfusion_amd import _lib
    cls = WORKLOADS[args.workload]
    if cpu_mode:
        wl = cls(dev, args.batch, args.points, seed_base=100 * rank, ddp=world > 1, local_rank=local_rank)
        work = {}
    elif issubclass(cls, _ModelWorkload):
        wl = cls(dev, args.batch, args.points, seed_base=100 * rank, ddp=world > 1 or force_ddp, local_rank=local_rank)
        # BENCH_NO_WORK=1 (counter passes under rocprofv3): no instrumented extra forward, so every kernel family runs exactly
        # warm-up + steps times
        work = {} if NO_WORK else wl.collect_work()
    else:
        wl = cls(dev, args.batch, args.points, seed_base=100 * rank)
        work = {"bev_pool_fwd": dict(bound="hbm", bytes=wl.dominant_bytes())}

    def barrier():
        if dist is not None:
            if cpu_mode or one_gpu:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])
        if not cpu_mode:
            torch.cuda.synchronize()

    first_loss = None
    # settling steps BEFORE the W warm-up steps of the contract (untimed initialisation, like building the model): the first steps of
    # a process pick MIOpen solutions, load code objects and grow the caching allocator and the library's workspaces; with a small
    # --warmup from the caller they would otherwise reach into the timed region.  BENCH_PREWARM_STEPS=0 switches them off.
    settle_steps = 0
    for i in range(0 if cpu_mode else PREWARM_STEPS):
        r = wl.step()
        settle_steps += 1
        if first_loss is None and torch.is_tensor(r):
            first_loss = r.detach()
    if not cpu_mode and PREWARM_STEPS > 0 and SETTLE_MAX_S > 0:
        # ... and until the step time has settled: blocks of 10 untimed steps until a block is within 5 % of the best block seen
        # (at least two blocks, at most SETTLE_MAX_S seconds).  The first process on a freshly provisioned box has been seen to run
        # its first seconds at 1.3-2x the steady step time with normal per-kernel GPU times (the host is still paging the image in):
        # `config.settle_steps` says how long this took
        t_settle, best_blk, blocks = time.perf_counter(), None, 0
        while dist is not None or time.perf_counter() - t_settle < SETTLE_MAX_S:
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(10):
                wl.step()
            torch.cuda.synchronize()
            blk = time.perf_counter() - tb
            settle_steps += 10
            blocks += 1
            if dist is not None:
                if blocks >= 2:   # several ranks: every step holds a collective, so every rank runs the same fixed two blocks
                    break
                continue
            if best_blk is not None and blocks >= 2 and blk <= 1.05 * best_blk:
                break
            best_blk = blk if best_blk is None else min(best_blk, blk)
    for i in range(args.warmup):
        r = wl.step()
        if first_loss is None and torch.is_tensor(r):
            first_loss = r.detach()
    barrier()
    graphed = bool(getattr(wl, "use_graph", False))
    if not cpu_mode and not graphed:
        _lib.profile_enable(PROFILE_LEVEL)
        for op in _lib.OPS:
            _lib.profile_read(op, reset=True)
    t0 = time.perf_counter()
    issue_t = [t0]  # host time at which each step's launches had been issued (no sync): shows a disturbed run (median << mean)
    last_loss = None
    dense_steps = 0
    for i in range(args.steps):
        if not cpu_mode and not graphed and PROFILE_LEVEL >= 2:
            # the dense ops add ~280 event pairs (1.8 ms of host time) to a step: they are timed on every DENSE_EVERY-th step of
            # the timed region only, the sparse / lift-splat ops on all of them
            dense = i % DENSE_EVERY == 0
            _lib.profile_enable(2 if dense else 1)
            dense_steps += dense
        last_loss = wl.step()
        issue_t.append(time.perf_counter())
    barrier()
    dt = time.perf_counter() - t0
    prof = {}
    prof_steps = args.steps
    if graphed and not cpu_mode:
        # a replayed graph does not pass through the library's event scopes: the per-op times of the roofline objects
        # come from a few EAGER steps of the same workload after the timed region (same kernels, same sizes)
        prof_steps = 5
        torch.cuda.synchronize()
        _lib.profile_enable(PROFILE_LEVEL)
        for op in _lib.OPS:
            _lib.profile_read(op, reset=True)
        for _ in range(prof_steps):
            wl._eager_step()
        dense_steps = prof_steps
        torch.cuda.synchronize()
    iso, iso_steps = {}, 0
    if not cpu_mode:
        _lib.profile_enable(False)
        prof = {op: _lib.profile_read(op, reset=True) for op in _lib.OPS}
        model = getattr(wl, "model", None)
        from bevfusion_amd import conv2d as _c2
        overlapped = bool(model is not None and (getattr(model, "lidar_side_stream", False) or _c2.WGRAD_SIDE_STREAM))
        # single process only: a step contains the gradient exchange, so with several ranks every rank would have to run it
        if ISOLATED_STEPS > 0 and not graphed and overlapped and dist is None:
            # kernel-time pass: the same workload on ONE queue (LiDAR branch and weight gradients in line), every op timed.
            # Outside the timed region.
            side_lidar, side_wgrad = model.lidar_side_stream, _c2.WGRAD_SIDE_STREAM
            model.lidar_side_stream, _c2.WGRAD_SIDE_STREAM = False, False
            torch.cuda.synchronize()
            _lib.profile_enable(PROFILE_LEVEL)
            for _ in range(ISOLATED_STEPS):
                wl.step()
            torch.cuda.synchronize()
            _lib.profile_enable(False)
            iso = {op: _lib.profile_read(op, reset=True) for op in _lib.OPS}
            iso_steps = ISOLATED_STEPS
            model.lidar_side_stream, _c2.WGRAD_SIDE_STREAM = side_lidar, side_wgrad
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    frames = args.batch * world * args.steps
    value = frames / dt
    # the collective library that actually carried the gradients ("nccl" is RCCL on ROCm), not an assumption
    backend_name = {"nccl": "RCCL"}.get(dist.get_backend(), dist.get_backend()) if dist is not None else "none"

    if rank == 0:
        ops = {op: {"ms_per_step": round(ms / (max(dense_steps, 1) if op in DENSE_OPS else prof_steps), 4),
                    "launches_per_step": cnt / (max(dense_steps, 1) if op in DENSE_OPS else prof_steps)}
               for op, (ms, cnt) in prof.items() if cnt}
        # dominant hand-written op of the step = the one with the largest accumulated event time
        def op_ms(o):  # one-queue kernel time where it was measured, else the event time of the timed region
            if iso.get(o, (0, 0))[1]:
                return iso[o][0] / iso_steps
            return prof[o][0] / (max(dense_steps, 1) if o in DENSE_OPS else prof_steps)

        dom = max((op for op in work if prof.get(op, (0, 0))[1] and op != "spconv_wgrad_main"), key=op_ms, default=None)
        # HBM traffic: PMC counters cannot be read from inside this process; the value is the rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE measurement committed under profiles/ (same batch-4 sizes), tagged with its source
        pmc, pmc_src = {}, None
        for cand in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))["kernels"]
                pmc_src = "profiles/" + cand + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of the same workload, per step; correction per op in that file's hbm_bytes_note; not measured by this run)"
                break
            except (OSError, KeyError, ValueError):
                continue

        def steps_of(op):  # timed steps the op's events cover
            return max(dense_steps, 1) if op in DENSE_OPS else prof_steps

        def roof_of(op):
            ms, cnt = prof[op]
            w = work[op]
            n_steps = steps_of(op)
            sec_per_step = ms * 1e-3 / n_steps
            traffic = int(pmc[op]["hbm_bytes"]) if (op in pmc and args.batch == 4 and args.points == 40000) else None
            # a counter figure far BELOW the algorithmic bytes is a mis-attributed kernel family, not evidence: null
            if traffic is not None and traffic < 0.5 * w["bytes"]:
                traffic = None
            r = {"kernel": op, "bound": w["bound"], "ms_per_step": round(ms / n_steps, 5), "launches_per_step": cnt / n_steps,
                 "timed_steps_sampled": n_steps,
                 "traffic": traffic, "traffic_source": pmc_src if traffic is not None else None}
            if iso.get(op, (0, 0))[1]:
                # the step runs on three HIP queues (camera / BEV chain, LiDAR branch, weight gradients): `ms_per_step` above is
                # the op's event time in the timed region and includes sharing the chip with the other queues; every figure
                # below is computed from the one-queue pass, which a kernel trace reproduces
                r["event_ms_per_step_in_timed_region"] = r["ms_per_step"]
                r["kernel_ms_per_step"] = round(iso[op][0] / iso_steps, 5)
                r["kernel_time_source"] = "%d steps after the timed region on one queue (no side streams)" % iso_steps
                sec_per_step = iso[op][0] * 1e-3 / iso_steps
            if "scope" in w:
                r["scope"] = w["scope"]
            if w["bound"] == "hbm":
                ach = w["bytes"] / sec_per_step / 1e9
                r.update(achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                         algorithmic_bytes_per_step=int(w["bytes"]))
                if "flops" in w:  # an op with matrix work on the HBM side of the ridge: the other ceiling, for reference
                    tf = w["flops"] / sec_per_step / 1e12
                    r.update(algorithmic_flops_per_step=w["flops"], tflops=round(tf, 2), mfma_frac=round(tf / w["unit_peak"], 4),
                             peak_note=w.get("peak_note", ""))
            else:
                ach = w["flops"] / sec_per_step / 1e12
                r.update(achieved=round(ach, 2), peak=w["unit_peak"], unit="TFLOP/s", frac=round(ach / w["unit_peak"], 4),
                         algorithmic_flops_per_step=w["flops"], algorithmic_bytes_per_step=int(w["bytes"]),
                         peak_note=w.get("peak_note", ""))
                r["hbm_frac_on_algorithmic_bytes"] = round(w["bytes"] / sec_per_step / 1e9 / HBM_PEAK_GBS, 4)
            return r

        roofline_ops = [roof_of(op) for op in work if prof.get(op, (0, 0))[1]]
        roof = roof_of(dom) if dom is not None else None
        line = {
            "metric": "nuScenes frames/sec (6-cam+LiDAR BEVFusion fwd+bwd)" if args.workload == "full" else
                      "nuScenes frames/sec (%s)" % args.workload,
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "step_issue_ms": _issue_stats(issue_t), "settle_steps": settle_steps,
            "vs_baseline": None, "dtype": "bf16" if getattr(wl, "amp", False) else "f32", "data": "synthetic",
            "config": {"workload": wl.name, "frames_per_gpu_per_step": args.batch, "global_batch": args.batch * world,
                       "points_per_frame": args.points, "frustum_rows_kept": wl.nk, "bev_intervals": wl.m,
                       "parallelism": ("dp%d (%s)" % (world, ("torch DDP buckets, %s all-reduce" % backend_name) if GRAD_SYNC == "ddp" else
                                                      "flat all-reduce: one gradient exchange per dtype after the backward over %s, form %s"
                                                      % (backend_name, getattr(getattr(wl, "grad_sync", None), "exchange", "all-reduce")))) if world > 1 and hasattr(wl, "step_model")
                       else "independent frames per rank"},
            "roofline": roof, "roofline_ops": roofline_ops, "ops": ops,
        }
        if hasattr(wl, "use_graph"):
            line["config"]["execution"] = ("whole step captured once into a hipGraph and replayed (per-op roofline times from %d eager "
                                           "steps after the timed region)" % prof_steps) if graphed else (
                                               "eager launches on %d HIP queues (camera / BEV chain%s%s)" % (
                                                   1 + int(bool(getattr(wl.model, "lidar_side_stream", False))) + int(wl._wgrad_join is not None),
                                                   ", LiDAR branch" if getattr(wl.model, "lidar_side_stream", False) else "",
                                                   ", weight gradients" if wl._wgrad_join is not None else ""))
        if hasattr(wl, "vt_bf16"):
            line["config"]["view_transform_conv_dtype"] = "bf16" if wl.vt_bf16 else "fp32 (reference fp32 island)"
            from bevfusion_amd import conv2d as _c2i
            line["config"]["head_fp32_conv"] = ("three bf16 products per multiply, fp32 accumulation (2^-16 relative per product)"
                                                if _c2i.FP32_SPLIT else "library fp32")
        if hasattr(wl, "n_params"):
            line["config"]["dense_weight_gradients"] = ("one grouped launch per tile shape at the end of the backward pass"
                                                        if getattr(wl, "wgrad_grouped", False) else "one launch pair per layer")
            line["config"]["backward_pass_thread"] = "caller" if getattr(wl, "backward_on_caller", False) else "autograd engine device thread"
            line["config"]["trainable_params"] = wl.n_params
        if torch.is_tensor(first_loss) and torch.is_tensor(last_loss):
            # the optimizer really steps: total loss of the (fixed) batch at the first warm-up step and at the last timed step
            line["config"]["loss_first_step"] = round(float(first_loss), 4)
            line["config"]["loss_last_step"] = round(float(last_loss.detach()), 4)
        if cpu_mode:
            line["data"] = "synthetic (CPU plumbing rehearsal, not a performance number)"
            line["config"]["grad_fingerprint"] = wl.grad_fingerprint()
        if (world == 1 and not cpu_mode and args.workload == "full" and not args.vt_fp32 and getattr(wl, "vt_bf16", False)
                and os.environ.get("BENCH_REFERENCE_NUMERICS", "1") == "1" and not NO_WORK):
            # the same step with the view transform's conv stacks in fp32 with fp32 weights = the reference's fp32 island
            # (BF/bevfusion.py:177): a second timed region of the same K steps in the same invocation, so that the driver's
            # line carries both numbers (`value` stays the bf16-conv-stack configuration BASELINE configs[3] names)
            del wl
            torch.cuda.empty_cache()
            os.environ["BENCH_VT_FP32"] = "1"
            from bevfusion_amd import conv2d as _c2
            split_was, _c2.FP32_SPLIT = _c2.FP32_SPLIT, False  # the head's fp32 island on the library's exact fp32 convolution
            wl = cls(dev, args.batch, args.points, seed_base=100 * rank, ddp=False, local_rank=local_rank)
            os.environ["BENCH_VT_FP32"] = "0"
            for _ in range(args.warmup):
                wl.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                wl.step()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            line["value_reference_numerics"] = round(args.batch * args.steps / dt2, 3)
            line["ms_per_step_reference_numerics"] = round(dt2 / args.steps * 1e3, 4)
            line["config"]["reference_numerics_region"] = ("second timed region, same steps / warm-up: view-transform conv stacks "
                                                           "(dtransform, depthnet, downsample) in fp32 with fp32 weights, the heat-map "
                                                           "head's fp32 convolution in exact fp32 (library) instead of three bf16 products")
            _c2.FP32_SPLIT = split_was
        if world == 1 and not args.no_cpu_baseline and not cpu_mode:
            res = wl.cpu_baseline(args.cpu_frames)
            line["cpu_baseline"] = {"value": round(res[0], 4), "unit": "frames/s", "cores": res[3] if len(res) > 3 else 1,
                                    "kind": "port", "sample": res[1]}
            if len(res) > 2:
                line["cpu_baseline"]["unit"] = "frames/s (forward only)"
                line["cpu_baseline"]["parts_ms_per_frame"] = res[2]
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if cpu_mode and os.environ.get("BENCH_FINGERPRINT_DIR"):  # every rank: DDP must leave identical gradients
        with open(os.path.join(os.environ["BENCH_FINGERPRINT_DIR"], "rank%d.txt" % rank), "w") as fh:
            fh.write(repr(wl.grad_fingerprint()))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
