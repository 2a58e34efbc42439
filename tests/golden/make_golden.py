#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE itself.  Run in the build container only
(needs /root/reference); the GPU box and the test-suite only read the committed .npz files.

What is executed from the reference:
  (A) its CPU voxelization extension, compiled unmodified from the sources in place by
      oracle/build_ref.sh  ->  hard_voxelize / dynamic_voxelize outputs
  (B) projects/BEVFusion/bevfusion/depth_lss.py, loaded BY PATH as python source.  Its two
      imports that are not installable here are satisfied with inert placeholders:
      `mmdet3d.registry.MODELS.register_module()` -> identity decorator (registration only),
      `.ops.bev_pool` -> the CPU oracle (the reference's op is CUDA-only).  Everything that is
      recorded as "reference output" below (frustum, geometry, bev_pool_aux cells / kept /
      ranks, the python glue around the op) is computed by the reference's own code.

Only data is stored: inputs are regenerated from seeds by the synthetic module (their sha256 is
stored to detect drift), outputs are stored in full when small and as sha256 + statistics
when large.
"""
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))

import bevfusion_amd  # noqa: E402,F401
from bevfusion_amd import synthetic  # noqa: E402
import oracle  # noqa: E402


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


# --------------------------------------------------------------------------------------- (A)
def ref_hard_voxelize(v, pts, voxel_size, rng, max_points, max_voxels):
    p = torch.from_numpy(pts)
    voxels = p.new_zeros((max_voxels, max_points, p.shape[1]))
    coors = p.new_zeros((max_voxels, 3), dtype=torch.int)
    num = p.new_zeros((max_voxels,), dtype=torch.int)
    n = v.hard_voxelize(p, voxels, coors, num, list(voxel_size), list(rng), max_points, max_voxels, 3, True)
    return voxels[:n].numpy(), coors[:n].numpy(), num[:n].numpy()


def ref_dynamic_voxelize(v, pts, voxel_size, rng):
    p = torch.from_numpy(pts)
    coors = p.new_zeros((p.shape[0], 3), dtype=torch.int)
    v.dynamic_voxelize(p, coors, list(voxel_size), list(rng), 3)
    return coors.numpy()


def make_voxel_goldens():
    import voxel_layer_ref as v
    out = {}
    # A1: the reference's only known-answer test (tests/test_models/test_task_modules/test_voxel/
    # test_voxel_generator.py:7-20): seed 0, 20 uniform points, expected coors (zyx) / counts.
    np.random.seed(0)
    kat_pts = np.random.uniform(0, 4, (20, 3)).astype(np.float32)
    vox, coors, num = ref_hard_voxelize(v, kat_pts, [5, 5, 1], [0, 0, 0, 20, 40, 4], 5, 20)
    assert (coors[:, ::-1] == np.array([[2, 0, 0], [3, 0, 0], [0, 0, 0], [1, 0, 0]])).all()
    assert (num == np.array([5, 5, 5, 3])).all()
    out.update(kat_points=kat_pts, kat_voxels=vox, kat_coors=coors, kat_num=num)

    # A2: cubic grid (where the reference's table indexing bug is harmless): 40k sweep
    pts = synthetic.lidar_sweep(40000, seed=1000)
    rng, vs = [-40.0, -40.0, -40.0, 40.0, 40.0, 40.0], [1.0, 1.0, 1.0]
    vox, coors, num = ref_hard_voxelize(v, pts, vs, rng, 10, 20000)
    out.update(cubic_in_sha=sha(pts), cubic_coors=coors, cubic_num=num, cubic_voxels_sha=sha(vox),
               cubic_voxels_head=vox[:64])
    # A3: cubic grid with a binding max_voxels cap and small max_points
    vox, coors, num = ref_hard_voxelize(v, pts, [0.5, 0.5, 0.5], rng, 3, 3000)
    out.update(cap_coors=coors, cap_num=num, cap_voxels_sha=sha(vox))
    # A4: uniform stress points on a cubic grid (M ~ N), includes out-of-range points
    upts = synthetic.uniform_points(20000, seed=7, rng_range=(-20, -20, -20, 20, 20, 20), margin=2.0)
    vox, coors, num = ref_hard_voxelize(v, upts, [0.5, 0.5, 0.5], [-20, -20, -20, 20, 20, 20], 10, 30000)
    out.update(uni_in_sha=sha(upts), uni_coors=coors, uni_num=num, uni_voxels_sha=sha(vox))
    # A5: dynamic voxelization at the real nuScenes grid (the reference's CPU path is fine here)
    N = synthetic.NUSC
    dc = ref_dynamic_voxelize(v, pts, N["voxel_size"], N["point_cloud_range"])
    out.update(dyn_nusc_coors=dc)
    upts2 = synthetic.uniform_points(40000, seed=11)
    out.update(dyn_uni_in_sha=sha(upts2),
               dyn_uni_coors=ref_dynamic_voxelize(v, upts2, N["voxel_size"], N["point_cloud_range"]))
    np.savez_compressed(os.path.join(HERE, "voxelization_ref.npz"), **out)
    print("voxelization_ref.npz:", {k: getattr(val, "shape", val) for k, val in out.items()})


# --------------------------------------------------------------------------------------- (B)
def load_reference_depth_lss():
    def oracle_bev_pool(feats, coords, ranks, B, D, H, W, is_training):
        starts, lengths = oracle.intervals_from_ranks(ranks.numpy())
        out = oracle.bev_pool_fwd(feats.numpy(), coords.int().numpy(), starts, lengths, int(B), int(D), int(H), int(W))
        return torch.from_numpy(out).permute(0, 4, 1, 2, 3).contiguous()

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    m3d = types.ModuleType("mmdet3d")
    reg = types.ModuleType("mmdet3d.registry")
    reg.MODELS = _Registry()
    m3d.registry = reg
    sys.modules.setdefault("mmdet3d", m3d)
    sys.modules.setdefault("mmdet3d.registry", reg)
    pkg = types.ModuleType("refbev")
    pkg.__path__ = []
    ops = types.ModuleType("refbev.ops")
    ops.bev_pool = oracle_bev_pool
    sys.modules["refbev"] = pkg
    sys.modules["refbev.ops"] = ops
    path = os.path.join(REF, "projects/BEVFusion/bevfusion/depth_lss.py")
    spec = importlib.util.spec_from_file_location("refbev.depth_lss", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refbev.depth_lss"] = mod
    spec.loader.exec_module(mod)
    return mod


def geometry_inputs(rig):
    """The tensors BaseViewTransform.forward derives before get_geometry (depth_lss.py:240-262)."""
    t = {k: torch.from_numpy(v) for k, v in rig.items()}
    intrins = t["camera_intrinsics"][..., :3, :3]
    post_rots = t["img_aug_matrix"][..., :3, :3]
    post_trans = t["img_aug_matrix"][..., :3, 3]
    c2l_rots = t["camera2lidar"][..., :3, :3]
    c2l_trans = t["camera2lidar"][..., :3, 3]
    extra_rots = t["lidar_aug_matrix"][..., :3, :3]
    extra_trans = t["lidar_aug_matrix"][..., :3, 3]
    return dict(c2l_rots=c2l_rots, c2l_trans=c2l_trans, intrins_inv=torch.inverse(intrins),
                post_rots_inv=torch.inverse(post_rots), post_trans=post_trans, extra_rots=extra_rots,
                extra_trans=extra_trans)


def make_lss_goldens():
    mod = load_reference_depth_lss()
    N = synthetic.NUSC
    out = {}
    # B1: tiny view transform, 2 samples x 6 cameras, train-time lidar augmentation
    tiny = dict(in_channels=16, out_channels=8, image_size=(64, 176), feature_size=(8, 22),
                xbound=[-54.0, 54.0, 1.2], ybound=[-54.0, 54.0, 1.2], zbound=[-10.0, 10.0, 20.0],
                dbound=[1.0, 61.0, 3.0])
    vt = mod.LSSTransform(**tiny)
    vt.eval()
    rig = synthetic.camera_rig(batch=2, seed=5, train_aug=True)
    # scale the image aug to the tiny image: 1600x900 -> 0.12 -> 192x108, crop to 176x64
    rig["img_aug_matrix"][..., 0, 0] = 0.12
    rig["img_aug_matrix"][..., 1, 1] = 0.12
    rig["img_aug_matrix"][..., 0, 3] = -8.0
    rig["img_aug_matrix"][..., 1, 3] = -44.0
    gi = geometry_inputs(rig)
    with torch.no_grad():
        geom = vt.get_geometry(gi["c2l_rots"], gi["c2l_trans"], gi["intrins_inv"], gi["post_rots_inv"],
                               gi["post_trans"], extra_rots=gi["extra_rots"], extra_trans=gi["extra_trans"])
        geom_feats, kept, ranks, indices = vt.bev_pool_aux(geom)
        g = torch.Generator().manual_seed(123)
        B, Ncam, D, fH, fW = geom.shape[:5]
        x = torch.randn(B, Ncam, D, fH, fW, tiny["out_channels"], generator=g)
        bev = vt.bev_pool(x, geom)   # reference glue + oracle op
    out.update(tiny_frustum=vt.frustum.detach().numpy(), tiny_geom=geom.numpy(),
               tiny_geom_feats=geom_feats.numpy().astype(np.int32), tiny_kept=kept.numpy(),
               tiny_ranks=ranks.numpy(), tiny_bev=bev.numpy(), tiny_x_sha=sha(x.numpy()),
               tiny_dx=vt.dx.detach().numpy(), tiny_bx=vt.bx.detach().numpy(), tiny_nx=vt.nx.detach().numpy())
    for k, v in rig.items():
        out["tiny_rig_" + k] = v

    # B2: full nuScenes-size geometry, eval aug, B=1: hashes + statistics only
    full = dict(in_channels=256, out_channels=80, image_size=N["image_size"], feature_size=N["feature_size"],
                xbound=N["xbound"], ybound=N["ybound"], zbound=N["zbound"], dbound=N["dbound"])
    vt = mod.LSSTransform(**full)
    rig = synthetic.camera_rig(batch=1)
    gi = geometry_inputs(rig)
    with torch.no_grad():
        geom = vt.get_geometry(gi["c2l_rots"], gi["c2l_trans"], gi["intrins_inv"], gi["post_rots_inv"],
                               gi["post_trans"], extra_rots=gi["extra_rots"], extra_trans=gi["extra_trans"])
        geom_feats, kept, ranks, indices = vt.bev_pool_aux(geom)
    starts, lengths = oracle.intervals_from_ranks(ranks.numpy())
    out.update(full_frustum_sha=sha(vt.frustum.detach().numpy()), full_geom_sha=sha(geom.numpy()),
               full_geom_sample=geom.numpy().reshape(-1, 3)[::997].copy(),
               full_kept_sha=sha(kept.numpy()), full_ranks_sha=sha(ranks.numpy()),
               full_geom_feats_sha=sha(geom_feats.numpy().astype(np.int32)),
               full_counts=np.array([kept.numel(), int(kept.sum()), len(starts), int(lengths.max()),
                                     int(np.median(lengths))], np.int64),
               full_dx=vt.dx.detach().numpy(), full_bx=vt.bx.detach().numpy(), full_nx=vt.nx.detach().numpy())
    # B3: the reference's sparse-depth rasteriser (BaseDepthTransform.forward :372-449) and GT depth histogram
    # (DepthLSSTransform.get_cam_feats :636-686), tiny rig, 2 samples.  The rasteriser lives inside forward(); the
    # depth images are captured at its call of get_cam_feats.
    dcfg = dict(tiny, in_channels=16, out_channels=8)
    dvt = mod.DepthLSSTransform(**dcfg)
    dvt.eval()
    rig = synthetic.camera_rig(batch=2, seed=5, train_aug=True)
    rig["img_aug_matrix"][..., 0, 0] = 0.12
    rig["img_aug_matrix"][..., 1, 1] = 0.12
    rig["img_aug_matrix"][..., 0, 3] = -8.0
    rig["img_aug_matrix"][..., 1, 3] = -44.0
    t = {k: torch.from_numpy(v) for k, v in rig.items()}
    pts = [torch.from_numpy(synthetic.lidar_sweep(6000, seed=300 + i)[:, :5].copy()) for i in range(2)]
    captured = {}

    class _Stop(Exception):
        pass

    real_get_cam_feats = dvt.get_cam_feats

    def capture(img, depth):
        captured["depth"] = depth.clone()
        raise _Stop()

    dvt.get_cam_feats = capture
    img = torch.zeros(2, 6, 16, 8, 22)
    try:
        with torch.no_grad():
            dvt(img, [p.clone() for p in pts], t["lidar2image"], t["camera_intrinsics"], t["camera2lidar"],
                t["img_aug_matrix"], t["lidar_aug_matrix"], None, None, None, None, None)
    except _Stop:
        pass
    with torch.no_grad():
        _, est, gt_distr, counts3d = real_get_cam_feats(img, captured["depth"])
    out.update(rast_depth=captured["depth"].numpy(), rast_gt_distr=gt_distr.numpy(), rast_counts=counts3d.numpy(),
               rast_points_sha=sha(np.stack([p.numpy() for p in pts])))
    for k, v in rig.items():
        out["rast_rig_" + k] = v
    np.savez_compressed(os.path.join(HERE, "lss_ref.npz"), **out)
    print("lss_ref.npz: counts (N', kept, intervals, max len, median len) =", out["full_counts"])


# --------------------------------------------------------------------------------------- (C)
def load_reference_head_utils():
    """(C) mmdet3d/models/utils/gaussian.py loads as is (numpy + torch only).  projects/BEVFusion/bevfusion/utils.py is
    loaded BY PATH; its imports that are not installable here get inert placeholders: registry decorators -> identity,
    `BaseBBoxCoder` / `BaseAssigner` / `AssignResult` / `InstanceData` -> empty base classes.  Only the reference's own
    pure-torch methods are executed: TransFusionBBoxCoder.encode / .decode, BBoxBEVL1Cost.__call__, IoU3DCost.__call__."""
    gpath = os.path.join(REF, "mmdet3d/models/utils/gaussian.py")
    spec = importlib.util.spec_from_file_location("ref_gaussian", gpath)
    gauss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gauss)

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    class _Empty:
        def __init__(self, *a, **k):
            pass

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("mmdet"); mod("mmdet.models")
    mod("mmdet.models.task_modules", AssignResult=_Empty, BaseAssigner=_Empty, BaseBBoxCoder=_Empty)
    mod("mmdet3d"); mod("mmdet3d.registry", TASK_UTILS=_Registry(), MODELS=_Registry())
    mod("mmengine"); mod("mmengine.structures", InstanceData=_Empty)
    upath = os.path.join(REF, "projects/BEVFusion/bevfusion/utils.py")
    spec = importlib.util.spec_from_file_location("ref_bevfusion_utils", upath)
    utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(utils)
    return gauss, utils


def make_head_goldens():
    gauss, utils = load_reference_head_utils()
    N = synthetic.NUSC
    out = {}
    pc, vs, osf = N["point_cloud_range"], N["voxel_size"], 8
    coder = utils.TransFusionBBoxCoder(pc_range=pc[:2], out_size_factor=osf, voxel_size=vs[:2],
                                       post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.0,
                                       code_size=10)
    boxes, labels = synthetic.gt_boxes(seed=3000, n=40)
    out.update(gt_in_sha=sha(boxes), labels_in_sha=sha(labels))
    # C1: encode (BF/utils.py:33-46)
    out["encode"] = coder.encode(torch.from_numpy(boxes)).numpy()
    # C2: decode (BF/utils.py:48-96) on seeded raw head outputs, B=2, P=50
    g = torch.Generator().manual_seed(77)
    B, P = 2, 50
    heat = torch.randn(B, 10, P, generator=g)
    rot = torch.randn(B, 2, P, generator=g)
    dim = torch.randn(B, 3, P, generator=g) * 0.5
    center = torch.rand(B, 2, P, generator=g) * 180
    height = torch.randn(B, 1, P, generator=g)
    vel = torch.randn(B, 2, P, generator=g)
    out.update(dec_heat=heat.numpy(), dec_rot=rot.numpy(), dec_dim=dim.numpy(), dec_center=center.numpy(),
               dec_height=height.numpy(), dec_vel=vel.numpy())
    dec = coder.decode(heat.clone(), rot.clone(), dim.clone(), center.clone(), height.clone(), vel.clone())
    out["dec_boxes"] = torch.stack([d["bboxes"] for d in dec]).numpy()
    out["dec_scores"] = torch.stack([d["scores"] for d in dec]).numpy()
    out["dec_labels"] = torch.stack([d["labels"] for d in dec]).numpy()
    # C3: BBoxBEVL1Cost (BF/utils.py:133-140) and IoU3DCost (:149-151)
    train_cfg = dict(point_cloud_range=pc)
    out["l1_cost"] = utils.BBoxBEVL1Cost(0.25)(dec[0]["bboxes"], torch.from_numpy(boxes), train_cfg).numpy()
    out["iou_cost_of_half"] = utils.IoU3DCost(0.25)(torch.full((2, 3), 0.5)).numpy()
    # C4: gaussian_radius (gaussian.py:62-92) on fp32 0-dim tensors, as BF/bevfusion_head.py:644-650 calls it
    hw = np.stack([np.random.RandomState(5).uniform(0.3, 22, 64), np.random.RandomState(6).uniform(0.3, 22, 64)], 1).astype(np.float32)
    out["radius_hw"] = hw
    out["radius"] = np.array([float(gauss.gaussian_radius((torch.tensor(h), torch.tensor(w)), min_overlap=0.1)) for h, w in hw],
                             np.float32)
    # C5: dense heat-map target: the reference's gaussian_radius + draw_heatmap_gaussian driven by a replay of the loop
    #     BF/bevfusion_head.py:636-662 (the loop is a method of the head class, which needs mmdet to construct)
    gt = torch.from_numpy(boxes)
    grid_size = torch.tensor([1440, 1440, 41])
    pc_range, voxel_size = torch.tensor(pc), torch.tensor(vs)
    fms = grid_size[:2] // osf
    heatmap = gt.new_zeros(10, int(fms[1]), int(fms[0]))
    radii = []
    for idx in range(len(gt)):
        width = gt[idx][3] / voxel_size[0] / osf
        length = gt[idx][4] / voxel_size[1] / osf
        if width > 0 and length > 0:
            radius = gauss.gaussian_radius((length, width), min_overlap=0.1)
            radius = max(2, int(radius))
            radii.append(radius)
            coor_x = (gt[idx][0] - pc_range[0]) / voxel_size[0] / osf
            coor_y = (gt[idx][1] - pc_range[1]) / voxel_size[1] / osf
            center_int = torch.tensor([coor_x, coor_y], dtype=torch.float32).to(torch.int32)
            gauss.draw_heatmap_gaussian(heatmap[labels[idx]], center_int[[1, 0]], radius)
    out["heatmap"] = heatmap.numpy()
    out["heatmap_radii"] = np.array(radii, np.int32)
    np.savez_compressed(os.path.join(HERE, "head_ref.npz"), **out)
    print("head_ref.npz:", {k: getattr(val, "shape", val) for k, val in out.items()})


# --------------------------------------------------------------------------------------- (D)
def load_reference_bev_pool():
    """(D) projects/BEVFusion/bevfusion/ops/bev_pool/bev_pool.py loaded BY PATH.  Its line 4 (`from . import bev_pool_ext`,
    the CUDA-only pybind module) is satisfied by a recording placeholder: `bev_pool_forward` / `bev_pool_backward` store
    the arguments the reference's python hands to its kernel and return zeros.  `QuickCumsum` (bev_pool.py:7-34), the
    reference's only CPU-capable formulation of the pooled sum, is pure torch and runs as is."""
    rec = {}
    ext = types.ModuleType("refbevpool.bev_pool_ext")

    def bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, B, D, H, W):
        rec.update(fwd_lengths=interval_lengths.clone(), fwd_starts=interval_starts.clone(), fwd_geom=geom_feats.clone())
        return x.new_zeros((B, D, H, W, x.shape[1]))

    def bev_pool_backward(out_grad, geom_feats, interval_lengths, interval_starts, B, D, H, W):
        rec.update(bwd_lengths=interval_lengths.clone(), bwd_starts=interval_starts.clone())
        return out_grad.new_zeros((geom_feats.shape[0], out_grad.shape[-1]))

    ext.bev_pool_forward, ext.bev_pool_backward = bev_pool_forward, bev_pool_backward
    pkg = types.ModuleType("refbevpool")
    pkg.__path__ = []
    pkg.bev_pool_ext = ext
    sys.modules["refbevpool"] = pkg
    sys.modules["refbevpool.bev_pool_ext"] = ext
    path = os.path.join(REF, "projects/BEVFusion/bevfusion/ops/bev_pool/bev_pool.py")
    spec = importlib.util.spec_from_file_location("refbevpool.bev_pool", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refbevpool.bev_pool"] = mod
    spec.loader.exec_module(mod)
    return mod, rec


def make_bev_pool_goldens():
    """Outputs of the REFERENCE's own `QuickCumsum` (evaluated in fp64, where its prefix-sum cancellation is exact to
    ~1e-16) and the interval tables its `QuickCumsumTrainingCuda` hands to the kernel."""
    mod, rec = load_reference_bev_pool()
    out = {}
    for name, (seed, n, C, B, D, H, W, integer) in dict(
            int=(41, 24000, 80, 2, 1, 32, 32, True), flt=(42, 40000, 80, 2, 1, 48, 48, False),
            odd=(43, 5000, 8, 1, 2, 16, 16, False)).items():
        x, geom, ranks = synthetic.bev_pool_case(seed, n, C, B, D, H, W, integer)
        xt = torch.from_numpy(x).double().requires_grad_(True)
        y, g = mod.QuickCumsum.apply(xt, torch.from_numpy(geom), torch.from_numpy(ranks))
        # dense [B, D, H, W, C] exactly as the reference's LSS code scatters QuickCumsum's rows (depth_lss.py bev_pool
        # with QuickCumsum: final[geom[:,3], :, geom[:,2], geom[:,0], geom[:,1]] = x)
        dense = torch.zeros(B, D, H, W, C, dtype=torch.float64)
        gl = g.long()
        dense[gl[:, 3], gl[:, 2], gl[:, 0], gl[:, 1]] = y.detach()
        wrng = np.random.default_rng(seed + 1000)
        gy = torch.from_numpy(wrng.integers(-4, 5, tuple(y.shape)).astype(np.float64))
        y.backward(gy)
        xg = xt.grad.numpy()
        # the reference's interval construction (bev_pool.py:48-54) as handed to its kernel
        mod.QuickCumsumTrainingCuda.apply(torch.from_numpy(x), torch.from_numpy(geom), torch.from_numpy(ranks), B, D, H, W)
        dense_grad = torch.zeros(B, D, H, W, C, dtype=torch.float64)
        dense_grad[gl[:, 3], gl[:, 2], gl[:, 0], gl[:, 1]] = gy
        out.update({f"{name}_cfg": np.array([seed, n, C, B, D, H, W, int(integer)], np.int64),
                    f"{name}_x_sha": sha(x), f"{name}_geom_sha": sha(geom), f"{name}_ranks_sha": sha(ranks),
                    f"{name}_rows": y.detach().numpy().astype(np.float64 if not integer else np.float32),
                    f"{name}_row_geom": g.numpy().astype(np.int32),
                    f"{name}_starts": rec["fwd_starts"].numpy(), f"{name}_lengths": rec["fwd_lengths"].numpy(),
                    f"{name}_dense_sha_f32": sha(dense.numpy().astype(np.float32)),
                    f"{name}_grad_rows": gy.numpy().astype(np.float32),
                    f"{name}_xgrad_sha_f32": sha(xg.astype(np.float32)),
                    f"{name}_xgrad_sample": xg.astype(np.float32)[::211].copy()})
    np.savez_compressed(os.path.join(HERE, "bev_pool_ref.npz"), **out)
    print("bev_pool_ref.npz:", {k: getattr(val, "shape", val) for k, val in out.items()})


# --------------------------------------------------------------------------------------- (E)
def make_real_sweep_goldens():
    """The one real LiDAR sweep the reference ships (demo/data/nuscenes/*LIDAR_TOP*.pcd.bin: raw float32 [N, 5], read with
    np.fromfile -- data, not a pickle) -> tests/golden/real_sweep.npz: the points themselves (693 760 B) and the outputs of
    the REFERENCE's compiled dynamic_voxelize (oracle/_ref) on them at the nuScenes grid.  The reference's hard_voxelize CPU
    path cannot run this grid (out-of-bounds table, voxelization_cpu.cpp:75,129-130), so the hard-voxelization quantities are
    DERIVED from the reference's per-point coordinates with the first-come rule of voxelization_cpu.cpp:70-98 in numpy
    (unique voxels in order of first occurrence, <= 10 points kept per voxel)."""
    import glob
    import voxel_layer_ref as v
    f = glob.glob(os.path.join(REF, "demo", "data", "nuscenes", "*LIDAR_TOP*.pcd.bin"))
    assert len(f) == 1, f
    pts = np.fromfile(f[0], dtype=np.float32).reshape(-1, 5)
    N = synthetic.NUSC
    dyn = ref_dynamic_voxelize(v, pts, N["voxel_size"], N["point_cloud_range"])
    inside = dyn[:, 0] >= 0
    lin = (dyn[:, 0].astype(np.int64) * 1440 + dyn[:, 1]) * 41 + dyn[:, 2]
    lin_in = lin[inside]
    uniq, first, counts = np.unique(lin_in, return_index=True, return_counts=True)
    order = np.argsort(first, kind="stable")            # voxel id = order of the first point (voxelization_cpu.cpp:78-90)
    hard_coors = dyn[inside][first[order]]
    hard_num = np.minimum(counts[order], N["max_num_points"]).astype(np.int32)
    out = {"points": pts, "dyn_coors_sha": sha(dyn), "dyn_coors_sample": dyn[::97].copy(), "n_inside": int(inside.sum()),
           "n_voxels": int(uniq.size), "hard_coors_sha": sha(hard_coors.astype(np.int32)), "hard_num_sha": sha(hard_num),
           "hard_coors_head": hard_coors[:64].astype(np.int32), "hard_num_head": hard_num[:64]}
    np.savez_compressed(os.path.join(HERE, "real_sweep.npz"), **out)
    print("real_sweep.npz:", {k: (val.shape if hasattr(val, "shape") and val.shape else val) for k, val in out.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "real_sweep":
        make_real_sweep_goldens()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "head":
        make_head_goldens()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bev_pool":
        make_bev_pool_goldens()
        sys.exit(0)
    make_voxel_goldens()
    make_lss_goldens()
    make_head_goldens()
    make_bev_pool_goldens()
    make_real_sweep_goldens()
