"""TEST / BASELINE INFRASTRUCTURE -- never imported by the product package.

CPU restatement of the reference's forward path for ONE batch of frames (BASELINE.json configs[0]: "single synthetic
nuScenes sample ... CPU-only forward through the projects/BEVFusion reference path"), assembled the way SURVEY 8(d)
asks for the CPU baseline:

  * hard voxelization + mean reduce        : the C oracle (single thread, like voxelization_cpu.cpp:46-144)
  * sparse encoder                         : rulebooks from the C oracle, arithmetic as gather -> torch.mm -> index_add_
                                             per kernel offset on all host cores (what spconv's CPU/native path does and
                                             what SC/sparse_functional.py:287-314 hands to ConvGemmOps), BN1d/ReLU = torch
  * sparse depth rasteriser / GT histogram : the C oracle (BF/depth_lss.py:372-449, 636-686)
  * frustum geometry, cells, ranks, sort   : the C oracle (BF/depth_lss.py:68-176)
  * bev_pool                               : `QuickCumsum`, the reference's only CPU-capable formulation
                                             (BF/ops/bev_pool/bev_pool.py:7-34: cumsum, keep the last row of each rank,
                                             difference) restated in torch with its autograd backward; fp32 as the
                                             reference would run it (timing) or fp64 (exact, parity tests)
  * every dense layer                      : torch.nn on the CPU (the product's module objects moved to the CPU run torch's
                                             CPU kernels: their HIP fast paths require CUDA tensors)

Only bench.py's `cpu_baseline` leg and tests/ use this file.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O


# ---------------------------------------------------------------------------------------------- bev_pool (QuickCumsum)
class QuickCumsum(torch.autograd.Function):
    """Restated BF/ops/bev_pool/bev_pool.py:7-34: rows sorted by rank; prefix-sum all rows, keep the last row of every
    rank, subtract the previous kept row.  Backward: every member row receives its rank's gradient."""

    @staticmethod
    def forward(ctx, x, geom, ranks):
        csum = torch.cumsum(x, 0)
        last = torch.ones(x.shape[0], dtype=torch.bool)
        last[:-1] = ranks[1:] != ranks[:-1]
        rows = csum[last]
        rows = torch.cat((rows[:1], rows[1:] - rows[:-1]))
        ctx.save_for_backward(last)
        ctx.mark_non_differentiable(geom)
        return rows, geom[last]

    @staticmethod
    def backward(ctx, g_rows, g_geom):
        (last,) = ctx.saved_tensors
        owner = torch.cumsum(last, 0)
        owner[last] -= 1
        return g_rows[owner], None, None


def bev_pool_quickcumsum(x_sorted, geom_sorted, ranks_sorted, B, D, H, W):
    """[n, C] rows sorted by rank -> dense [B, C, D, H, W] (the layout `bev_pool` returns, bev_pool.py:170)."""
    rows, g = QuickCumsum.apply(x_sorted, geom_sorted, ranks_sorted)
    g = g.long()
    out = torch.zeros((B, D, H, W, x_sorted.shape[1]), dtype=rows.dtype)
    out[g[:, 3], g[:, 2], g[:, 0], g[:, 1]] = rows
    return out.permute(0, 4, 1, 2, 3)


# ---------------------------------------------------------------------------------------------- sparse encoder
def sparse_conv_mm(feats, weight, pair_fwd, n_out):
    """out[N_out, Cout] = sum_k  W[:, k, :] . in[pair_fwd[k, :]]  as gather -> mm -> index_add_ per kernel offset."""
    cout, cin = weight.shape[0], weight.shape[-1]
    w = weight.reshape(cout, -1, cin)
    out = feats.new_zeros((n_out, cout))
    pf = torch.from_numpy(np.ascontiguousarray(pair_fwd)).long()
    for k in range(pf.shape[0]):
        rows = torch.nonzero(pf[k] >= 0).squeeze(1)
        if rows.numel() == 0:
            continue
        out.index_add_(0, rows, feats.index_select(0, pf[k].index_select(0, rows)) @ w[:, k, :].t())
    return out


def _bn1d(bn, x, residual=None, relu=False):
    y = F.batch_norm(x, None, None, bn.weight, bn.bias, training=True, eps=bn.eps) if bn.training else \
        F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, training=False, eps=bn.eps)
    if residual is not None:
        y = y + residual
    return torch.relu(y) if relu else y


class _Sp:
    def __init__(self, feats, idx, shape):
        self.feats, self.idx, self.shape = feats, idx, list(shape)
        self.subm = None  # SubM rulebook of this index set (the reference rebuilds it per conv: indice_key=None)


def _conv(conv, x, B):
    if conv.subm:
        if x.subm is None:
            x.subm = O.rulebook_subm(x.idx, x.shape, conv.kernel_size, conv.dilation[0])
        y = _Sp(sparse_conv_mm(x.feats, conv.weight, x.subm, x.idx.shape[0]), x.idx, x.shape)
        y.subm = x.subm
    else:
        oi, pf, pb, osz = O.rulebook_sparse(x.idx, x.shape, conv.kernel_size, conv.stride, conv.padding)
        y = _Sp(sparse_conv_mm(x.feats, conv.weight, pf, oi.shape[0]), oi, osz)
    if conv.bias is not None:
        y.feats = y.feats + conv.bias
    return y


def _conv_module(seq, x, B):
    """SparseSequential(conv, BN1d, ReLU) of make_sparse_convmodule."""
    mods = list(seq._modules.values())
    y = _conv(mods[0], x, B)
    bn = mods[1] if len(mods) > 1 else None
    if bn is not None:
        y.feats = _bn1d(bn, y.feats, relu=len(mods) > 2)
    return y


def sparse_encoder_forward(enc, feats, coords, B):
    """BEVFusionSparseEncoder.forward (BF/sparse_encoder.py:112-156) on the CPU; feats f32[N, 5] torch, coords i32[N, 4] numpy."""
    from bevfusion_amd.sparse_encoder import SparseBasicBlock
    x = _Sp(feats, np.ascontiguousarray(coords, dtype=np.int32), enc.sparse_shape)
    x = _conv_module(enc.conv_input, x, B)
    for stage in enc.encoder_layers:
        for blk in stage:
            if isinstance(blk, SparseBasicBlock):
                identity = x.feats
                y = _conv(blk.conv1, x, B)
                y.feats = _bn1d(blk.norm1, y.feats, relu=True)
                y = _conv(blk.conv2, y, B)
                y.feats = _bn1d(blk.norm2, y.feats, residual=identity, relu=True)
                x = y
            else:
                x = _conv_module(blk, x, B)
    x = _conv_module(enc.conv_out, x, B)
    X, Y, Z = x.shape
    n, c = x.feats.shape
    # .dense() -> permute(0,1,4,2,3) -> view(B, C*Z, X, Y)   (BF/sparse_encoder.py:147-151), differentiable scatter
    idx = torch.from_numpy(x.idx.astype(np.int64))
    dense = x.feats.new_zeros((B, c, Z, X, Y))
    dense[idx[:, 0], :, idx[:, 3], idx[:, 1], idx[:, 2]] = x.feats
    return dense.view(B, c * Z, X, Y)


def voxelize_mean(points_np, nusc):
    """BEVFusion.voxelize (BF/bevfusion.py:227-255): per-sample hard voxelization, batch id, mean reduce."""
    feats, coords = [], []
    for b, pts in enumerate(points_np):
        vox, coors, num = O.hard_voxelize(pts, nusc["voxel_size"], nusc["point_cloud_range"], nusc["max_num_points"],
                                          nusc["max_voxels"][0])
        feats.append(O.voxel_mean(vox, num))
        coords.append(np.concatenate([np.full((len(coors), 1), b, np.int32), coors], 1))
    return torch.from_numpy(np.concatenate(feats)), np.concatenate(coords)


# ---------------------------------------------------------------------------------------------- camera branch
def view_transform_forward(vt, img_feats, points_np, mats, exact=False):
    """DepthLSSTransform.forward (BF/depth_lss.py:344-551, 624-733) on the CPU.  img_feats [B, N, C, fH, fW];
    mats: dict of the five 4x4 matrix batches (numpy).  exact=True evaluates QuickCumsum in fp64."""
    B, N = img_feats.shape[:2]
    iH, iW = vt.image_size
    fH, fW = vt.feature_size
    D, C = vt.D, vt.C
    # sparse depth images (:372-449) and GT depth distribution (:636-686)
    inv_aug = np.linalg.inv(mats["lidar_aug_matrix"].astype(np.float64)).astype(np.float32)
    depth_img = np.stack([O.rasterise_depth(points_np[b], inv_aug[b, :3, :3], mats["lidar_aug_matrix"][b, :3, 3],
                                            mats["lidar2img"][b], mats["img_aug_matrix"][b], iH, iW) for b in range(B)])
    counts, gt_distr = O.depth_histogram(depth_img.reshape(B * N, iH, iW), fH, fW, D, vt.dbound)
    d = torch.from_numpy(depth_img.reshape(B * N, 1, iH, iW))
    x = img_feats.reshape(B * N, -1, fH, fW)
    x = vt.depthnet(torch.cat([vt.dtransform(d), x], dim=1))
    depth = x[:, :D].softmax(dim=1)
    if vt.training:  # straight-through max(gt, pred) (:702-706)
        aux = torch.from_numpy(gt_distr).view(B * N, fH, fW, D).permute(0, 3, 1, 2)
        depth = depth + (torch.maximum(aux, depth) - depth).detach()
    feat = x[:, D:D + C]
    outer = depth.unsqueeze(1) * feat.unsqueeze(2)                                   # [BN, C, D, fH, fW] (:723)
    outer = outer.view(B, N, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2)                # (:725)
    gf, kept, ranks, order = bev_geometry(vt, mats, B)
    rows = outer.reshape(B * N * D * fH * fW, C)[torch.from_numpy(kept)][torch.from_numpy(order)]  # (:190-195) two gathers
    nx = vt._nx_host
    pooled = bev_pool_quickcumsum(rows.double() if exact else rows, torch.from_numpy(gf), torch.from_numpy(ranks), B,
                                  nx[2], nx[0], nx[1])
    bev = torch.cat(pooled.float().unbind(dim=2), 1)                                 # (:203) collapse Z
    return vt.downsample(bev)


def bev_geometry(vt, mats, B):
    """geometry -> cells -> kept / ranks / argsort (BF/depth_lss.py:68-176) through the C oracle."""
    t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in mats.items()}
    cal = vt._calibration(t["cam2img"], t["cam2lidar"], t["img_aug_matrix"], t["lidar_aug_matrix"])
    c = {k: v.numpy() for k, v in cal.items()}
    combine = cal["camera2lidar_rots"].matmul(cal["intrins_inverse"]).numpy()
    geom = O.frustum_geometry(vt.frustum.detach().numpy(), c["post_trans"], c["post_rots_inverse"], combine,
                              c["camera2lidar_trans"], c["extra_rots"], c["extra_trans"])
    return O.bev_pool_aux(geom, B, np.array(vt._origin_host, np.float32), np.array(vt._dx_host, np.float32),
                          np.array(vt._nx_host, np.int32))


def model_forward(model, points_np, imgs, mats, nusc, exact=False, timings=None):
    """BEVFusion.extract_feat + bbox_head forward (BF/bevfusion.py:305-381, BF/bevfusion_head.py:198-299) on the CPU.
    `model`: the product model object moved to the CPU in fp32 (weights shared with the GPU run in parity tests)."""
    tm = timings if timings is not None else {}

    def clock(name, t0):
        tm[name] = tm.get(name, 0.0) + time.perf_counter() - t0

    B = len(points_np)
    feats = []
    if imgs is not None and model.view_transform is not None:
        t0 = time.perf_counter()
        x = model.img_backbone(imgs.reshape(B * imgs.shape[1], *imgs.shape[2:]))
        x = model.img_neck(x)
        x = x[0] if not torch.is_tensor(x) else x
        clock("img_backbone_neck", t0)
        t0 = time.perf_counter()
        feats.append(view_transform_forward(model.view_transform, x.reshape(B, -1, *x.shape[1:]), points_np, mats, exact))
        clock("view_transform", t0)
    if model.pts_middle_encoder is not None:
        t0 = time.perf_counter()
        vf, coords = voxelize_mean(points_np, nusc)
        clock("voxelize", t0)
        t0 = time.perf_counter()
        feats.append(sparse_encoder_forward(model.pts_middle_encoder, vf, coords, B))
        clock("sparse_encoder", t0)
    t0 = time.perf_counter()
    x = model.fusion_layer(feats) if model.fusion_layer is not None else feats[0]
    x = model.pts_neck(model.pts_backbone(x))
    clock("fuser_backbone_neck", t0)
    t0 = time.perf_counter()
    outs = model.bbox_head(x) if model.bbox_head is not None else x
    clock("head", t0)
    return outs, x, feats
