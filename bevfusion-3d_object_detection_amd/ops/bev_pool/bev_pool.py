"""Host mirror of BF/ops/bev_pool/bev_pool.py: `bev_pool(feats, coords, ranks, B, D, H, W, is_training)`
plus the autograd Functions the reference exposes (QuickCumsumTrainingCuda :43-90,
QuickCumsumCuda :93-143).  The arithmetic runs in csrc/bev_pool.hip.
"""
import torch

from . import bev_pool_ext


def intervals_from_ranks(ranks):
    """Interval starts/lengths of equal-rank runs (BF/ops/bev_pool/bev_pool.py:48-54).

    ranks: sorted int tensor [n].  Returns (interval_starts i32[m], interval_lengths i32[m]).
    """
    n = ranks.shape[0]
    if n == 0:
        empty = torch.zeros(0, dtype=torch.int32, device=ranks.device)
        return empty, empty.clone()
    is_start = torch.ones(n, device=ranks.device, dtype=torch.bool)
    is_start[1:] = ranks[1:] != ranks[:-1]
    starts = torch.nonzero(is_start, as_tuple=False).flatten().to(torch.int32)
    ends = torch.cat((starts[1:], starts.new_tensor([n])))
    return starts, (ends - starts).to(torch.int32)


class QuickCumsumTrainingCuda(torch.autograd.Function):
    """Training path: builds the intervals on device and keeps them for backward
    (reference: BF/ops/bev_pool/bev_pool.py:43-90)."""

    @staticmethod
    def forward(ctx, x, geom_feats, ranks, B, D, H, W):
        starts, lengths = intervals_from_ranks(ranks)
        geom_feats = geom_feats.int().contiguous()
        x = x.contiguous()
        out = bev_pool_ext.bev_pool_forward(x, geom_feats, lengths, starts, B, D, H, W)
        ctx.save_for_backward(starts, lengths, geom_feats)
        ctx.saved_shapes = (int(B), int(D), int(H), int(W))
        return out

    @staticmethod
    def backward(ctx, out_grad):
        starts, lengths, geom_feats = ctx.saved_tensors
        B, D, H, W = ctx.saved_shapes
        # intervals built from ranks partition [0, n): the zero-fill of x_grad can be skipped
        x_grad = bev_pool_ext.bev_pool_backward(out_grad.contiguous(), geom_feats, lengths, starts,
                                                B, D, H, W, _cover_all=True)
        return x_grad, None, None, None, None, None, None


class QuickCumsumCuda(torch.autograd.Function):
    """Inference path with caller-provided intervals (reference: bev_pool.py:93-143; its backward
    raises NotImplementedError, and so does this one)."""

    @staticmethod
    def forward(ctx, x, geom_feats, interval_lengths, interval_starts, B, D, H, W):
        return bev_pool_ext.bev_pool_forward(x.contiguous(), geom_feats, interval_lengths, interval_starts,
                                             B, D, H, W)

    @staticmethod
    def backward(ctx, out_grad):
        raise NotImplementedError


def _as_int(v):
    return int(v.item()) if torch.is_tensor(v) else int(v)


def bev_pool(feats, coords, ranks, B, D, H, W, is_training):
    """feats f32[n,c] sorted by rank, coords int[n,4]=(x,y,z,b), ranks int[n] -> f32[B,c,D,H,W]
    (reference: BF/ops/bev_pool/bev_pool.py:146-172)."""
    assert feats.shape[0] == coords.shape[0]
    B, D, H, W = _as_int(B), _as_int(D), _as_int(H), _as_int(W)
    if is_training:
        x = QuickCumsumTrainingCuda.apply(feats, coords, ranks, B, D, H, W)
    else:
        starts, lengths = intervals_from_ranks(ranks)
        if coords.dtype != torch.int32:
            coords = coords.int()
        x = QuickCumsumCuda.apply(feats, coords.contiguous(), lengths, starts, B, D, H, W)
    # [B, D, H, W, C] -> [B, C, D, H, W]
    return x.permute(0, 4, 1, 2, 3).contiguous()
