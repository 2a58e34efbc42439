"""Cross attention of a few queries over very many keys (csrc/attn.hip): the TransFusion decoder layer's 200 object
queries against all 32 400 BEV cells (BF/transformer.py:60-105).  Same maths as torch's scaled_dot_product_attention
(softmax(Q K^T / sqrt(d)) with dropout on the weights, times V) in bf16 with fp32 softmax statistics; the key axis is split
over the chip instead of the (tiny) query axis.  Inputs stay in the [B, L, H*16] layout the projections produce."""
import itertools
import math
import os

import torch

from . import _lib

ENABLED = os.environ.get("BFHIP_SPLITK_ATTN", "1") == "1"
MIN_KEYS = 2048
_WS = {}
_calls = itertools.count(1)


def supported(q, k, v, num_heads):
    return (ENABLED and q.is_cuda and q.dtype == k.dtype == v.dtype == torch.bfloat16 and q.dim() == 3
            and q.shape[2] == num_heads * 16 and q.shape[1] <= 256 and k.shape[1] >= MIN_KEYS and k.shape == v.shape
            and q.shape[0] == k.shape[0])


def _workspace(t, nbytes, stream):
    key = (t.device, stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=t.device)
    return ws


def next_seed():
    """A fresh 64-bit dropout seed per call, reproducible under torch.manual_seed (no device RNG state is consumed)."""
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + next(_calls) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


_step_counter = {}


def step_counter(device):
    """Device-side u64 call counter mixed into every dropout seed (csrc/attn.hip eff_seed).  Advance it INSIDE a captured
    training step (`step_counter(dev).add_(1)`): a replayed hipGraph repeats its host-side seed argument, the counter still
    gives every replay its own mask.  Zero (the default) leaves the host seed as it is."""
    key = torch.device(device)
    t = _step_counter.get(key)
    if t is None:
        t = _step_counter[key] = torch.zeros(1, dtype=torch.int64, device=key)
    return t


def dropout_mask(B, H, Lq, Lk, p, seed, device):
    """The keep mask the kernels use (bool [B, H, Lq, Lk]); for tests."""
    m = torch.empty(B * H, Lq, Lk, dtype=torch.uint8, device=device)
    _lib.call("bfhip_attn_dropout_mask", B, H, Lq, Lk, float(p), seed, step_counter(device).data_ptr(), m.data_ptr(),
              _lib.stream_of(m))
    return m.view(B, H, Lq, Lk).bool()


class _CrossAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, H, dropout_p, seed):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, Lq, E = q.shape
        Lk = k.shape[1]
        o = torch.empty_like(q)
        lse = torch.empty(B * H, Lq, dtype=torch.float32, device=q.device)
        stream = _lib.stream_of(q)
        nbytes = _lib.call_size("bfhip_attn_workspace_bytes", B, H, Lq, Lk)
        ws = _workspace(q, nbytes, stream)
        scale = 1.0 / math.sqrt(E // H)
        _lib.call("bfhip_attn_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), B, H, Lq, Lk, scale, dropout_p, seed,
                  step_counter(q.device).data_ptr(), o.data_ptr(), lse.data_ptr(), ws.data_ptr(), ws.numel(), stream)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.cfg = (H, dropout_p, seed, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        H, dropout_p, seed, scale = ctx.cfg
        B, Lq, E = q.shape
        Lk = k.shape[1]
        do = do.contiguous()
        if do.dtype != torch.bfloat16:
            do = do.to(torch.bfloat16)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        stream = _lib.stream_of(q)
        ws = _workspace(q, _lib.call_size("bfhip_attn_workspace_bytes", B, H, Lq, Lk), stream)
        _lib.call("bfhip_attn_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), B, H,
                  Lq, Lk, scale, dropout_p, seed, step_counter(q.device).data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                  ws.data_ptr(), ws.numel(), stream)
        return dq, dk, dv, None, None, None


def cross_attention(q, k, v, num_heads, dropout_p=0.0, seed=None):
    """q [B, Lq, H*16], k / v [B, Lk, H*16] (bf16) -> [B, Lq, H*16]; heads are channel groups of 16."""
    if dropout_p > 0.0 and seed is None:
        seed = next_seed()
    return _CrossAttention.apply(q, k, v, num_heads, float(dropout_p), int(seed or 0))
