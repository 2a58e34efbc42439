"""bf16 parameters with fp32 master weights for the dense (MFMA) layers.

Under plain autocast every conv / linear weight is re-cast fp32 -> bf16 in each forward and its bf16 gradient is cast
back to fp32 in each backward: ~320 tiny launches per step for this model.  Here the weights of those layers ARE bf16
(what the kernels consume), the optimizer owns fp32 master copies, and the two conversions are one multi-tensor copy
each per step.  The arithmetic is unchanged: the forward uses the same bf16 values autocast would produce and the weight
gradient the same bf16 tensor the bf16 kernels emit.  With DDP the gradient all-reduce of these layers moves half the
bytes (bf16 buckets).  Layers that the model runs in fp32 islands, BatchNorm / LayerNorm and the sparse encoder keep
fp32 parameters."""
import os

import torch
from torch import nn

_LOW_TYPES = (nn.Conv1d, nn.Conv2d, nn.ConvTranspose2d, nn.Linear, nn.MultiheadAttention)


def low_precision_parameters(model, exclude=("pts_middle_encoder", "heatmap_head")):
    """Parameters of the conv / linear / attention-projection layers that run under bf16 autocast."""
    out, seen = [], set()
    for name, mod in model.named_modules():
        if any(part in name.split(".") for part in exclude) or not isinstance(mod, _LOW_TYPES):
            continue
        for p in mod.parameters(recurse=False):
            if p.requires_grad and p.dtype == torch.float32 and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


def skip_nonfinite_step(opt, total_norm):
    """A step whose gradient norm is NaN / inf leaves parameters, moments and step counters untouched -- decided on the
    device (the fused AdamW kernels' `found_inf` input, what GradScaler uses), no host read.  This is how a poisoned loss
    (an invalid Hungarian cost matrix; a frame that overflowed a static row capacity, BEVFusion.loss) costs one skipped
    step instead of NaN weights.  The reference's OptimWrapper has no such guard: with it a NaN loss ends the run."""
    opt.found_inf = (~torch.isfinite(total_norm)).to(torch.float32).reshape(())
    opt.grad_scale = None


class MasterWeightAdamW:
    """AdamW (fused) over fp32 master copies of `low` (converted to bf16 in place) plus the remaining fp32 parameters."""

    def __init__(self, model, lr, weight_decay, max_grad_norm=None, exclude=("pts_middle_encoder", "heatmap_head"),
                 capturable=False):
        self.low = low_precision_parameters(model, exclude)
        low_ids = {id(p) for p in self.low}
        self.other = [p for p in model.parameters() if p.requires_grad and id(p) not in low_ids]
        self.master = [p.detach().clone().float() for p in self.low]
        for p in self.low:
            p.data = p.data.to(torch.bfloat16)
        for m in self.master:
            m.grad = torch.zeros_like(m)
        self.max_grad_norm = max_grad_norm
        # capturable: step counters live on the device, so step() can sit inside a captured hipGraph
        self.opt = torch.optim.AdamW(self.master + self.other, lr=lr, weight_decay=weight_decay, fused=True,
                                     capturable=capturable)
        # Direct path (default): the same fused multi-tensor kernels torch.optim.AdamW(fused=True) and clip_grad_norm_ launch,
        # called on lists prepared ONCE -- the optimizer object re-derives its per-parameter lists, state dicts and device
        # groups in Python on every step (2.8 ms of host time per step for the ~450 tensors of this model, on a step whose host
        # and GPU sides are balanced).  Same arithmetic, same found_inf skip; BFHIP_DIRECT_ADAMW=0 goes through the object.
        # Flat path (default, csrc/optim.hip): clipping + AdamW + the bf16 refresh of ALL tensors in three launches from a
        # device-resident table; per step only the gradient pointers are uploaded.  BFHIP_FLAT_ADAMW=0 selects the paths below.
        self.flat = (os.environ.get("BFHIP_FLAT_ADAMW", "1") == "1" and not capturable and max_grad_norm is not None
                     and all(p.is_cuda for p in self.master + self.other))
        if self.flat:
            self._build_flat(lr, weight_decay)
        self.direct = os.environ.get("BFHIP_DIRECT_ADAMW", "1") == "1" and not capturable and not self.flat
        if self.direct:
            self._params = self.master + self.other
            dev = self._params[0].device
            self._exp_avg = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self._params]
            self._exp_avg_sq = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self._params]
            self._steps_flat = torch.zeros(len(self._params), dtype=torch.float32, device=dev)
            self._steps = list(self._steps_flat.unbind(0))  # 0-d views: one add_ on the flat tensor advances them all
            g = self.opt.param_groups[0]
            self._hyper = dict(lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], weight_decay=g["weight_decay"], eps=g["eps"])
        # transposed bf16 copies of the conv weights for the HIP data gradients, all layers in one launch after every update
        # (conv2d.TransposedWeights); built last: the parameters have their final storage now.  BFHIP_WT_CACHE=0: per-call transposes
        self.transposed = None
        if os.environ.get("BFHIP_WT_CACHE", "1") == "1" and all(p.is_cuda for p in self.master + self.other):
            from .conv2d import TransposedWeights
            self.transposed = TransposedWeights(model.modules())

    # ------------------------------------------------------------------ flat path
    def _build_flat(self, lr, weight_decay):
        import numpy as np
        from . import _lib
        lib = _lib.load()
        g = self.opt.param_groups[0]
        self._hyper_flat = (float(lr), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(weight_decay))
        dev = self.master[0].device if self.master else self.other[0].device
        # (parameter whose .grad is read, fp32 master, bf16 copy or None)
        self._flat_items = [(p, m, p) for p, m in zip(self.low, self.master)] + [(p, p.data, None) for p in self.other]
        for _, m, _ in self._flat_items:
            assert m.dtype == torch.float32
        self._flat_m = [torch.zeros_like(m, memory_format=torch.preserve_format) for _, m, _ in self._flat_items]
        self._flat_v = [torch.zeros_like(m, memory_format=torch.preserve_format) for _, m, _ in self._flat_items]
        seg_dt = np.dtype([("master", "<u8"), ("m", "<u8"), ("v", "<u8"), ("lowp", "<u8"), ("n", "<i8"), ("grad_bf16", "<i4"),
                           ("pad", "<i4")])
        assert seg_dt.itemsize == lib.bfhip_adamw_segment_bytes()
        chunk = lib.bfhip_adamw_chunk_elems()
        segs = np.zeros(len(self._flat_items), seg_dt)
        chunks = []
        for i, ((p, m, low), em, ev) in enumerate(zip(self._flat_items, self._flat_m, self._flat_v)):
            # element i of every array of a record must be the same logical element: all share the parameter's dense layout
            assert em.stride() == m.stride() and (low is None or low.stride() == m.stride()), "layout mismatch"
            segs[i] = (m.data_ptr(), em.data_ptr(), ev.data_ptr(), low.data_ptr() if low is not None else 0, m.numel(),
                       1 if low is not None else 0, 0)
            chunks += [(i, c) for c in range(-(-m.numel() // chunk))]
        self._n_chunks = len(chunks)
        self._segs_dev = torch.from_numpy(segs.view(np.uint8).copy()).to(dev)
        self._chunks_dev = torch.tensor(chunks, dtype=torch.int32, device=dev)
        self._partial_dev = torch.empty(self._n_chunks, dtype=torch.float32, device=dev)
        self.scalars = torch.zeros(8, dtype=torch.float32, device=dev)  # [0] clip, [1] found_inf, [2] step, [5] gradient norm
        n = len(self._flat_items)
        self._gptr_host = [torch.zeros(n, dtype=torch.int64).pin_memory() for _ in range(2)]  # alternating: a copy may be in flight
        self._gptr_np = [t.numpy() for t in self._gptr_host]
        self._gptr_dev = torch.zeros(n, dtype=torch.int64, device=dev)
        self._flip = 0
        self._grad_dtype = [torch.bfloat16 if low is not None else torch.float32 for _, _, low in self._flat_items]

    def _step_flat(self):
        from . import _lib
        ptrs, keep = [], []
        for (p, m, _), dt in zip(self._flat_items, self._grad_dtype):
            g = p.grad
            if g is None:
                ptrs.append(0)
                continue
            if g.dtype != dt or g.stride() != m.stride():
                # a gradient that is not laid out like its parameter (or not in its dtype): one conforming copy
                c = torch.empty_strided(m.size(), m.stride(), dtype=dt, device=m.device)
                c.copy_(g)
                keep.append(c)
                g = c
            ptrs.append(g.data_ptr())
        k = self._flip
        self._flip ^= 1
        self._gptr_np[k][:] = ptrs
        self._gptr_dev.copy_(self._gptr_host[k], non_blocking=True)
        lr, b1, b2, eps, wd = self._hyper_flat
        _lib.call("bfhip_adamw_step", self._segs_dev.data_ptr(), self._gptr_dev.data_ptr(), self._chunks_dev.data_ptr(), self._n_chunks,
                  self._partial_dev.data_ptr(), self.scalars.data_ptr(), lr, b1, b2, eps, wd, float(self.max_grad_norm),
                  _lib.stream_of(self._gptr_dev))
        self._keep = keep  # conforming copies stay alive until the next step (the kernels read them asynchronously)

    def zero_grad(self):
        for p in self.low:
            p.grad = None
        for p in self.other:
            p.grad = None

    @torch.no_grad()
    def step(self):
        self._update()
        if self.transposed is not None:
            self.transposed.refresh()

    def _update(self):
        if self.flat:
            return self._step_flat()
        have = [(m.grad, p.grad) for m, p in zip(self.master, self.low) if p.grad is not None]
        if have:
            torch._foreach_copy_([a for a, _ in have], [b for _, b in have])  # bf16 -> fp32, one multi-tensor kernel
        if len(have) != len(self.low):
            # a parameter unused in this step: zero gradient, as the flat all-reduce path produces at world size > 1
            # (grad_sync.FlatGradAllReduce zero-fills), so the update does not depend on the world size.  Deviation from
            # torch.optim (and the reference's optimizer), which SKIP a parameter without a gradient: here it still gets its
            # weight decay and moment decay -- for every parameter kind alike (bf16-with-master and fp32)
            torch._foreach_zero_([m.grad for m, p in zip(self.master, self.low) if p.grad is None])
        for p in self.other:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        if self.direct:
            grads = [p.grad for p in self._params]
            found_inf = None
            if self.max_grad_norm is not None:
                # clip_grad_norm_(foreach=True): per-tensor 2-norms in one multi-tensor launch, the norm of the norms, one scale
                total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads, 2.0)), 2.0)
                torch._foreach_mul_(grads, torch.clamp(self.max_grad_norm / (total + 1e-6), max=1.0))
                found_inf = (~torch.isfinite(total)).to(torch.float32)
            self._steps_flat.add_(1)
            torch._fused_adamw_(self._params, grads, self._exp_avg, self._exp_avg_sq, [], self._steps, amsgrad=False,
                                maximize=False, grad_scale=None, found_inf=found_inf, **self._hyper)
            if found_inf is not None:
                self._steps_flat.sub_(found_inf)   # a skipped step does not advance the bias correction
        else:
            if self.max_grad_norm is not None:
                norm = torch.nn.utils.clip_grad_norm_(self.master + self.other, self.max_grad_norm, foreach=True)
                skip_nonfinite_step(self.opt, norm)
            self.opt.step()
        torch._foreach_copy_(self.low, self.master)                  # fp32 -> bf16
