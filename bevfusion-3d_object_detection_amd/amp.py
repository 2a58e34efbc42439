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
        self.direct = os.environ.get("BFHIP_DIRECT_ADAMW", "1") == "1" and not capturable
        if self.direct:
            self._params = self.master + self.other
            dev = self._params[0].device
            self._exp_avg = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self._params]
            self._exp_avg_sq = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self._params]
            self._steps_flat = torch.zeros(len(self._params), dtype=torch.float32, device=dev)
            self._steps = list(self._steps_flat.unbind(0))  # 0-d views: one add_ on the flat tensor advances them all
            g = self.opt.param_groups[0]
            self._hyper = dict(lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], weight_decay=g["weight_decay"], eps=g["eps"])

    def zero_grad(self):
        for p in self.low:
            p.grad = None
        for p in self.other:
            p.grad = None

    @torch.no_grad()
    def step(self):
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
