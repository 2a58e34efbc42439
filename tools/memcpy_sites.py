#!/usr/bin/env python3
"""Which host-side calls of one full training step end in a memcpy (HtoD / DtoD / DtoH)?  torch.profiler, one step: memcpy
events by kind and size, and the aten ops (with input shapes) that launched them."""
import collections
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

wl = bench.FullModel(torch.device("cuda:0"), 4, 40000)
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    wl.step()
    torch.cuda.synchronize()
kinds = collections.Counter()
for ev in prof.events():
    n = ev.name
    if "Memcpy" in n or "memcpy" in n or "Memset" in n:
        kinds[n] += 1
print(dict(kinds))
ops = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::_to_copy", "aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::tensor", "aten::lift_fresh", "aten::_local_scalar_dense", "aten::item"):
        shapes = str(ev.input_shapes)[:60]
        ops[(ev.name, shapes)] += 1
for (n, sh), c in ops.most_common(60):
    print("%4d  %-26s %s" % (c, n, sh))
