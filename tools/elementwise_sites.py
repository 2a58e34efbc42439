#!/usr/bin/env python3
"""Which call sites launch the torch copy / add / cat / fill / reduce kernels of one full training step?
torch.profiler with shapes and python stacks; device time summed per (op, shapes, innermost frame inside this repo)."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    wl.step()
    torch.cuda.synchronize()
WANT = ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::fill_", "aten::zero_", "aten::sum", "aten::mul", "aten::_to_copy",
        "aten::contiguous", "aten::clone", "aten::max_pool2d_with_indices_backward", "aten::_softmax", "aten::index", "aten::where")
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.name not in WANT:
        continue
    dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
    self_dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if not self_dt:
        continue
    site = "?"
    for fr in (ev.stack or []):
        if "bevfusion" in fr or "bench.py" in fr:
            site = fr.split("/")[-1][:70]
            break
    key = (ev.name, str(ev.input_shapes)[:70], site)
    agg[key][0] += self_dt
    agg[key][1] += 1
tot = 0.0
for (name, shapes, site), (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print("%8.1f us n=%3d %-22s %-70s %s" % (us, n, name, shapes, site))
    tot += us
print("listed total %.2f ms" % (tot / 1e3))
