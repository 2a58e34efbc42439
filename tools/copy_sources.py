#!/usr/bin/env python3
"""Attribute the generic elementwise / copy kernels of one full training step to python source lines (torch profiler)."""
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
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    wl.step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue  # keep leaf aten ops only
    frames = [f for f in (ev.stack or []) if "/repo/" in f and "tools/" not in f]
    where = frames[0].split("/repo/")[-1] if frames else "(autograd backward / library)"
    shapes = str(ev.input_shapes)[:70] if ev.input_shapes else ""
    key = (ev.name, where[:90], shapes)
    agg[key][0] += ev.device_time_total
    agg[key][1] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in agg.values())
print("total device time of leaf aten ops: %.2f ms" % (tot / 1e3))
for (name, where, shapes), (us, n) in rows[:70]:
    print("%8.1f us  n=%3d  %-34s %-70s %s" % (us, n, name[:34], where, shapes))
