#!/usr/bin/env python3
"""aten ops of one full training step by (name, input shapes): count and device time (torch.profiler) -- to spot small ops that
come in dozens (slice backward zero-fills, dtype round trips, scalar adds)."""
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
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    dt = getattr(ev, "self_device_time_total", 0) or 0
    if not ev.name.startswith("aten::") or not dt:
        continue
    k = (ev.name, str(ev.input_shapes)[:90])
    agg[k][0] += 1
    agg[k][1] += dt
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
print("ops with device time: %d launches-ish, %.2f ms" % (sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values()) / 1e3))
for (n, sh), (c, t) in rows[:70]:
    print("%4d %8.1f us  %-34s %s" % (c, t, n, sh))
