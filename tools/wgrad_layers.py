#!/usr/bin/env python3
"""Per-layer pair counts of the sparse encoder at batch 4 + the wgrad kernel's time per layer (HIP events)."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
import torch  # noqa: E402

wl = bench.LidarOnly(torch.device("cuda:0"), 4, 40000)
wl.collect_work()
from bevfusion_amd import spconv as sp  # noqa: E402
times = []
orig = sp._SparseConvFunction.backward


def spy(ctx, g):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = orig(ctx, g)
    b.record()
    times.append((a, b))
    return r


for _ in range(3):
    wl.step()
sp._SparseConvFunction.backward = staticmethod(spy)
wl.step()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in times][::-1]
out = []
for (P, ci, co, ni, no), t in zip(wl._layer_stats, ms):
    out.append(dict(pairs=P, cin=ci, cout=co, n_in=ni, n_out=no, bwd_ms=round(t, 3), gflop=round(2.0 * P * ci * co / 1e9, 2)))
print(json.dumps(out, indent=0))
