#!/usr/bin/env python3
"""Wall time of the first training steps of the full workload (how many warm-up steps does steady state need?)."""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
import torch  # noqa: E402

t0 = time.perf_counter()
wl = bench.FullModel(torch.device("cuda:0"), 4, 40000)
torch.cuda.synchronize()
print("construct %.1f s" % (time.perf_counter() - t0))
for i in range(12):
    t = time.perf_counter()
    wl.step()
    torch.cuda.synchronize()
    print("step %2d: %.1f ms" % (i, (time.perf_counter() - t) * 1e3), flush=True)
