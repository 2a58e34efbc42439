#!/usr/bin/env python3
"""Is the full step host-bound?  Times (a) python issue time of a step with no sync, (b) wall per step with a final sync."""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
import torch  # noqa: E402

wl = bench.FullModel(torch.device("cuda:0"), 4, 40000)
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
issue = []
for _ in range(K):
    a = time.perf_counter()
    wl.step()
    issue.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("issue ms/step (python returns): %.2f   wall ms/step: %.2f   drain after last issue: %.2f ms" %
      (sum(issue) / K * 1e3, (t2 - t0) / K * 1e3, (t2 - t1) * 1e3))
print("per-step issue ms:", " ".join("%.1f" % (x * 1e3) for x in issue))
