#!/usr/bin/env python3
"""cProfile of the host side of the full training step (where does the python / launch time go)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
import torch  # noqa: E402

wl = bench.FullModel(torch.device("cuda:0"), 4, 40000)
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    wl.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(60)
st.sort_stats("tottime").print_stats(35)
