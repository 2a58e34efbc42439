#!/usr/bin/env python3
"""wgrad kernel time with fp32 vs bf16 feature storage on a synthetic SubM rulebook."""
import sys, os
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bevfusion_amd  # noqa
from bevfusion_amd import _lib, spconv as sp

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
shape = (360, 360, 11)
n = 60000
# clustered voxels (street-like): random walk blobs
base = rng.integers(0, [shape[0], shape[1], shape[2]], size=(n // 20, 3))
pts = (base[:, None, :] + rng.integers(-3, 4, size=(n // 20, 20, 3))).reshape(-1, 3)
pts = np.clip(pts, 0, np.array(shape) - 1)
pts = np.unique(pts, axis=0)
idx = np.concatenate([np.zeros((len(pts), 1), np.int64), pts], 1).astype(np.int32)
N = len(idx)
ind = torch.from_numpy(idx).to(dev)
data = sp.build_subm_rulebook(ind, 1, list(shape), [3, 3, 3], [1, 1, 1])
print("rows", N, "pairs", int(data.n_pairs.sum()))
lib = _lib.load()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for C in (16, 32, 64, 128):
    f32 = torch.randn(N, C, device=dev)
    g32 = torch.randn(N, C, device=dev)
    f16, g16 = f32.to(torch.bfloat16), g32.to(torch.bfloat16)
    dw = torch.empty(C, 3, 3, 3, C, device=dev)
    ws = torch.empty(lib.bfhip_spconv_wgrad_workspace_bytes(27, C, C, N), dtype=torch.uint8, device=dev)

    def run(a, b, io):
        rc = lib.bfhip_spconv_wgrad(a.data_ptr(), b.data_ptr(), data.pair_fwd.data_ptr(), N, 27, N, C, C, None, dw.data_ptr(), io,
                                    ws.data_ptr(), ws.numel(), _lib.stream_of(a))
        assert rc == 0
    t32 = timed(lambda: run(f32, g32, 0))
    ref = dw.clone()
    t16 = timed(lambda: run(f16, g16, 1))
    err = float((dw - ref).abs().max() / ref.abs().max())
    print("C=%3d  fp32 %.1f us   bf16 %.1f us   rel diff %.3g" % (C, t32, t16, err))
