#!/usr/bin/env python3
"""wgrad main-kernel time (library profiler scope) and whole-call time (HIP events) on synthetic SubM rulebooks sized like
the encoder's stages.  BFHIP_WGRAD_V selects the 64x64 kernel variant while two are kept for A/B."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bevfusion_amd  # noqa
from bevfusion_amd import _lib, spconv as sp

dev = torch.device("cuda:0")
lib = _lib.load()


def rulebook(n_target, shape, seed):
    rng = np.random.default_rng(seed)
    m = n_target // 12
    base = rng.integers(0, [shape[0], shape[1], shape[2]], size=(m, 3))
    pts = (base[:, None, :] + rng.integers(-2, 3, size=(m, 20, 3))).reshape(-1, 3)
    pts = np.unique(np.clip(pts, 0, np.array(shape) - 1), axis=0)
    idx = np.concatenate([np.zeros((len(pts), 1), np.int64), pts], 1).astype(np.int32)
    ind = torch.from_numpy(idx).to(dev)
    return len(idx), sp.build_subm_rulebook(ind, 1, list(shape), [3, 3, 3], [1, 1, 1])


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    _lib.profile_read("spconv_wgrad")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms, cnt = _lib.profile_read("spconv_wgrad")
    _lib.profile_enable(False)
    return a.elapsed_time(b) / n * 1e3, ms / max(cnt, 1) * 1e3


if __name__ == "__main__":
    for C, n_target, shape in ((16, 95000, (1440, 1440, 41)), (32, 132000, (720, 720, 21)), (64, 63000, (360, 360, 11)),
                               (128, 24000, (180, 180, 5))):
        N, data = rulebook(n_target, shape, C)
        pairs = int(data.n_pairs.sum())
        for io in (0, 1):
            f = torch.randn(N, C, device=dev)
            g = torch.randn(N, C, device=dev)
            if io:
                f, g = f.to(torch.bfloat16), g.to(torch.bfloat16)
            dw = torch.empty(C, 3, 3, 3, C, device=dev)
            ws = torch.empty(lib.bfhip_spconv_wgrad_workspace_bytes(27, C, C, N), dtype=torch.uint8, device=dev)

            def run():
                rc = lib.bfhip_spconv_wgrad(f.data_ptr(), g.data_ptr(), data.pair_fwd.data_ptr(), N, 27, N, C, C, None, dw.data_ptr(), io,
                                            ws.data_ptr(), ws.numel(), _lib.stream_of(f))
                assert rc == 0
            call_us, kern_us = timed(run)
            gf = 2.0 * pairs * C * C / 1e9
            print("C=%3d rows %6d pairs %7d io16=%d  call %.1f us  main kernel %.1f us  (%.1f TF/s; reduce+launch %.1f us)" %
                  (C, N, pairs, io, call_us, kern_us, gf / kern_us * 1e-3 * 1e3, call_us - kern_us), flush=True)
