#!/usr/bin/env python3
"""Forward gather-GEMM time per distinct sparse layer of the encoder (real rulebooks of the synthetic batch of 4), fp32 and
bf16 feature storage: the library's profiler scope around the main kernel.  BFHIP_GEMM_V selects the kernel variant while
two are kept for A/B."""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
from bevfusion_amd import _lib, spconv as sp  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.LidarOnly(dev, 4, 40000)
seen = {}
orig = sp._SparseConvFunction.forward


def spy(ctx, features, weight, data, n_in):
    key = (weight.shape[-1], weight.shape[0], tuple(data.pair_fwd.shape))
    if key not in seen:
        seen[key] = (n_in, data, weight.detach())
    return orig(ctx, features, weight, data, n_in)


sp._SparseConvFunction.forward = staticmethod(spy)
with torch.no_grad():
    wl.model.extract_pts_feat(wl.inputs)
sp._SparseConvFunction.forward = staticmethod(orig)

tot = {0: 0.0, 1: 0.0}
for (cin, cout, shape), (n_in, data, w) in seen.items():
    if cin % 8:
        continue
    n_out = data.pair_fwd.shape[1]
    pairs = int(data.n_pairs.sum())
    for io16 in (0, 1):
        x = torch.randn(n_in, cin, device=dev)
        if io16:
            x = x.to(torch.bfloat16)

        def run():
            return sp._gemm(x, w, data.pair_fwd, n_out, False, False, data.perm_fwd, data.mask_fwd, bf16=True, io16=bool(io16))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        _lib.profile_read("spconv_fwd")
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        ms, cnt = _lib.profile_read("spconv_fwd")
        _lib.profile_enable(False)
        us = ms / max(cnt, 1) * 1e3
        tot[io16] += us
        print("Cin %3d Cout %3d KV %2d rows %6d pairs %7d io16=%d  %.1f us  (%.1f TFLOP/s)" %
              (cin, cout, shape[0], n_out, pairs, io16, us, 2.0 * pairs * cin * cout / us * 1e-6), flush=True)
print("sum over distinct layers: fp32 storage %.1f us, bf16 storage %.1f us" % (tot[0], tot[1]))
