#!/usr/bin/env python3
"""Per-wave timeline of the 64 x 64 wgrad kernel (library built with BFHIP_EXTRA_FLAGS=-DBFHIP_WGRAD_TRACE)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bevfusion_amd  # noqa
from bevfusion_amd import _lib, spconv as sp
sys.path.insert(0, os.path.dirname(__file__))
from wgrad_micro2 import rulebook  # noqa  (runs the micro as a side effect: fine)

dev = torch.device("cuda:0")
lib = _lib.load()


def trace(C, N, pair_fwd, n_pairs, label):
    f = torch.randn(N, C, device=dev).to(torch.bfloat16)
    g = torch.randn(N, C, device=dev).to(torch.bfloat16)
    dw = torch.empty(C, 3, 3, 3, C, device=dev)
    wsb = lib.bfhip_spconv_wgrad_workspace_bytes(27, C, C, N)
    ws = torch.zeros(wsb + (1 << 22), dtype=torch.uint8, device=dev)
    ld = pair_fwd.shape[1]
    for _ in range(3):
        lib.bfhip_spconv_wgrad(f.data_ptr(), g.data_ptr(), pair_fwd.data_ptr(), ld, 27, N, C, C, None, dw.data_ptr(), 1,
                               ws.data_ptr(), wsb, _lib.stream_of(f))
    torch.cuda.synchronize()
    GI = GJ = (C + 63) // 64
    occ = int(os.environ["BFHIP_WGRAD_BLOCKS_PER_CU"])   # set it: the library's own choice is not visible from here
    P = occ * torch.cuda.get_device_properties(0).multi_processor_count
    T, Ut = 27 * GI * GJ, (N + 63) // 64
    NR = (4 if GI * GJ == 4 else 8) if Ut >= 64 else 1       # the library's rule (spconv.hip, bfhip_spconv_wgrad)
    Ur = -(-Ut // NR)
    P = min(P, NR * T * Ur)
    nwaves, S = P * 4, -(-NR * T * Ur // P)
    off = (P + NR * T) * 4096 * 4 + 64 * 32 * 4
    tr = ws[off:off + nwaves * 48].cpu().numpy().view(np.uint64).reshape(nwaves, 6).astype(np.int64)
    t0 = tr[:, 0].min()
    st, t1, t2, en, cnt, kk = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0, (tr[:, 3] - t0) / 100.0, tr[:, 4], tr[:, 5] >> 32
    live = cnt > 0
    npk = n_pairs.cpu().numpy()
    print("%s C=%d rows %d pairs %d (per offset: min %d median %d max %d) units/block=%d waves %d  kernel span %.1f us  (MFMA-bound %.1f us)" %
          (label, C, N, npk.sum(), npk.min(), np.median(npk), npk.max(), S, nwaves, en.max(), 2.0 * npk.sum() * C * C / 157.3e6))
    print("  wave duration us: mean %.1f  p50 %.1f  p90 %.1f  max %.1f" % ((en - st).mean(), np.median(en - st), np.percentile(en - st, 90), (en - st).max()))
    print("  compaction (start -> list ready): mean %.2f us;  epilogue (loop end -> done): mean %.2f us" % ((t1 - st)[live].mean(), (en - t2).mean()))
    loop = (t2 - t1)[live]
    steps = np.ceil(cnt[live] / 4.0)
    print("  pipeline loop: mean %.2f us for mean %.0f K-steps -> %.1f ns per K-step (one wave alone on the MFMA: %.1f ns at 2.4 GHz)" %
          (loop.mean(), steps.mean(), 1e3 * loop.sum() / steps.sum(), 512 / 2.4))
    span = en.max()
    for i in range(6):
        lo, hi = span * i / 6, span * (i + 1) / 6
        print("  %5.1f-%5.1f us: waves alive %5d, starting %5d" % (lo, hi, ((st < hi) & (en > lo)).sum(), ((st >= lo) & (st < hi)).sum()))
    if "--per-offset" in sys.argv:
        for k in range(27):
            m = kk == k
            if m.any():
                print("    k=%2d waves %3d  pairs/wave mean %5.0f (min %4d max %4d)  duration mean %6.1f us (min %5.1f max %5.1f)" %
                      (k, m.sum(), cnt[m].mean(), cnt[m].min(), cnt[m].max(), (en - st)[m].mean(), (en - st)[m].min(), (en - st)[m].max()))
    ctr = kk == 13
    print("  centre offset waves: %d, duration mean %.1f us; others duration mean %.1f us" % (ctr.sum(), (en - st)[ctr].mean(), (en - st)[~ctr].mean()))


if "--real" in sys.argv:
    _argv = list(sys.argv)
    sys.argv = [sys.argv[0]]
    import bench
    wl = bench.LidarOnly(dev, 4, 40000)
    seen = {}
    orig = sp._SparseConvFunction.forward

    def spy(ctx, features, weight, data, n_in):
        key = (weight.shape[-1], weight.shape[0])
        if key in ((64, 64), (128, 128)) and key not in seen and data.pair_fwd.shape[0] == 27:
            seen[key] = (n_in, data)
        return orig(ctx, features, weight, data, n_in)

    sp._SparseConvFunction.forward = staticmethod(spy)
    with torch.no_grad():
        wl.model.extract_pts_feat(wl.inputs)
    sp._SparseConvFunction.forward = staticmethod(orig)
    for (C, _), (n_in, data) in sorted(seen.items()):
        sys.argv = _argv
        trace(C, n_in, data.pair_fwd, data.n_pairs, "encoder layer")
else:
    for C, n_target, shape in ((64, 63000, (360, 360, 11)), (128, 24000, (180, 180, 5))):
        N, data = rulebook(n_target, shape, C)
        trace(C, N, data.pair_fwd, data.n_pairs, "synthetic")
