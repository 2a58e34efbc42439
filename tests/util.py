"""Shared helpers for the test-suite (test infrastructure; may import the oracle)."""
import hashlib

import numpy as np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_bev_pool_case(n, c, b, d, h, w, seed, integer_valued=False, long_tail=False):
    """Random sorted-rank bev_pool input: x f32[n,c], geom i32[n,4]=(x,y,z,b), ranks i64[n]."""
    rng = np.random.default_rng(seed)
    cells = b * d * h * w
    if long_tail:
        # a few very long intervals + many short ones, like the near-field BEV cells
        hot = rng.integers(0, cells, max(1, cells // 200))
        pick = np.where(rng.random(n) < 0.3, rng.choice(hot, n), rng.integers(0, cells, n))
    else:
        pick = rng.integers(0, cells, n)
    # rank = x*(W*D*B) + y*(D*B) + z*B + b   (BF/depth_lss.py:165-169), H=nx[0] (x), W=nx[1] (y)
    gx = pick // (w * d * b)
    rem = pick % (w * d * b)
    gy = rem // (d * b)
    rem = rem % (d * b)
    gz = rem // b
    gb = rem % b
    ranks = np.sort(pick).astype(np.int64)
    order = np.argsort(pick, kind="stable")
    geom = np.stack([gx, gy, gz, gb], 1)[order].astype(np.int32)
    if integer_valued:
        x = rng.integers(-8, 9, (n, c)).astype(np.float32)
    else:
        x = rng.standard_normal((n, c)).astype(np.float32)
    return x, geom, ranks


def rel_err(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    denom = np.maximum(np.abs(want).max(), 1e-30)
    return np.abs(got - want).max() / denom
