"""Tuning artefacts for the dense (MIOpen) layers around the hot path.

`miopen_userdb/` holds MIOpen's own user find-db / perf-db text files for gfx950, produced by running
bench.py once with BENCH_MIOPEN_FIND=1 MIOPEN_FIND_MODE=3 on an MI355X (the exhaustive find takes ~3 extra
minutes on a fresh box).  `use_shipped_miopen_db()` points MIOpen at a writable copy so that the immediate
mode picks the tuned solutions without running find: full workload 76.5 -> 64.5 ms/step on a box with empty
caches, and the first-step warm-up drops from ~90 s to ~15 s.  Call it before the first convolution runs.
"""
import os
import shutil
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))


def use_shipped_miopen_db():
    if os.environ.get("MIOPEN_USER_DB_PATH") or os.environ.get("BFHIP_NO_SHIPPED_MIOPEN_DB") == "1":
        return os.environ.get("MIOPEN_USER_DB_PATH")
    src = os.path.join(_HERE, "miopen_userdb")
    # one writable copy per local rank: ranks of one node must not share MIOpen's db files
    dst = os.path.join(tempfile.gettempdir(), "bfhip_miopen_userdb_%d_r%s" % (os.getuid(), os.environ.get("LOCAL_RANK", "0")))
    try:
        os.makedirs(dst, exist_ok=True)
        for f in os.listdir(src):
            target = os.path.join(dst, f)
            if not os.path.exists(target):
                shutil.copy(os.path.join(src, f), target)
        os.environ["MIOPEN_USER_DB_PATH"] = dst
        return dst
    except OSError:
        return None
