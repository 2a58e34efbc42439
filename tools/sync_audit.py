#!/usr/bin/env python3
"""Lists every synchronising call of one steady-state `full` training step (torch.cuda.set_sync_debug_mode("warn")):
what stands between the step and a whole-step hipGraph capture."""
import os
import sys
import traceback
import warnings

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    wl = bench.FullModel(dev, 4, 40000)
    for _ in range(4):
        wl.step()
    torch.cuda.synchronize()
    seen = {}

    def showwarning(message, category, filename, lineno, file=None, line=None):
        stack = [f for f in traceback.extract_stack() if ROOT in f.filename and "sync_audit" not in f.filename]
        key = (str(message)[:80], tuple((os.path.relpath(f.filename, ROOT), f.lineno) for f in stack[-3:]))
        seen[key] = seen.get(key, 0) + 1

    warnings.showwarning = showwarning
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    wl.step()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    for (msg, where), n in seen.items():
        print(n, "x", msg, "@", " <- ".join("%s:%d" % w for w in reversed(where)))
    print("total distinct sync sites:", len(seen))


if __name__ == "__main__":
    main()
