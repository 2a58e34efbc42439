#!/usr/bin/env python3
"""Summarise the LAST optimizer step of a rocprofv3 --kernel-trace CSV of bench.py (steady state)."""
import collections
import csv
import re
import sys


def category(n):
    if "adamw_" in n:
        return "optimizer"
    if "bfhip" in n:
        return "bfhip:" + re.sub(r".*::", "", n.split("(")[0])[:34]
    low = n.lower()
    if n.startswith("igemm") or "conv" in low or "Cijk" in n or "gemm" in low:
        return "conv/gemm " + ("fp32" if ("fp32" in n or "float" in n or "_S_" in n) else "bf16" if ("bf16" in n or "ushort" in n or "BBS" in n or "_B_" in n) else "?")
    if "atchNorm" in n or "batch_norm" in n:
        return "batchnorm"
    if "attn" in n or "bwd_kernel" in n:
        return "attention"
    if "multi_tensor" in n or "adamw_" in n:
        return "optimizer"
    if "elementwise" in n or "SubTensor" in n or "copy" in low or "fill" in low:
        return "elementwise/copy/fill"
    if "rocprim" in n:
        return "rocprim"
    return "other"


def main(path, top=45, which=-1):
    """which: optimizer step to summarise, counted from the end (-1 = last; bench.py appends BENCH_ISOLATED_STEPS one-queue
    steps behind the timed region, so the last TIMED step is -(isolated + 1))."""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    adam = [i for i, n in enumerate(names) if "adamw_update_kernel" in n]  # the flat optimizer step (csrc/optim.hip) ends a step
    if not adam:
        adam = [i for i, n in enumerate(names) if "multi_tensor_apply" in n]
    clusters = []
    for i in adam:
        if not clusters or i - clusters[-1][-1] > 50:
            clusters.append([i])
        else:
            clusters[-1].append(i)
    win = rows[clusters[which - 1][-1] + 1:clusters[which][-1] + 1]
    t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["End_Timestamp"])
    cat = collections.defaultdict(lambda: [0, 0])
    ker = collections.defaultdict(lambda: [0, 0])
    for r in win:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for table, key in ((cat, category(r["Kernel_Name"])), (ker, r["Kernel_Name"][:110])):
            table[key][0] += d
            table[key][1] += 1
    busy = sum(v[0] for v in cat.values())
    print("step window %.2f ms, GPU busy %.2f ms, %d kernels" % ((t1 - t0) / 1e6, busy / 1e6, len(win)))
    # per queue: busy time; over all queues: time covered by at least one kernel (the rest of the window is idle GPU)
    per_q = collections.defaultdict(lambda: [0, 0])
    for r in win:
        q = r.get("Queue_Id", "?")
        per_q[q][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        per_q[q][1] += 1
    for q, (d, n) in sorted(per_q.items(), key=lambda kv: -kv[1][0]):
        print("queue %-6s n=%5d busy=%8.3f ms" % (q, n, d / 1e6))
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win)
    cov, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    gaps = []
    for a, b in iv[1:]:
        if a > cur_e:
            cov += cur_e - cur_s
            gaps.append(a - cur_e)
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    cov += cur_e - cur_s
    gaps.sort()
    print("covered by >= 1 kernel: %.2f ms; idle: %.2f ms in %d gaps (median %.1f us, p90 %.1f us, max %.1f us)"
          % (cov / 1e6, ((t1 - t0) - cov) / 1e6, len(gaps), gaps[len(gaps) // 2] / 1e3 if gaps else 0,
             gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0, gaps[-1] / 1e3 if gaps else 0))
    for k, (d, n) in sorted(cat.items(), key=lambda kv: -kv[1][0]):
        print("%-48s n=%5d ms=%8.3f" % (k, n, d / 1e6))
    print("---- top kernels")
    for k, (d, n) in sorted(ker.items(), key=lambda kv: -kv[1][0])[:top]:
        print("%-110s n=%4d ms=%7.3f" % (k, n, d / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45, int(sys.argv[3]) if len(sys.argv) > 3 else -1)
