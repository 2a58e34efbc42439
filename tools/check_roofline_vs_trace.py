#!/usr/bin/env python3
"""Does the bench line reproduce from the kernel trace of the same run?

    rocprofv3 --kernel-trace --stats --output-format csv -d D -o full -- python3 bench.py --steps K --warmup W --no-cpu-baseline > line.json
    python3 tools/check_roofline_vs_trace.py line.json D/.../full_kernel_trace.csv > profiles/rNN_roofline_vs_trace.txt

For every entry of `roofline_ops` the per-step time bench.py printed (HIP events of the library around the op: `kernel_ms_per_step`
from the one-queue pass for the ops of the LiDAR branch, `ms_per_step` of the timed region for the rest) is set beside the same
op's kernels in the trace: sum of kernel durations / number of steps, taken over the part of the trace the bench figure was
measured in (the one-queue pass = the optimizer steps after the timed region; the timed region = the K steps before it).
An event pair brackets the whole op, so it also sees the launch gaps between the op's kernels: `gap` columns show how much.
"""
import collections
import csv
import json
import re
import sys

# bench op -> regexes of its kernels
FAMILIES = {
    "hard_voxelize": [r"vox_", r"voxel_"],
    "spconv_fwd+spconv_bwd": [r"spconv_gemm", r"spconv_scalar_kernel", r"pack_weights"],
    "spconv_wgrad_main": [r"spconv_wgrad"],
    "rulebook": [r"fill_pair_kernel", r"subm_insert_kernel", r"subm_pairs", r"sparse_mark_kernel", r"words_count_kernel",
                 r"blocks_scan_kernel", r"words_prefix_kernel", r"sparse_out_indices_kernel", r"sparse_pairs", r"row_mask_kernel",
                 r"merge_sort|block_sort|wrapped_merge", r"sort_rows_chunk_kernel"],
    "lift_splat_fwd": [r"lift_splat_fwd"],
    "lift_splat_bwd": [r"lift_splat_bwd"],
    "conv2d_fwd": [r"conv_igemm_kernel<[^>]*, 0>"],
    "conv2d_dgrad": [r"conv_igemm_kernel<[^>]*, [12]>", r"conv_weight_transpose_kernel"],
    "conv2d_pw_fwd": [r"conv_pw_kernel<[^>]*, 0>"],
    "conv2d_pw_dgrad": [r"conv_pw_kernel<[^>]*, 1>"],
    "conv2d_wgrad": [r"conv_wgrad_kernel", r"conv_wgrad_wide_kernel", r"conv_wgrad_reduce_kernel", r"conv_wgrad_group_kernel",
                     r"conv_wgrad_group_reduce_kernel"],
    "bn2d_fwd": [r"bn2d_stats", r"bn2d_finalize_kernel<[^>]*FwdFin", r"bn2d_apply"],
    "bn2d_bwd": [r"bn2d_bwd", r"bn2d_finalize_kernel<[^>]*BwdFin"],
}
LIDAR = ("hard_voxelize", "spconv_fwd+spconv_bwd", "spconv_wgrad_main", "rulebook")


def main(line_path, trace_path, plain_path=None):
    line = json.loads(open(line_path).read().strip().splitlines()[-1])
    # optional: the bench line of a run WITHOUT the tracer (same code, same workload): event scopes are a few percent wider under
    # rocprofv3, kernel durations are not, so the plain run's figures are the ones to hold against the kernel sums
    plain = {}
    if plain_path:
        pl = json.loads(open(plain_path).read().strip().splitlines()[-1])
        plain = {r["kernel"]: r.get("kernel_ms_per_step", r["ms_per_step"]) for r in pl["roofline_ops"]}
        if "spconv_fwd" in plain and "spconv_bwd" in plain:
            plain["spconv_fwd+spconv_bwd"] = plain["spconv_fwd"] + plain["spconv_bwd"]
    rows = list(csv.DictReader(open(trace_path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # optimizer steps = clusters of the fused AdamW launches
    adam = [i for i, n in enumerate(names) if "adamw_update_kernel" in n or "FusedOptimizer" in n or ("multi_tensor_apply" in n and "Adam" in n)]
    ends = []
    for i in adam:
        if ends and i - ends[-1] < 50:
            ends[-1] = i
        else:
            ends.append(i)
    K, W = line["steps"], line["warmup"]
    iso = len(ends) - K - W
    print("trace: %d kernels, %d optimizer steps = %d warm-up + %d timed + %d one-queue steps after the timed region" % (len(rows), len(ends), W, K, iso))
    assert iso >= 0, "fewer optimizer steps in the trace than the bench line claims"
    timed = (ends[W - 1] + 1 if W else 0, ends[W + K - 1] + 1)
    after = (ends[W + K - 1] + 1, ends[-1] + 1) if iso else None
    ops = {r["kernel"]: r for r in line["roofline_ops"]}
    bench = {}
    for op, r in ops.items():
        bench[op] = (r.get("kernel_ms_per_step", r["ms_per_step"]), "one-queue pass" if "kernel_ms_per_step" in r else "timed region")
    if "spconv_fwd" in bench and "spconv_bwd" in bench:
        bench["spconv_fwd+spconv_bwd"] = (bench["spconv_fwd"][0] + bench["spconv_bwd"][0], bench["spconv_fwd"][1])
    # the slab-sum kernel serves the sparse weight gradient too (20 of its launches per step): it is booked under conv2d_wgrad
    # here and under spconv_wgrad in bench.py; spconv_wgrad_main (main kernel only) is the comparable sparse figure
    print("columns: bench ms = HIP-event scope per step as printed by bench.py; kernels ms = sum of the op's kernel durations per step;\n"
          "span ms = per op instance, first kernel start -> last kernel end (what an event pair brackets: kernel time + the dispatch gaps\n"
          "between the op's own kernels, which tracing itself widens), summed per step; ratio = bench / span")
    print("%-24s %-15s %10s %10s %10s %7s %11s %7s %9s   %s" % ("op", "bench figure", "bench ms", "kernels ms", "span ms", "ratio", "plain-run ms", "/kern", "launches", "kernels"))
    worst = worst_plain = 0.0
    neutral = re.compile(r"__amd_rocclr_(fill|copy)Buffer")
    for op, pats in FAMILIES.items():
        if op not in bench:
            continue
        b, where = bench[op]
        lo, hi = (after if (where == "one-queue pass" and after) else timed)
        n_steps = iso if (where == "one-queue pass" and after) else K
        tot, cnt = 0, 0
        kn = collections.Counter()
        for r in rows[lo:hi]:
            if any(re.search(p, r["Kernel_Name"]) for p in pats):
                tot += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                cnt += 1
                kn[re.sub(r".*::", "", r["Kernel_Name"].split("(")[0])[:28]] += 1
        # spans: per queue, maximal runs of the op's kernels (runtime memset / copy kernels in between do not break a run)
        span = 0
        by_q = collections.defaultdict(list)
        for r in rows[lo:hi]:
            by_q[r.get("Queue_Id", "0")].append(r)
        for q_rows in by_q.values():
            first = last = None
            pending = 0  # runtime fill / copy kernels seen since the run's last own kernel: the op's own memsets when the run goes on
            for r in q_rows:
                fam = any(re.search(p, r["Kernel_Name"]) for p in pats)
                if fam:
                    first = first if first is not None else int(r["Start_Timestamp"])
                    last = int(r["End_Timestamp"])
                    tot += pending
                    pending = 0
                elif neutral.search(r["Kernel_Name"]):
                    if first is not None:
                        pending += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                elif first is not None:
                    span += last - first
                    first = last = None
                    pending = 0
            if first is not None:
                span += last - first
        t = tot / 1e6 / n_steps  # own kernels + the runtime memsets between them
        sp = span / 1e6 / n_steps
        if op == "conv2d_wgrad" and "spconv_wgrad_main" in bench:  # remove the sparse layers' share of the shared slab-sum kernel
            red = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[lo:hi] if "conv_wgrad_reduce_kernel" in r["Kernel_Name"]]
            per = len(red) / n_steps
            # every dense per-layer launch has exactly one slab sum behind its main kernel (the grouped launch has its own reduce kernel)
            dense = sum(1 for r in rows[lo:hi] if re.search(r"conv_wgrad_kernel|conv_wgrad_wide_kernel", r["Kernel_Name"])) / n_steps
            if per > dense:
                cut = sum(red) / 1e6 / n_steps * (per - dense) / per
                t -= cut
                sp -= cut
        ratio = b / sp if sp else float("nan")
        worst = max(worst, abs(ratio - 1.0)) if sp else worst
        pr = plain.get(op)
        if pr is not None and t:
            worst_plain = max(worst_plain, abs(pr / t - 1.0))
        print("%-24s %-15s %10.4f %10.4f %10.4f %7.2f %11s %7s %9.1f   %s" % (
            op, where, b, t, sp, ratio, "%.4f" % pr if pr is not None else "-", "%.2f" % (pr / t) if (pr is not None and t) else "-",
            cnt / n_steps, ", ".join("%s x%d" % (k, v // n_steps) for k, v in kn.most_common(4))))
    print("largest deviation of a bench figure (this traced run) from its span in the trace: %.1f %%" % (100 * worst))
    if plain:
        print("largest deviation of a plain-run bench figure from the sum of its kernels' durations in the trace: %.1f %%" % (100 * worst_plain))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
