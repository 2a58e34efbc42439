#!/usr/bin/env python3
"""Per-kernel timeline and per-stage sums of ONE LiDAR pass (hard voxelize + sparse encoder forward + backward) from a
rocprofv3 --kernel-trace CSV of tools/sparse_micro.py:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sp -o sp -- python3 tools/sparse_micro.py
    python3 tools/sparse_pass_trace.py gpurun_out/prof_sp/*/sp_kernel_trace.csv > profiles/rNN_sparse_pass.txt

The pass shown is the second-to-last one of the run (steady state; the last may be cut by the end of the trace)."""
import collections
import csv
import re
import sys

STAGES = [
    ("voxelize", r"vox_|voxel_"),
    ("rulebook: hash build (fill + insert)", r"fill_pair_kernel|subm_insert_kernel"),
    ("rulebook: SubM pairs + row masks + keys", r"subm_pairs"),
    ("rulebook: strided mark / count / prefix", r"sparse_mark_kernel|words_count_kernel|blocks_scan_kernel|words_prefix_kernel"),
    ("rulebook: strided output coordinates", r"sparse_out_indices_kernel"),
    ("rulebook: strided pairs + row masks + keys", r"sparse_pairs"),
    ("rulebook: row masks of the forward table", r"row_mask_kernel"),
    ("rulebook: row sorts (rocPRIM)", r"rocprim"),
    ("gather-GEMM forward / dgrad", r"spconv_gemm|spconv_scalar_kernel"),
    ("weight packing", r"pack_weights"),
    ("weight gradient (main + reduce + counts)", r"wgrad"),
    ("BatchNorm forward", r"bn2d_stats|bn2d_finalize|bn2d_apply|bn_stats|bn_finalize|bn_apply"),
    ("BatchNorm backward", r"bn2d_bwd|bn_bwd"),
    ("dense BEV map", r"to_bev|bev_to_sparse|bev_nhwc"),
    ("memset / copy (runtime)", r"__amd_rocclr"),
    ("torch elementwise / reduce / cat", r"at::native"),
]


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    starts = [i for i, n in enumerate(names) if "vox_init_kernel" in n]
    passes = []
    for i in starts:
        if not passes or i - passes[-1][-1] > 100:
            passes.append([i])
        else:
            passes[-1].append(i)
    if len(passes) < 3:
        raise SystemExit("need at least 3 passes in the trace")
    win = rows[passes[-3][0]:passes[-2][0]]
    t0 = int(win[0]["Start_Timestamp"])
    span = (int(win[-1]["End_Timestamp"]) - t0) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win) / 1e3
    print("one LiDAR pass: %d kernels, span %.1f us, kernel time %.1f us (rocprofv3 timestamps; launch gaps are inflated by "
          "the tracer)" % (len(win), span, busy))
    agg = collections.OrderedDict((s, [0, 0.0]) for s, _ in STAGES)
    agg["other"] = [0, 0.0]
    for r in win:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for s, pat in STAGES:
            if re.search(pat, r["Kernel_Name"]):
                agg[s][0] += 1
                agg[s][1] += d
                break
        else:
            agg["other"][0] += 1
            agg["other"][1] += d
    print("---- per stage")
    for s, (n, d) in agg.items():
        print("%9.1f us  %4d launches  %s" % (d, n, s))
    print("---- timeline (start us, duration us, grid, kernel)")
    for r in win:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        print("%9.1f %8.1f us  grid=%-9d %s" % ((s - t0) / 1e3, (e - s) / 1e3, grid, r["Kernel_Name"][:120]))


if __name__ == "__main__":
    main(sys.argv[1])
