#!/usr/bin/env python3
"""Per-training-step HBM traffic of kernel families from two rocprofv3 PMC passes of bench.py:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d D -o fetch -- python3 bench.py --steps K --warmup W --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d D -o write -- python3 bench.py ... (same flags)
    python tools/pmc_step.py D/fetch_counter_collection.csv D/write_counter_collection.csv K+W > families.json
Counter values are in KiB (x 1024 B).  FETCH_SIZE is reported raw: wide 16-B/lane streaming reads are tallied at half
their bytes on gfx950 (MI355X_MICROARCH.md) and gathers are uncalibrated -- apply the correction per kernel when quoting."""
import collections
import csv
import json
import sys

FAMILIES = [
    ("spconv_wgrad", lambda n: "spconv_wgrad" in n),
    ("wgrad_reduce + offset counts", lambda n: "wgrad_reduce" in n or "wgrad_offset_counts" in n),
    ("spconv_gemm (fwd + dgrad)", lambda n: "spconv_gemm" in n),
    ("bn2d (all six kernels)", lambda n: "bn2d_" in n),
    ("bn1d", lambda n: "bn1d_" in n),
    ("attn (all)", lambda n: "attn_" in n),
    ("lift_splat_fwd", lambda n: "lift_splat_fwd" in n),
    ("lift_splat_bwd", lambda n: "lift_splat_bwd" in n),
    ("rulebook", lambda n: "sparse_" in n or "subm_" in n or "rulebook" in n),
]


def per_family(path, counter):
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for fam, pred in FAMILIES:
            if pred(r["Kernel_Name"]):
                tot[fam] += float(r["Counter_Value"]) * 1024.0
                cnt[fam] += 1
                break
    return tot, cnt


def main(fetch_csv, write_csv, steps):
    steps = float(steps)
    f, fc = per_family(fetch_csv, "FETCH_SIZE")
    w, _ = per_family(write_csv, "WRITE_SIZE")
    out = {}
    for fam, _ in FAMILIES:
        if fc.get(fam):
            out[fam] = dict(launches_per_step=round(fc[fam] / steps, 2), fetch_bytes_raw_per_step=round(f[fam] / steps, 1),
                            write_bytes_per_step=round(w.get(fam, 0.0) / steps, 1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
