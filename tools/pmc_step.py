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
import re
import sys

FAMILIES = [
    ("spconv_wgrad", lambda n: "spconv_wgrad" in n),
    ("wgrad_reduce + offset counts", lambda n: ("wgrad_reduce" in n and "conv_wgrad_reduce" not in n) or "wgrad_offset_counts" in n),
    ("conv_wgrad_reduce (slab sum of the sparse bf16 and the dense weight gradients)", lambda n: "conv_wgrad_reduce" in n or "conv_wgrad_group_reduce" in n),
    ("spconv_gemm (fwd + dgrad)", lambda n: "spconv_gemm" in n),
    ("bn2d backward (reduce, finalize, apply)", lambda n: "bn2d_bwd" in n or ("bn2d_finalize" in n and "BwdFin" in n)),
    ("bn2d forward (stats, finalize, apply)", lambda n: "bn2d_" in n),
    # last template argument of conv_igemm_kernel = MODE: 0 forward, 1 data gradient (transposed gather), 2 data gradient of a
    # strided layer by parity classes (round 2 matched a bool that no longer exists and booked these under the forward)
    ("conv2d pointwise dgrad (conv_pw)", lambda n: re.search(r"conv_pw_kernel<[^>]*, 1>", n) is not None),
    ("conv2d pointwise forward (conv_pw)", lambda n: "conv_pw_kernel" in n),
    ("conv2d dgrad (conv_igemm, transposed gather)", lambda n: re.search(r"conv_igemm_kernel<[^>]*, [12]>", n) is not None),
    ("conv2d forward (conv_igemm)", lambda n: "conv_igemm_kernel" in n),
    ("conv2d wgrad (main kernel)", lambda n: "conv_wgrad_kernel" in n or "conv_wgrad_wide_kernel" in n or "conv_wgrad_group_kernel" in n),
    ("conv weight transpose", lambda n: "conv_weight_transpose" in n),
    ("bn1d", lambda n: "bn1d_" in n),
    ("attn (all)", lambda n: "attn_" in n),
    ("lift_splat_fwd", lambda n: "lift_splat_fwd" in n),
    ("lift_splat_bwd", lambda n: "lift_splat_bwd" in n),
    ("rulebook", lambda n: "sparse_" in n or "subm_" in n or "row_mask" in n or "words_" in n or "blocks_scan" in n or "fill_pair" in n),
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


# bench.py op name -> kernel family above (ops whose launches can be told apart by kernel name)
BENCH_OPS = {"spconv_wgrad": ["spconv_wgrad", "wgrad_reduce + offset counts"], "lift_splat_fwd": ["lift_splat_fwd"],
             "lift_splat_bwd": ["lift_splat_bwd"], "rulebook": ["rulebook"],
             "bn2d_fwd": ["bn2d forward (stats, finalize, apply)"], "bn2d_bwd": ["bn2d backward (reduce, finalize, apply)"],
             "conv2d_fwd": ["conv2d forward (conv_igemm)"],
             "conv2d_dgrad": ["conv2d dgrad (conv_igemm, transposed gather)", "conv weight transpose"],
             "conv2d_wgrad": ["conv2d wgrad (main kernel)"],
             "conv2d_pw_fwd": ["conv2d pointwise forward (conv_pw)"], "conv2d_pw_dgrad": ["conv2d pointwise dgrad (conv_pw)"],
             "spconv_fwd+bwd": ["spconv_gemm (fwd + dgrad)"]}
# 16-byte-per-lane streaming readers: FETCH_SIZE reports half their bytes on gfx950 (MI355X_MICROARCH.md) -> doubled
STREAMING = {"bn2d_fwd", "bn2d_bwd"}


def main(fetch_csv, write_csv, steps):
    steps = float(steps)
    f, fc = per_family(fetch_csv, "FETCH_SIZE")
    w, _ = per_family(write_csv, "WRITE_SIZE")
    fam = {}
    for name, _ in FAMILIES:
        if fc.get(name):
            fam[name] = dict(launches_per_step=round(fc[name] / steps, 2), fetch_bytes_raw_per_step=round(f[name] / steps, 1),
                             write_bytes_per_step=round(w.get(name, 0.0) / steps, 1))
    kernels = {}
    for op, names in BENCH_OPS.items():
        got = [fam[n] for n in names if n in fam]
        if got:
            fetch = sum(g["fetch_bytes_raw_per_step"] for g in got)
            write = sum(g["write_bytes_per_step"] for g in got)
            if op in STREAMING:
                kernels[op] = dict(fetch_bytes_raw=fetch, write_bytes=write, hbm_bytes=2 * fetch + write,
                                   hbm_bytes_note="per training step; 2 x FETCH_SIZE (wide streaming reads are tallied at half "
                                                  "their bytes on gfx950, MI355X_MICROARCH.md) + WRITE_SIZE")
            else:
                kernels[op] = dict(fetch_bytes_raw=fetch, write_bytes=write, hbm_bytes=fetch + write,
                                   hbm_bytes_note="per training step; FETCH_SIZE raw (gather kernels: uncalibrated lower bound, "
                                                  "MI355X_MICROARCH.md) + WRITE_SIZE")
    print(json.dumps({"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate --kernel-trace passes of "
                                 "bench.py --workload full (batch 4, 40k points), summed per kernel family and divided by the "
                                 "%d steps of the run (tools/pmc_step.py)" % int(steps),
                      "_units": "counter values x 1024 B", "kernels": kernels, "full_step": fam}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
