#!/bin/bash
# SQ counters of the dense conv kernels on one layer shape (tools/conv_micro.py <layer>): where the wave cycles go
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_conv; mkdir -p $O
L=${1:-ConvFuser}
timeout -k 10 170 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/p -o c -- python3 tools/conv_micro.py $L > $O/run.json 2> $O/run.err || { tail -5 $O/run.err; exit 1; }
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/p/c_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:8]:
    w = c.get("SQ_WAVE_CYCLES", 1)
    print("%-60s n=%3d" % (k, n[k]), " ".join("%s=%.3g" % (a.replace("SQ_", ""), b / max(n[k], 1)) for a, b in sorted(c.items())))
    print("    wait_any/wave %.2f  wait_inst/wave %.2f  (lds part %.2f)  active/wave %.2f  mfma_busy/busy %.3f  lds_conflict/wave %.3f" % (
        c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w, c.get("SQ_WAIT_INST_LDS", 0) / w, c.get("SQ_ACTIVE_INST_ANY", 0) / w,
        c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("SQ_BUSY_CYCLES", 1), 1), c.get("SQ_LDS_BANK_CONFLICT", 0) / w))
PY
rm -rf $O/p
