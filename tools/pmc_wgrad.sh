#!/bin/bash
# SQ counters of the dense weight-gradient kernels on one layer shape (tools/conv_micro.py <layer>), both tile shapes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_wgrad; mkdir -p $O
L=${1:-ConvFuser}
for wide in 0 1; do
for pass in A B; do
  if [ $pass = A ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT";
  else C="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"; fi
  BFHIP_WGRAD_WIDE=$wide timeout -k 10 170 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p -o c -- python3 tools/conv_micro.py $L > $O/run.json 2> $O/run.err || { tail -5 $O/run.err; exit 1; }
  python3 - <<PY
import csv, collections, re
rows = list(csv.DictReader(open("$O/p/c_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    m = re.search(r"conv_wgrad\w*(<[^>]*>)?", k)
    if not m or "reduce" in k: continue
    k = m.group(0)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in agg.items():
    print("wide=$wide pass=$pass %-40s n=%3d" % (k, n[k]), " ".join("%s=%.4g" % (a.replace("SQ_", ""), b / max(n[k], 1)) for a, b in sorted(c.items())))
PY
  rm -rf $O/p
done
done
