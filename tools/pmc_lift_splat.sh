#!/bin/bash
# SQ / TA / TCP counters of the fused lift-splat kernels (bf16 features, bf16 BEV map: the `full` workload's variant)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_ls; mkdir -p $O
python3 tools/lift_splat_variants.py > $O/variants.json 2> $O/variants.err || { tail -5 $O/variants.err; exit 1; }
cat $O/variants.json
for pass in A B C; do
  case $pass in
    A) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAVES";;
    B) C="TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum";;
    C) C="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM";;
  esac
  LS_ONLY=feat_bf16_out_bf16 timeout -k 10 170 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p -o c -- python3 tools/lift_splat_variants.py > $O/run_$pass.json 2> $O/run_$pass.err || { tail -3 $O/run_$pass.err; continue; }
  python3 - <<PY
import csv, collections, glob, re
f = glob.glob("$O/p/**/c_counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
first = None
for r in rows:
    k = r["Kernel_Name"]
    m = re.search(r"lift_splat_(fwd|bwd)\w*(<\w+>)?", k)
    if not m: continue
    k = m.group(0)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    first = first or r["Counter_Name"]
    if r["Counter_Name"] == first: n[k] += 1
for k, c in agg.items():
    print("pass=$pass %-34s n=%3d" % (k, n[k]), " ".join("%s=%.4g" % (a, b / max(n[k], 1)) for a, b in sorted(c.items())))
PY
  rm -rf $O/p
done
