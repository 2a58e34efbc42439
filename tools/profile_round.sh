#!/bin/bash
# Regenerates the measurements behind profiles/rNN_* on the MI355X box (run through gpurun from the repo root):
#   bash tools/profile_round.sh A|B|C|D  -> gpurun_out/rnd/*   (four parts so that one call stays well inside the time limit)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/rnd; mkdir -p $O
part=${1:-A}
run() { echo "== $*" >&2; "$@"; }
if [ "$part" = A ]; then
  run timeout -k 10 300 python3 bench.py > $O/bench_full.json 2> $O/bench_full.err || exit 1
  for w in lidar_only lidar_branch camera_only hotpath_v1; do
    run timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err || exit 1
  done
  run timeout -k 10 200 python3 bench.py --points 250000 --no-cpu-baseline > $O/bench_full_250k_points.json 2> $O/bench_250k.err || exit 1
  run timeout -k 10 200 python3 bench.py --vt-fp32 --no-cpu-baseline > $O/bench_full_vt_fp32.json 2> $O/bench_vt.err || exit 1
  BENCH_GRAPH=1 run timeout -k 10 200 python3 bench.py --no-cpu-baseline > $O/bench_full_graph.json 2> $O/bench_graph.err || exit 1
  BFHIP_SPCONV_SORT=0 run timeout -k 10 200 python3 bench.py --no-cpu-baseline > $O/bench_full_nosort.json 2> $O/bench_nosort.err || exit 1
elif [ "$part" = B ]; then
  export BENCH_REFERENCE_NUMERICS=0
  export BENCH_PREWARM_STEPS=0   # the trace tools count optimizer steps: exactly warm-up + timed + one-queue steps
  # the same command without the tracer first: its figures are what the kernel sums of the trace are held against
  run timeout -k 10 300 python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline > $O/bench_full_plain_same_box.json 2> $O/plain.err || exit 1
  run timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_full -o full -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline > $O/bench_full_under_rocprof.json 2> $O/prof_full.err || exit 1
  T=$(find $O/prof_full -name 'full_kernel_trace.csv' | head -1); S=$(find $O/prof_full -name 'full_kernel_stats.csv' | head -1)
  cp $S $O/bench_full_kernel_stats.csv
  python3 tools/trace_step.py $T 60 -6 > $O/bench_full_last_step_breakdown.txt   # last TIMED step (5 one-queue steps follow it)
  python3 tools/check_roofline_vs_trace.py $O/bench_full_under_rocprof.json $T $O/bench_full_plain_same_box.json > $O/roofline_vs_trace.txt
  rm -rf $O/prof_full
  run timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $O/prof_sp -o sp -- python3 tools/sparse_micro.py > $O/sparse_micro_under_rocprof.json 2> $O/prof_sp.err || exit 1
  python3 tools/sparse_pass_trace.py $(find $O/prof_sp -name 'sp_kernel_trace.csv' | head -1) > $O/sparse_pass.txt
  rm -rf $O/prof_sp
  run timeout -k 10 200 python3 tools/sparse_micro.py > $O/sparse_micro.json 2> $O/sparse_micro.err || exit 1
  run timeout -k 10 300 python3 tools/conv_micro.py > $O/conv_layers.json 2> $O/conv_layers.err || exit 1
  run timeout -k 10 200 python3 tools/gemm_micro.py > $O/sparse_gemm_layers.txt 2> $O/gemm_micro.err || exit 1
  run timeout -k 10 200 python3 tools/wgrad_layers.py > $O/sparse_layers.json 2> $O/wgrad_layers.err || exit 1
elif [ "$part" = D ]; then
  run timeout -k 10 280 python3 tools/wgrad2d_layers.py > $O/conv_wgrad_layers.txt 2> $O/wgrad2d.err || exit 1
  run timeout -k 10 200 python3 tools/wgrad_group_micro.py > $O/conv_wgrad_group.json 2> $O/wgrad_group.err || exit 1
  run timeout -k 10 280 python3 tools/bn2d_layers.py > $O/bn2d_layers.txt 2> $O/bn2d.err || exit 1
  run timeout -k 10 200 python3 tools/lift_splat_variants.py > $O/lift_splat_variants.json 2> $O/ls.err || exit 1
  run bash tools/pmc_lift_splat.sh > $O/lift_splat_counters.txt 2>&1 || exit 1
  run bash tools/pmc_wgrad.sh ConvFuser > $O/conv_wgrad_counters.txt 2>&1 || exit 1
elif [ "$part" = C ]; then
  # hardware counters: one counter per pass, kernel trace only (the pool refuses / hangs on wider combinations)
  export BENCH_NO_WORK=1
  export BENCH_PREWARM_STEPS=0   # counter totals are divided by the 5 steps of the run
  run timeout -k 10 170 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o fetch -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
  run timeout -k 10 170 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o write -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
  python3 tools/pmc_step.py $(find $O/pmc_f -name 'fetch_counter_collection.csv' | head -1) $(find $O/pmc_w -name 'write_counter_collection.csv' | head -1) 5 > $O/pmc_traffic.json
  rm -rf $O/pmc_f $O/pmc_w
fi
ls -la $O | tail -n 30
