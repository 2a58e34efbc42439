#!/bin/bash
# kernel traces of the last step, previous round's tree (_r02) vs this tree, on one box -> gpurun_out/trace_ab/{old,new}.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$PWD/gpurun_out/trace_ab; mkdir -p $O
for tag in old new; do
  d=.; [ $tag = old ] && d=_r02
  ( cd $d && BENCH_REFERENCE_NUMERICS=0 BENCH_ISOLATED_STEPS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o t -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err ) || { tail -5 $O/$tag.err; exit 1; }
  f=$(find $O/prof_$tag -name 't_kernel_trace.csv' | head -1)
  python3 tools/trace_step.py $f 60 > $O/$tag.txt
  cp $(find $O/prof_$tag -name 't_kernel_stats.csv' | head -1) $O/${tag}_kernel_stats.csv
  rm -rf $O/prof_$tag
done
