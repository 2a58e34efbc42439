#!/bin/bash
# A/B of the step time against another tree on ONE box (boxes differ by up to 10 %): `_r02/` is a git worktree of the commit to
# compare with, with its own built .so (`git worktree add -f _r02 <commit> && (cd _r02 && python bevfusion-3d_object_detection_amd/_build.py)`;
# add `_r02/` to .gitignore while it exists; `git worktree remove --force _r02` afterwards), runs alternate old / new.
#   bash tools/ab_rounds.sh [N pairs] [extra bench args]
set -u
n=${1:-2}; shift || true
O=$PWD/gpurun_out/ab; mkdir -p $O
one() { # dir tag
  ( cd $1 && BENCH_REFERENCE_NUMERICS=0 BENCH_ISOLATED_STEPS=0 timeout -k 10 240 python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline "${@:3}" > $O/$2.json 2> $O/$2.err ) || { tail -5 $O/$2.err; return 1; }
  python3 - <<PY
import json
d=json.loads(open('$O/$2.json').read().strip().splitlines()[-1])
print('$2', d['ms_per_step'], {r['kernel']: round(r['ms_per_step'], 2) for r in d['roofline_ops']})
PY
}
for i in $(seq 1 $n); do
  one _r02 old$i "$@" || exit 1
  one . new$i "$@" || exit 1
done
