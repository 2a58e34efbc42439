#!/bin/bash
# same-box A/B of environment switches on THIS tree, alternating:  bash tools/ab_env.sh N "A_ENV=1 ..." "B_ENV=1 ..."
set -u
n=${1:-2}; a="$2"; b="$3"
O=$PWD/gpurun_out/abenv; mkdir -p $O
one() { # tag envs
  ( env $2 BENCH_REFERENCE_NUMERICS=0 BENCH_ISOLATED_STEPS=0 timeout -k 10 240 python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline > $O/$1.json 2> $O/$1.err ) || { tail -5 $O/$1.err; return 1; }
  python3 - <<PY
import json
d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1])
print('$1 [$2]', d['ms_per_step'])
PY
}
for i in $(seq 1 $n); do one a$i "$a" || exit 1; one b$i "$b" || exit 1; done
