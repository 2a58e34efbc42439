set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/ab_$name.json 2>gpurun_out/ab_$name.err || { tail -5 gpurun_out/ab_$name.err; exit 1; }; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$name.json').read().strip().splitlines()[-1])
print('$name', d['ms_per_step'], {r['kernel']:round(r['ms_per_step'],2) for r in d['roofline_ops']})
"; }
# usage: bash tools/ab_step.sh NAME=ENVVAR=VALUE ... (same box, back to back); without arguments: the default twice
if [ $# -eq 0 ]; then set -- base=X=1 base2=X=1; fi
for spec in "$@"; do run "${spec%%=*}" "${spec#*=}"; done
