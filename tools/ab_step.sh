set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/ab_$name.json 2>gpurun_out/ab_$name.err; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$name.json').read().strip().splitlines()[-1])
print('$name', d['ms_per_step'], {r['kernel']:round(r['ms_per_step'],2) for r in d['roofline_ops']})
"; }
run base X=1
run nofold BFHIP_BN2D_FOLD=0
run sort BFHIP_SPCONV_SORT=1
run nofold_sort BFHIP_BN2D_FOLD=0 BFHIP_SPCONV_SORT=1
run base2 X=1
