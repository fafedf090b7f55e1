for v in 1 0 1 0; do CAPDEC_LN_BWD_STAGED=$v timeout -k 10 200 python3 bench.py --no-variants --no-cpu-baseline --steps 3 --train-steps 20 2>/dev/null > gpurun_out/ab_$v.json || exit 1; python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab_$v.json')); print('staged=$v', d['train']['value'], d['train']['ms_per_step'])"; done
