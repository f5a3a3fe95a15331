#!/bin/bash
# usage: bible_ab.sh tag [libpath] [opts] -- the config-3 stand-in (shakespeare x 4, vocab 10,000) with a variant library / options
tag=$1; lib=$2; opts=$3
MBPE_LIB=$lib MBPE_BENCH_OPTS=$opts python bench.py --config bible --steps 10 --warmup 2 --no-cpu-baseline --no-full-run > gpurun_out/r4_bab_$tag.json 2> gpurun_out/r4_bab_$tag.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_bab_$tag.json').read().strip().splitlines()[-1])
b=d['batches']
print('$tag', 'ms/training %.2f'%d['ms_per_step'], 'value %.0f'%d['value'], 'passes', b['n_batches'], 'skipped', b['n_skipped'], 'skip_cut', b['n_skip_cut'], 'drops', b['n_validation_drops'], 'fused dropped', b['n_fused_dropped'], 'ok', (d.get('checks') or {}).get('ok'))
PY
