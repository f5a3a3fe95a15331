#!/bin/bash
# usage: ab_bench.sh tag libpath [opts]
tag=$1; lib=$2; opts=$3
MBPE_LIB=$lib MBPE_BENCH_OPTS=$opts python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-full-run > gpurun_out/r4_ab_$tag.json 2> gpurun_out/r4_ab_$tag.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_ab_$tag.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$tag', 'value %.0f'%d['value'], 'passes', d['timed_region']['stream_passes']/3, 'fused ms avg %.3f'%r['avg_launch_ms'], 'frac %.3f'%r['frac'], 'pc %.3f'%d['roofline_pair_count']['frac'])
PY
