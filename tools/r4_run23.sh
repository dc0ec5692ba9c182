#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1200 python bench.py 2> gpurun_out/r4_bench_pipe.err > gpurun_out/r4_bench_pipe.json
tail -3 gpurun_out/r4_bench_pipe.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_bench_pipe.json'))
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['config'].get('pipeline'))
r=d['roofline']; print({k:r[k] for k in r if k not in('legs','gmm','serial_order','launch')})
print('serial', r['serial_order'])
for k,v in d['legs'].items():
    print(k, v if isinstance(v,str) else (v.get('value'), v.get('ms_per_step'), v.get('roofline',{}).get('frac'), v.get('cpu_oracle',{}).get('identical_1best')))
print(d['cpu_baseline'])
PY
