#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_broker.py tests/test_gpu_lazy.py -q -m gpu -s -k "broker or threads or processes or generations" > gpurun_out/r4_tests_b3.log 2>&1
grep -n "callers through\|broker:\|passed\|failed\|Error" gpurun_out/r4_tests_b3.log | grep -v print | head
python bench.py --no-cpu-baseline > gpurun_out/r4_bench_b.json 2> gpurun_out/r4_bench_b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d['legs'].items():
    if isinstance(v,dict): print(k, {kk:v.get(kk) for kk in ('value','ms_per_step','search_ms','batches_in_flight')}, v.get('roofline',{}).get('frac'))
PY
