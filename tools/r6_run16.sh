mkdir -p gpurun_out/r6q
python -m pytest tests -x -q -m gpu -k "not multirank" > gpurun_out/r6q/pytest.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6q/pytest.log
JD_VERBOSE=1 python bench.py > gpurun_out/r6q/bench.json 2> gpurun_out/r6q/bench.err; echo "bench rc=$?"
grep "per-state words" gpurun_out/r6q/bench.err | sort | uniq -c
python - <<'P'
import json
d=json.loads(open('gpurun_out/r6q/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print({k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['legs'].items()})
P
