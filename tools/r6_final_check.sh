mkdir -p gpurun_out/r6final
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
T0=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6final/bench.json 2> gpurun_out/r6final/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"; tail -2 gpurun_out/r6final/bench.err | cut -c1-400
python - <<'P'
import json
d=json.loads(open('gpurun_out/r6final/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_design'), d['roofline'].get('frac_measured'), d['cpu_baseline']['value'], d['cpu_baseline'].get('reference_cpu',{}).get('differential_cases_identical'), d['wer_vs_oracle']['wer'])
print({k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['legs'].items()})
P
