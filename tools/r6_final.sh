mkdir -p gpurun_out/r6final
python -m pytest tests -x -q -m gpu > gpurun_out/r6final/pytest_full.log 2>&1; echo "full rc=$?"; tail -3 gpurun_out/r6final/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6final/bench_driver_like.json 2> gpurun_out/r6final/bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r6final/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r6final/bench_driver_like.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_design'], d['roofline']['frac_measured'], d['roofline']['l2_requests'])
print(d['wer_vs_oracle']['wer'], d['wer_vs_oracle']['identical_1best'], d['cpu_baseline']['value'])
print({k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['legs'].items()})
P
