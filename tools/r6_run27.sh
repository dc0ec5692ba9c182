mkdir -p gpurun_out/r6y
export JD_DEV=1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_slot.py -x -q > gpurun_out/r6y/pytest_a.log 2>&1; echo "parity+slot rc=$?"; tail -3 gpurun_out/r6y/pytest_a.log
for ns in 1 0 1 0; do
  if [ $ns = 1 ]; then export JD_NO_SOLE=1; else unset JD_NO_SOLE; fi
  for leg in clg north c3 c2; do
    JD_BENCH_NO_LAZY=1 python tools/run_leg.py $leg 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nosole $ns $leg', d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'))"
  done
  python bench.py --no-extra-legs --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nosole $ns headline', d['value'], d['ms_per_step'])"
done
