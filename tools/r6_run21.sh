mkdir -p gpurun_out/r6v
export JD_DEV=1
for rn in 1 2 3 1 2 3; do
  export JD_RENUMBER=$rn
  for leg in clg; do
    JD_BENCH_NO_LAZY=1 python tools/run_leg.py $leg 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('renumber $rn $leg', d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'))"
  done
done
for rn in 0 2 3; do
  export JD_RENUMBER=$rn
  for leg in north c3; do
    JD_BENCH_NO_LAZY=1 python tools/run_leg.py $leg 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('renumber $rn $leg', d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'))"
  done
done
