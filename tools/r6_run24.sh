export JD_DEV=1
for rn in 5 10 11 12 5 10 11 12; do
  export JD_RENUMBER=$rn
  for leg in clg; do
    JD_BENCH_NO_LAZY=1 python tools/run_leg.py $leg 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('renumber $rn $leg', d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'))"
  done
done
for rn in 0 10 11 12; do
  export JD_RENUMBER=$rn
  for leg in north c3; do
    JD_BENCH_NO_LAZY=1 python tools/run_leg.py $leg 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('renumber $rn $leg', d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'))"
  done
done
