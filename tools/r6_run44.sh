export JD_DEV=1
for xc in 0 1 0 1; do
  if [ $xc = 1 ]; then export JD_XCUT=1; else unset JD_XCUT; fi
  for leg in north c3 clg; do
    JD_BENCH_NO_LAZY=1 python tools/run_leg.py $leg 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcut $xc $leg', d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'))"
  done
done
