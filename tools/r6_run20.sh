mkdir -p gpurun_out/r6u
export JD_DEV=1
for sp in 2 3 2 3; do
  export JD_SREC_SPLIT=$sp
  for leg in north c3 clg c2; do
    JD_BENCH_NO_LAZY=1 python tools/run_leg.py $leg 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $sp $leg', d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'))"
  done
  python bench.py --no-extra-legs --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $sp headline', d['value'], d['ms_per_step'])"
done
export JD_SREC_SPLIT=3
python -m pytest tests -x -q -m gpu -k "not multirank" > gpurun_out/r6u/pytest3.log 2>&1; echo "tests split3 rc=$?"; tail -3 gpurun_out/r6u/pytest3.log
