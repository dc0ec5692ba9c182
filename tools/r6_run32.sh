mkdir -p gpurun_out/r6ad
export JD_DEV=1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_slot.py tests/test_gpu_refgolden.py -x -q > gpurun_out/r6ad/pytest_a.log 2>&1; echo "parity+slot+golden rc=$?"; tail -3 gpurun_out/r6ad/pytest_a.log
for nl in 1 0 1 0; do
  if [ $nl = 1 ]; then export JD_NO_LINK=1; else unset JD_NO_LINK; fi
  for leg in clg north c3 c2; do
    JD_BENCH_NO_LAZY=1 python tools/run_leg.py $leg 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nolink $nl $leg', d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'), d['per_stream_frame'].get('tot_arcs_walked'))"
  done
  python bench.py --no-extra-legs --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nolink $nl headline', d['value'], d['ms_per_step'])"
done
