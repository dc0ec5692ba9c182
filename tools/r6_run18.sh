mkdir -p gpurun_out/r6s
export JD_DEV=1
for rn in 1 0 1 0; do
  if [ "$rn" = "1" ]; then export JD_NO_RENUMBER=1; else unset JD_NO_RENUMBER; fi
  for leg in clg north c3 c2; do
    JD_VERBOSE=1 JD_BENCH_NO_LAZY=1 python tools/run_leg.py $leg 3 2> gpurun_out/r6s/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_renumber=$rn $leg', d['value'], d['ms_per_step'], d['roofline']['frac'])"
    grep "state numbers\|per-state words" gpurun_out/r6s/err.txt | sort | uniq -c | head -4
  done
  python bench.py --no-extra-legs --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_renumber=$rn headline', d['value'], d['ms_per_step'])"
done
unset JD_NO_RENUMBER JD_DEV
python -m pytest tests -x -q -m gpu -k "not multirank" > gpurun_out/r6s/pytest.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6s/pytest.log
