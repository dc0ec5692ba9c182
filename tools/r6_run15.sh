mkdir -p gpurun_out/r6p
for v in nosplit split nosplit split; do
  cp build_ab/lib_$v.so juicer_amd/libjuicer_amd.so
  python bench.py --no-extra-legs --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v headline', d['value'], d['ms_per_step'])"
done
for v in nosplit split nosplit split; do
  cp build_ab/lib_$v.so juicer_amd/libjuicer_amd.so
  for leg in north c3 clg c2 c512slot; do
    JD_BENCH_NO_LAZY=1 python tools/run_leg.py $leg 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $leg', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
cp build_ab/lib_split.so juicer_amd/libjuicer_amd.so
python -m pytest tests -x -q -m gpu -k "not multirank" > gpurun_out/r6p/pytest_split.log 2>&1; echo "split tests rc=$?"; tail -3 gpurun_out/r6p/pytest_split.log
