mkdir -p gpurun_out/r6a
python -m pytest tests -x -q -m gpu > gpurun_out/r6a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6a/pytest.log
for v in nocount count nocount count; do
  cp build_ab/lib_$v.so juicer_amd/libjuicer_amd.so
  python bench.py --no-extra-legs --no-cpu-baseline --steps 50 > gpurun_out/r6a/bench_$v.$RANDOM.json 2> gpurun_out/r6a/bench_err.log
done
tail -3 gpurun_out/r6a/pytest.log
for f in gpurun_out/r6a/bench_*.json; do python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
