mkdir -p gpurun_out/r6e
python -m pytest tests -x -q -m gpu > gpurun_out/r6e/pytest_full.log 2>&1; echo "full rc=$?"; tail -4 gpurun_out/r6e/pytest_full.log
python tools/counters_diag.py 2>&1 | grep -v amdgpu.ids
for leg in c512slot c512slotfast; do python tools/run_leg.py $leg 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$leg', d['value'], d['ms_per_step'], d['gmm_ms'], d['roofline']['frac'], d['roofline'].get('frac_design'))"; done
