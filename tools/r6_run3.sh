mkdir -p gpurun_out/r6c
O=gpurun_out/r6c
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "falls_back" > $O/pytest_fallback.log 2>&1; echo "fallback rc=$?"; tail -3 $O/pytest_fallback.log
for v in item_always item_cond item_always item_cond; do
  cp build_ab/lib_$v.so juicer_amd/libjuicer_amd.so
  python bench.py --no-extra-legs --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v headline', d['value'], d['ms_per_step'], d['roofline']['per_stream_frame'].get('tot_entry_items'), d['roofline']['per_stream_frame'].get('tot_recs_read'))"
done
for v in item_always item_cond; do
  cp build_ab/lib_$v.so juicer_amd/libjuicer_amd.so
  for leg in north c3 clg c2; do
    JD_BENCH_NO_LAZY=1 python tools/run_leg.py $leg 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $leg', d['value'], d['ms_per_step'], d['roofline']['frac'], d['per_stream_frame'].get('tot_entry_items'), d['per_stream_frame'].get('tot_recs_read'))"
  done
done
# the scoring option with the chip split between slots (two per CU) and scoring: development knob JD_SLOT_KEEP_SE = CUs per shader engine the slots get
for cfg in "0 256 9 exact" "0 256 9 fast" "6 384 13 fast" "5 320 11 fast" "7 448 15 fast" "6 384 13 exact"; do
  set -- $cfg
  if [ "$1" = "0" ]; then unset JD_SLOT_KEEP_SE; else export JD_DEV=1 JD_SLOT_KEEP_SE=$1; fi
  python bench.py --no-extra-legs --no-cpu-baseline --steps 50 --pipeline-slots $2 --pipeline-depth $3 --scoring $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keep_se $1 slots $2 depth $3 $4:', d['value'], d['ms_per_step'], d['config']['pipeline_error'])"
done
unset JD_SLOT_KEEP_SE
python -m pytest tests -x -q -m gpu > $O/pytest_full.log 2>&1; echo "full rc=$?"; tail -3 $O/pytest_full.log
