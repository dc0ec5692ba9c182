export JD_DEV=1
for ch in 128 64 32 128 64; do
  export JD_PIPE_CHUNK=$ch
  for k in 20 50; do
    python bench.py --no-extra-legs --no-cpu-baseline --steps $k 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $ch K $k:', d['value'], d['ms_per_step'])"
  done
done
