#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for extra in "" "--pipeline-slots 300"; do
echo "== bench.py $extra"
timeout 600 python bench.py --no-cpu-baseline --no-extra-legs $extra 2>/tmp/err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['roofline']['kernel'], d['config']['batches_in_flight'], d['config']['pipeline_error'])"
grep "bench.py:" /tmp/err
done
