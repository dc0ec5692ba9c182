#!/bin/bash
# the headline through the slot pipeline at several slot counts (GPU box): tools/slots_sweep.sh "272 288 304" "9" [repeats]
for s in ${1:-256 288 320}; do for d in ${2:-9}; do for r in $(seq 1 ${3:-1}); do
echo -n "slots $s depth $d: "; timeout 200 python bench.py --no-cpu-baseline --no-extra-legs --pipeline-slots $s --pipeline-depth $d 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(round(j['value']), j['ms_per_step'], round(j['roofline']['frac'], 3))"
done; done; done
