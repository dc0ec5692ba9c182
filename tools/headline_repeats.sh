#!/bin/bash
# python bench.py --no-cpu-baseline --no-extra-legs N times on one box (GPU): value, ms per step, K, W, frac, frac_measured, frac_design,
# frames inside the brackets, pipeline error - one line per run (profiles/r05_headline_repeats.log)
cd "$(dirname "$0")/.." || exit 1
for r in $(seq 1 ${1:-5}); do
timeout 200 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(j['value'], j['ms_per_step'], j['steps'], j['warmup'], r['frac'], r['frac_measured'], r.get('frac_design'), j.get('frames_timed'), j['config'].get('pipeline_error'))"
done
